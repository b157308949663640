import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def golden_csr(g, prefix):
    import scipy.sparse as sp
    n = int(g["meta"][0])
    return sp.csr_matrix((g[prefix + "_data"], g[prefix + "_indices"], g[prefix + "_indptr"]), shape=(n, n))


def golden_batches(g):
    out, o = [], 0
    for s in g["batch_sizes"]:
        out.append(g["batches"][o:o + int(s)])
        o += int(s)
    return out


def golden_params(g, prefix):
    return {k: g[f"{prefix}_{k}"].copy() for k in ("W1", "b1", "W2", "b2")}


OP_CASES = ["toy_sif_d64_L2", "knn_n200_d16_L2", "knn_n2000_d64_L3", "edge_n600_d128_L2"]


@pytest.fixture(params=OP_CASES)
def op_case(request):
    return request.param, load_golden(request.param)


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def record_measured(name, **values):
    """GSS_RECORD_PARITY=<file>: append what a parity test measured (the figures its tolerances are a stated multiple of) as a JSON line"""
    path = os.environ.get("GSS_RECORD_PARITY")
    if not path:
        return
    import json
    with open(path, "a") as f:
        f.write(json.dumps({"test": name, **{k: float(v) for k, v in values.items()}}) + "\n")
