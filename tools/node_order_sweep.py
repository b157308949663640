#!/usr/bin/env python3
"""How much does the node order matter for the cache-resident SpMM (config 2)?  plain SpMM time under several relabellings."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402
from scipy.sparse.csgraph import reverse_cuthill_mckee  # noqa: E402

import gcn_drug_repurposing_amd as pkg  # noqa: E402
from gcn_drug_repurposing_amd import _lib, synth  # noqa: E402
from gcn_drug_repurposing_amd.graph import GssGraph  # noqa: E402

lib = pkg.load()
adj, ntype, _ = synth.whole_graph_standin(1)
n, d = adj.shape[0], 128
a = sp.csr_matrix(adj)
deg = np.diff(a.indptr) + np.diff(sp.csr_matrix(a.T).indptr)
sym = sp.csr_matrix(((a + a.T) > 0).astype(np.int8))
orders = {
    "loader order (networkx insertion)": np.arange(n),
    "degree descending (hub-first)": np.argsort(-deg, kind="stable"),
    "degree ascending": np.argsort(deg, kind="stable"),
    "random": np.random.RandomState(0).permutation(n),
    "reverse Cuthill-McKee": np.asarray(reverse_cuthill_mckee(sym, symmetric_mode=True)),
    "by node type, then degree descending": np.lexsort((-deg, ntype)),
}
xx = torch.randn(4096, 4096, device="cuda")
for _ in range(300):
    xx @ xx
st = _lib.current_stream()
for name, perm in orders.items():
    ap = a[perm][:, perm]
    g = GssGraph(ap, need_transpose=False)
    x = torch.randn(n, d, device="cuda")
    y = torch.empty(n, d, device="cuda")
    best = 1e9
    for _ in range(3):
        for _ in range(30):
            lib.gss_spmm(g.a.handle, d, x.data_ptr(), y.data_ptr(), None, None, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            lib.gss_spmm(g.a.handle, d, x.data_ptr(), y.data_ptr(), None, None, st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
    print(f"{name:42s} {best:6.1f} us")
