"""CPU: how much do equally valid executions of the reference's op sequence differ after the fixtures' training steps?  The answer is
what the multi-step GPU tolerances are a multiple of (tests/tolerances.py)."""
import numpy as np
import pytest

import tolerances as T
from conftest import golden_batches, golden_csr, golden_params, load_golden

torch = pytest.importorskip("torch")


def _run(g, threads, dtype):
    from oracle.torch_cpu_path import TorchCpuPath
    n, d, L = (int(v) for v in g["meta"])
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        cpu = TorchCpuPath(golden_csr(g, "Ahat").astype(np.float32), g["X"], golden_params(g, "init"), L, float(g["decay"]), float(g["alpha"]),
                           float(g["lr"]), dtype=dtype)
        losses, emb = [], None
        for idx in golden_batches(g):
            emb, loss = cpu.step(idx.astype(np.int64), float(g["beta"]))
            losses.append(loss)
        return (np.array(losses), emb.numpy().astype(np.float64), {k: cpu.p[k].detach().numpy().astype(np.float64) for k in cpu.p})
    finally:
        torch.set_num_threads(old)


def _diff(a, b, lr):
    dl = np.abs(a[0] - b[0]).max() / np.abs(b[0]).max()
    de = np.abs(a[1] - b[1]).max() / np.abs(b[1]).max()
    dw = max(np.abs(a[2][k] - b[2][k]).max() for k in a[2]) / lr
    return dl, de, dw


@pytest.mark.parametrize("case", ["toy_sif_d64_L2", "knn_n200_d16_L2", "knn_n2000_d64_L3", "edge_n600_d128_L2"])
def test_spread_between_cpu_executions_is_what_the_gpu_tolerances_are_a_multiple_of(case):
    g = load_golden(case)
    lr = float(g["lr"])
    runs = {"fp32, 1 thread": _run(g, 1, torch.float32), "fp32, 8 threads": _run(g, 8, torch.float32), "fp64": _run(g, 8, torch.float64)}
    fixture = (g["losses"], g["emb_last"].astype(np.float64), {k: g["final_" + k].astype(np.float64) for k in ("W1", "b1", "W2", "b2")})
    runs["the reference's own run (fixture)"] = fixture
    names = list(runs)
    worst = [0.0, 0.0, 0.0]
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            for k, v in enumerate(_diff(runs[names[i]], runs[names[j]], lr)):
                worst[k] = max(worst[k], float(v))
    print(f"{case}: spread over {len(names)} executions: loss {worst[0]:.2e} rel, emb {worst[1]:.2e} rel, weights {worst[2]:.2e} lr")
    # the recorded spread bounds what is measured here (otherwise the tolerances derived from it would be claims again)
    assert worst[0] <= T.SPREAD_LOSS_REL, worst
    assert worst[1] <= T.SPREAD_EMB_REL, worst
    assert worst[2] <= T.SPREAD_WEIGHT_LR, worst
    # and the port agrees with the reference's own run far inside the device tolerances
    dl, de, dw = _diff(runs["fp32, 8 threads"], fixture, lr)
    assert dl <= T.TRAJ_LOSS_RTOL / 10 and de <= T.TRAJ_EMB_REL / 10 and dw <= T.TRAJ_WEIGHT_LR / 10


def test_tolerances_are_a_stated_multiple_of_the_spread():
    assert T.TRAJ_LOSS_RTOL == T.TRAJ_K * T.SPREAD_LOSS_REL and T.TRAJ_EMB_REL == T.TRAJ_K * T.SPREAD_EMB_REL
    assert T.TRAJ_WEIGHT_LR == T.TRAJ_K * T.SPREAD_WEIGHT_LR and 5 <= T.TRAJ_K <= 100
