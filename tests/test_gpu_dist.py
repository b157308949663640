"""-m gpu: the sharded step with the HIP op backend.  P virtual ranks run as threads of one process on the one
GPU of the test box (ThreadComm); the result must match the single-GPU plan."""
import threading

import numpy as np
import pytest

from conftest import golden_batches, golden_csr, golden_params, load_golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _run_sharded(case, world):
    from gcn_drug_repurposing_amd.dist import HipOps, ShardedEngine, ThreadComm
    g = load_golden(case)
    n, d, L = (int(v) for v in g["meta"])
    shared = ThreadComm.Shared(world)
    shared.barrier = threading.Barrier(world, timeout=120)
    results, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                eng = ShardedEngine(golden_csr(g, "A"), g["X"], golden_params(g, "init"), num_layers=L, layer_decay=float(g["decay"]),
                                    alpha=float(g["alpha"]), lr=float(g["lr"]), comm=ThreadComm(shared, rank), ops=HipOps("cuda:0"),
                                    device=torch.device("cuda:0"))
                losses = []
                for idx in golden_batches(g):
                    eng.step(torch.from_numpy(idx.astype(np.int32)).cuda(), float(g["beta"]))
                    losses.append(float(eng.loss.item()))
                emb = eng.gather_embeddings().cpu().numpy()
                results[rank] = (losses, emb, [p.cpu().numpy() for p in eng.params])
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            shared.barrier.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errors, errors
    return g, results


@pytest.mark.parametrize("world,case", [(2, "edge_n600_d128_L2"), (3, "knn_n2000_d64_L3"), (4, "knn_n200_d16_L2")])
def test_sharded_hip_matches_reference(world, case):
    g, res = _run_sharded(case, world)
    for r in range(1, world):
        assert res[r][0] == res[0][0]
        np.testing.assert_array_equal(res[r][1], res[0][1])
    losses, emb, params = res[0]
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-4, atol=1e-8)
    assert np.abs(emb - g["emb_last"]).max() / np.abs(g["emb_last"]).max() < 2e-3
    for k, p in zip(("W1", "b1", "W2", "b2"), params):
        assert np.abs(p - g["final_" + k]).max() < 2.5 * float(g["lr"]), k
