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


def test_bench_sharded_path_over_rccl_single_rank(tmp_path):
    """bench.py's multi-GPU branch (torch.distributed 'nccl' == RCCL, TorchComm, ShardedEngine) with one rank, launched
    the way the driver launches it; must agree with the single-GPU plan's loss after the same steps."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSS_FORCE_SHARDED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    sharded = json.loads(line)
    assert "node-range shards" in sharded["config"]["parallelism"]
    ref = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600)
    assert ref.returncode == 0, ref.stderr[-2000:]
    single = json.loads([l for l in ref.stdout.splitlines() if l.startswith("{")][-1])
    assert abs(sharded["config"]["final_loss"] - single["config"]["final_loss"]) < 2e-4 * abs(single["config"]["final_loss"])


def test_trainer_cli_sharded_mode_writes_the_same_embeddings(tmp_path):
    """train.py under the RCCL sharded engine (one rank) vs the single-GPU plan: same graph_embs.txt"""
    import os
    import subprocess
    import sys
    g = load_golden("train_py_n200_d16")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emb_path = tmp_path / "in.embs.txt"
    emb_path.write_bytes(bytes(g["in_embs_txt"]))
    outs = []
    for mode, env_extra in (("single", {}), ("sharded", {"GSS_FORCE_SHARDED": "1", "MASTER_PORT": "29541"})):
        out = tmp_path / f"{mode}.txt"
        cmd = [sys.executable, os.path.join(root, "train.py"), "--emb-file", str(emb_path), "--num-layers", "2", "--hidden-units", "16",
               "--k", "5", "--epochs", "3", "--lr", "0.0003", "--beta-percentile", "98", "--batch-size", "64", "--seed", "7", "--out", str(out)]
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **env_extra), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.loadtxt(str(out)))
    assert np.abs(outs[0] - outs[1]).max() < 1e-5
    ref = np.loadtxt(bytes(g["graph_embs_txt"]).decode().splitlines())
    assert np.abs(outs[1] - ref).max() < 5e-3
