"""-m gpu: the NATIVE sharded plan (gss_plan_create_sharded).  P ranks run as threads of one process on the one GPU of the
test box (gss_comm_create_local); the result must match the fixtures and the single-GPU plan.  The RCCL backend runs with the one
rank a single GPU allows; with two or more visible GPUs test_two_rccl_ranks_* starts real ranks as child processes."""
import threading

import numpy as np
import pytest

import tolerances as T
from conftest import golden_batches, golden_csr, golden_params, load_golden

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]
torch = pytest.importorskip("torch")


def _run_native(adj, X, params, L, batches, beta, world, decay=0.3, alpha=1.0, lr=3e-4, max_batch=None, fused_step=True, a_hat=None):
    """the NATIVE sharded plan (gss_plan_create_sharded + gss_comm_create_local): `world` ranks as threads of this process,
    each with its own stream on the one GPU -> per rank (losses, full embeddings, params, grads)"""
    from gcn_drug_repurposing_amd.dist import local_comms, sharded_plan_engine
    comms = local_comms(world)
    results, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                eng = sharded_plan_engine(adj, X, params, comms[rank], num_layers=L, layer_decay=decay, alpha=alpha, lr=lr,
                                          max_batch=max_batch, device=torch.device("cuda:0"), a_hat=a_hat)
                losses = []
                for idx in batches:
                    t = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int32)).cuda()
                    if fused_step:
                        eng.step(t, beta)
                    else:
                        eng.forward()
                        eng.loss_backward(t, beta)
                        eng.adam()
                    losses.append(float(eng.loss.item()))
                emb = eng.gather_embeddings().cpu().numpy()
                torch.cuda.current_stream().synchronize()
                results[rank] = (losses, emb, [p.cpu().numpy() for p in eng.params], [g.cpu().numpy() for g in eng.grads], eng.part.bounds.copy())
        except Exception as e:  # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))
            comms[rank].abort()               # release the peers now instead of after the barrier timeout

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errors, errors
    assert all(r is not None for r in results), "a rank thread did not finish"
    return results


@pytest.mark.parametrize("fused_step", [True, False])
@pytest.mark.parametrize("world,case", [(1, "edge_n600_d128_L2"), (2, "edge_n600_d128_L2"), (3, "knn_n2000_d64_L3"), (4, "knn_n200_d16_L2"),
                                        (8, "toy_sif_d64_L2")])
def test_native_sharded_plan_matches_reference(world, case, fused_step):
    """C1-C3 inside the C++ plan.  toy_sif at world 8 has more shards than some ranks have rows (empty shards)."""
    g = load_golden(case)
    n, d, L = (int(v) for v in g["meta"])
    res = _run_native(golden_csr(g, "A"), g["X"], golden_params(g, "init"), L, golden_batches(g), float(g["beta"]), world,
                      decay=float(g["decay"]), alpha=float(g["alpha"]), lr=float(g["lr"]), fused_step=fused_step)
    for r in range(1, world):
        assert res[r][0] == res[0][0]                       # replicated state is bit-identical on every rank
        np.testing.assert_array_equal(res[r][1], res[0][1])
        for a, b in zip(res[r][2], res[0][2]):
            np.testing.assert_array_equal(a, b)
    losses, emb, params, _, _ = res[0]
    np.testing.assert_allclose(losses, g["losses"], rtol=T.TRAJ_LOSS_RTOL, atol=1e-9)
    assert np.abs(emb - g["emb_last"]).max() / np.abs(g["emb_last"]).max() < T.TRAJ_EMB_REL
    for k, p in zip(("W1", "b1", "W2", "b2"), params):
        assert np.abs(p - g["final_" + k]).max() < T.TRAJ_WEIGHT_LR * float(g["lr"]), k


def test_native_sharded_plan_first_step_equals_single_gpu_plan():
    """one forward + loss + backward, world 3 vs the single-GPU plan on the same inputs: embeddings and loss bit-identical
    (row results do not depend on the shard), gradients to fp32 summation order"""
    from gcn_drug_repurposing_amd.engine import GssEngine
    from gcn_drug_repurposing_amd.graph import GssGraph
    g = load_golden("knn_n2000_d64_L3")
    n, d, L = (int(v) for v in g["meta"])
    idx = golden_batches(g)[0]
    res = _run_native(golden_csr(g, "A"), g["X"], golden_params(g, "init"), L, [idx], float(g["beta"]), 3, decay=float(g["decay"]),
                      alpha=float(g["alpha"]), lr=float(g["lr"]), fused_step=False)
    graph = GssGraph(golden_csr(g, "A"))        # both normalise on the device: the same A_hat bits
    params = [torch.from_numpy(g["init_" + k].copy()).cuda() for k in ("W1", "b1", "W2", "b2")]
    eng = GssEngine(graph, torch.from_numpy(g["X"]).cuda(), params, num_layers=L, layer_decay=float(g["decay"]), alpha=float(g["alpha"]),
                    lr=float(g["lr"]))
    eng.forward()
    emb1 = eng.emb.cpu().numpy().copy()
    eng.loss_backward(torch.from_numpy(idx.astype(np.int32)).cuda(), float(g["beta"]))
    np.testing.assert_array_equal(res[0][1], emb1)          # the embeddings of the forward (Adam does not touch them)
    assert res[0][0][0] == float(eng.loss.item())
    for got, ref in zip(res[0][3], eng.grads):
        ref = ref.cpu().numpy()
        assert np.abs(got - ref).max() < 1e-5 * np.abs(ref).max() + 1e-12


@pytest.mark.parametrize("world,L,cache", [(2, 1, False), (5, 3, False), (3, 2, True)])
def test_native_sharded_plan_other_depths_and_layer1_cache(world, L, cache):
    """depths the fixtures do not have (L = 1: no A_hat^T, no halo of it; L = 3 at an odd world) and cache_layer1 on shards,
    against the single-GPU plan: embeddings and losses bit-identical over three steps"""
    from gcn_drug_repurposing_amd.dist import local_comms, sharded_plan_engine
    from gcn_drug_repurposing_amd.engine import GssEngine
    from gcn_drug_repurposing_amd.graph import GssGraph
    g = load_golden("knn_n2000_d64_L3")
    adj, X, p0 = golden_csr(g, "A"), g["X"], golden_params(g, "init")
    batches = golden_batches(g)[:3]
    kw = dict(num_layers=L, layer_decay=0.4, alpha=1.0, lr=1e-3)
    ref = GssEngine(GssGraph(adj, need_transpose=L > 1), torch.from_numpy(X).cuda(),
                    [torch.from_numpy(p0[k].copy()).cuda() for k in ("W1", "b1", "W2", "b2")], cache_layer1=cache, **kw)
    ref_out = []
    for idx in batches:
        ref.step(torch.from_numpy(idx.astype(np.int32)).cuda(), 0.3)
        ref_out.append((ref.loss.item(), ref.emb.cpu().numpy().copy()))
    comms = local_comms(world)
    results, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                eng = sharded_plan_engine(adj, X, p0, comms[rank], device=torch.device("cuda:0"), cache_layer1=cache, **kw)
                out = []
                for idx in batches:
                    eng.step(torch.from_numpy(idx.astype(np.int32)).cuda(), 0.3)
                    eng.check_guards()
                    out.append((eng.loss.item(), eng.gather_embeddings().cpu().numpy()))
                results[rank] = out
        except Exception as e:  # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))
            comms[rank].abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errors, errors
    for rank in range(world):
        for (l_ref, e_ref), (l_got, e_got) in zip(ref_out, results[rank]):
            # step 1 is bit-identical; later steps see weights whose gradients were summed in another order (fp32)
            assert abs(l_got - l_ref) < 1e-6 * abs(l_ref) + 1e-9
            assert np.abs(e_got - e_ref).max() < 2e-6
        assert results[rank][0][0] == ref_out[0][0]
        np.testing.assert_array_equal(results[rank][0][1], ref_out[0][1])


def test_native_collectives_local_backend():
    """gss_allgather_rows / gss_allreduce_sum through the in-process backend, uneven shards"""
    from gcn_drug_repurposing_amd.dist import local_comms
    world, d, maxr = 3, 16, 5
    rows = [5, 3, 0]
    comms = local_comms(world)
    out, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                src = torch.zeros(maxr, d, device="cuda")
                src[:rows[rank]] = rank + 1 + torch.arange(rows[rank], device="cuda").float()[:, None]
                dst = torch.full((world * maxr, d), -1.0, device="cuda")
                comms[rank].all_gather_rows(src, dst)
                t = torch.full((7,), float(rank + 1), device="cuda")
                comms[rank].all_reduce_sum_(t)
                torch.cuda.current_stream().synchronize()
                out[rank] = (dst.cpu(), t.cpu())
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            comms[rank].abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errors, errors
    for r in range(world):
        dst, t = out[r]
        assert torch.equal(t, torch.full((7,), 6.0))
        for o in range(world):
            blk = dst[o * maxr:(o + 1) * maxr]
            for k in range(rows[o]):
                assert torch.equal(blk[k], torch.full((d,), float(o + 1 + k)))
            assert torch.equal(blk[rows[o]:], torch.zeros(maxr - rows[o], d))


def test_local_backend_records_and_replays_what_collectives_deliver():
    """gss_comm_local_mode (the measurement aid behind tools/scaling_forecast.py): in mode 1 every collective keeps a device copy of what it
    DELIVERED to this rank; in mode 2 the same call sequence is served from those copies with no peer taking part -- here by ONE rank
    thread while the other never calls --, a different size is refused, and mode 0 forgets the log.  Also: the sharded plan's recorded step
    replays alone (same collective sequence) and leaves its guards intact."""
    from gcn_drug_repurposing_amd import GssError
    from gcn_drug_repurposing_amd.dist import local_comms
    world, d = 2, 16
    comms = local_comms(world)
    out, errors = [None] * world, []
    off_send = [np.array([0, 0, 3], dtype=np.int64), np.array([0, 2, 2], dtype=np.int64)]     # rank 0 sends 3 rows to rank 1, rank 1 sends 2 to rank 0
    off_recv = [np.array([0, 0, 2], dtype=np.int64), np.array([0, 3, 3], dtype=np.int64)]

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                c = comms[rank]
                c.local_mode(1)
                send = (100.0 * (rank + 1) + torch.arange(4 * d, device="cuda").float()).view(4, d)
                recv = torch.zeros(4, d, device="cuda")
                c.exchange_rows(d, send, off_send[rank], recv, off_recv[rank])
                t = torch.full((9,), float(rank + 1), device="cuda")
                c.all_reduce_sum_(t)
                g = c.allgather_bytes(torch.full((5,), float(10 + rank), device="cuda"))
                torch.cuda.current_stream().synchronize()
                out[rank] = (recv.clone(), t.clone(), g.clone(), c.local_log())
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            comms[rank].abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errors, errors
    n_recv0 = int(off_recv[0][-1])
    assert out[0][3] == [n_recv0 * d * 4, 9 * 4, 2 * 5 * 4]
    # replay on rank 0 ALONE (rank 1 does nothing): no barrier, the recorded payloads come back
    c = comms[0]
    c.local_mode(2)
    recv = torch.full((4, d), -1.0, device="cuda")
    c.exchange_rows(d, torch.zeros(4, d, device="cuda"), off_send[0], recv, off_recv[0])
    t = torch.zeros(9, device="cuda")
    c.all_reduce_sum_(t)
    g = c.allgather_bytes(torch.zeros(5, device="cuda"))
    torch.cuda.synchronize()
    assert torch.equal(recv[:n_recv0], out[0][0][:n_recv0]) and torch.equal(recv[n_recv0:], torch.full((4 - n_recv0, d), -1.0, device="cuda"))
    assert torch.equal(t, torch.full((9,), 3.0, device="cuda")) and torch.equal(g, out[0][2])
    with pytest.raises(GssError):                      # a fourth collective was never recorded
        c.all_reduce_sum_(t)
    c.local_mode(2)                                    # rewinds
    with pytest.raises(GssError):                      # the first recorded collective delivered another size
        c.all_reduce_sum_(t)
    c.local_mode(0)
    assert c.local_log() == []
    from gcn_drug_repurposing_amd.dist import rccl_comm
    with pytest.raises(GssError):                      # only the in-process backend records
        rccl_comm(1, 0).local_mode(1)


def test_native_halo_exchange_local_backend():
    """gss_exchange_rows (the boundary form of C1) through the in-process backend: uneven lists, an empty pair"""
    import ctypes as C
    from gcn_drug_repurposing_amd import _lib
    from gcn_drug_repurposing_amd.dist import local_comms
    world, d = 3, 8
    # rows rank r sends to q: counts[r][q]
    counts = np.array([[0, 2, 5], [1, 0, 0], [3, 4, 0]], dtype=np.int64)
    comms = local_comms(world)
    out, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                send_off = np.concatenate([[0], np.cumsum(counts[rank])]).astype(np.int64)
                recv_off = np.concatenate([[0], np.cumsum(counts[:, rank])]).astype(np.int64)
                send = torch.zeros(max(int(send_off[-1]), 1), d, device="cuda")
                for q in range(world):
                    for k in range(int(counts[rank][q])):
                        send[send_off[q] + k] = 100 * rank + 10 * q + k
                recv = torch.full((max(int(recv_off[-1]), 1), d), -1.0, device="cuda")
                _lib.check(_lib.load().gss_exchange_rows(comms[rank].handle, d, send.data_ptr(), send_off.ctypes.data, recv.data_ptr(),
                                                         recv_off.ctypes.data, _lib.current_stream()))
                torch.cuda.current_stream().synchronize()
                out[rank] = (recv.cpu(), recv_off)
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            comms[rank].abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errors, errors
    for rank in range(world):
        recv, recv_off = out[rank]
        for q in range(world):
            for k in range(int(counts[q][rank])):
                assert torch.equal(recv[recv_off[q] + k], torch.full((d,), float(100 * q + 10 * rank + k)))


def test_rccl_backend_single_rank_collectives():
    """the RCCL code path itself (ncclAllGather / ncclAllReduce / the grouped point-to-point exchange) with a one-rank
    communicator -- all that one GPU allows; the multi-rank schedule is covered by the in-process backend above"""
    from gcn_drug_repurposing_amd import _lib
    from gcn_drug_repurposing_amd.dist import rccl_comm
    comm = rccl_comm(1, 0)
    src = torch.arange(5 * 16, dtype=torch.float32, device="cuda").view(5, 16)
    dst = torch.zeros(5, 16, device="cuda")
    comm.all_gather_rows(src, dst)
    t = torch.full((9,), 3.0, device="cuda")
    comm.all_reduce_sum_(t)
    off = np.zeros(2, dtype=np.int64)
    _lib.check(_lib.load().gss_exchange_rows(comm.handle, 16, src.data_ptr(), off.ctypes.data, dst.data_ptr(), off.ctypes.data, _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(dst, src) and torch.equal(t, torch.full((9,), 3.0, device="cuda"))
    assert comm.count() == 1                      # ncclCommCount
    comm.check()                                  # ncclCommGetAsyncError: healthy
    comm.sync(30.0)


def test_rccl_watchdog_aborts_instead_of_hanging():
    """gss_comm_sync: a stream that does not drain before the deadline (here: a one-second device spin; in a job: a peer that stopped
    taking part in a collective) aborts the communicator (ncclCommAbort) and returns GSS_ETIMEOUT; every later call on that
    communicator fails with GSS_ECOMM instead of enqueueing"""
    from gcn_drug_repurposing_amd import GssError
    from gcn_drug_repurposing_amd.dist import rccl_comm
    comm = rccl_comm(1, 0)
    t = torch.ones(8, device="cuda")
    comm.all_reduce_sum_(t)
    comm.sync(30.0)
    torch.cuda._sleep(int(2.0e9))                 # ~1 s of spinning on the current stream
    with pytest.raises(GssError, match="-110"):
        comm.sync(0.05)
    with pytest.raises(GssError, match="-104"):
        comm.all_reduce_sum_(t)
    with pytest.raises(GssError, match="-104"):
        comm.check()
    torch.cuda.synchronize()
    del comm                                      # destroying an aborted communicator is fine


def test_rccl_abort_from_another_thread_while_enqueueing():
    """RcclComm holds its mutex for pointer bookkeeping only, never across a call into RCCL -- abort() from a watchdog
    thread returns at once even while the main thread is busy enqueueing, the enqueues in flight end (success or -104, never a hang or a
    crash) and every later one is refused with -104.  (A rank stalled INSIDE RCCL behind a peer that never joins needs two devices:
    test_two_rccl_ranks_abort_releases_a_rank_whose_peer_never_joins.)"""
    import threading
    import time
    from gcn_drug_repurposing_amd import GssError
    from gcn_drug_repurposing_amd.dist import rccl_comm
    comm = rccl_comm(1, 0)
    t = torch.ones(1 << 16, device="cuda")
    comm.all_reduce_sum_(t)
    comm.sync(30.0)
    took = []

    def watchdog():
        time.sleep(0.05)
        t0 = time.time()
        comm.abort()
        took.append(time.time() - t0)

    th = threading.Thread(target=watchdog)
    th.start()
    refused = 0
    t_end = time.time() + 20.0
    while time.time() < t_end and refused < 5:
        try:
            comm.all_reduce_sum_(t)
        except GssError as e:
            assert "-104" in str(e)
            refused += 1
    th.join(30.0)
    assert not th.is_alive() and took and took[0] < 5.0, took
    assert refused == 5
    torch.cuda.synchronize()
    del comm


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_rccl_ranks_abort_releases_a_rank_whose_peer_never_joins(tmp_path):
    """rank 1 never calls the exchange; rank 0's watchdog thread aborts after a second: abort() must return and rank 0's enqueue / sync
    must end with an error instead of hanging (tests/mp_abort_job.py)"""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tests", "mp_abort_job.py")], capture_output=True, text=True,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), timeout=300)
    assert "rank 0: abort returned" in r.stdout and "rank 0: released with" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_local_backend_abort_is_reported_by_check():
    from gcn_drug_repurposing_amd import GssError
    from gcn_drug_repurposing_amd.dist import local_comms
    comms = local_comms(2)
    assert comms[0].count() == 2
    comms[0].check()
    comms[1].abort()
    with pytest.raises(GssError, match="-104"):
        comms[0].check()


def _visible_gpus():
    return torch.cuda.device_count()


@pytest.mark.skipif(_visible_gpus() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_rccl_ranks_trainer_matches_single_gpu(tmp_path):
    """The multi-process RCCL path proper: `train.py --ngpus 2` under torch.distributed.run (the torch NCCL process group and the
    plan's own communicator sharing the two GPUs; grouped ncclSend/ncclRecv halo exchange; grouped all-reduce of the four gradients)
    against the single-GPU trainer on the reference-generated fixture.  Started as a CHILD process before this process touches
    a second GPU."""
    import os
    import socket
    import subprocess
    import sys
    g = load_golden("train_py_n200_d16")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emb_path = tmp_path / "in.embs.txt"
    emb_path.write_bytes(bytes(g["in_embs_txt"]))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    common = ["--emb-file", str(emb_path), "--num-layers", "2", "--hidden-units", "16", "--k", "5", "--epochs", "3", "--lr", "0.0003",
              "--beta-percentile", "98", "--batch-size", "64", "--seed", "7"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "train.py")] + common + ["--ngpus", "2", "--out", str(tmp_path / "two.txt")],
                         capture_output=True, text=True, env=env, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    one = subprocess.run([sys.executable, os.path.join(root, "train.py")] + common + ["--out", str(tmp_path / "one.txt")],
                         capture_output=True, text=True, env=env, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    a, b = np.loadtxt(str(tmp_path / "two.txt")), np.loadtxt(str(tmp_path / "one.txt"))
    # the weight gradients are summed over the ranks in RCCL's order (not the single GPU's slice order): rounding-level differences
    assert np.abs(a - b).max() < 1e-5
    ref = np.loadtxt(bytes(g["graph_embs_txt"]).decode().splitlines())
    assert np.abs(a - ref).max() < T.TRAJ_CLI_EMB_ABS


@pytest.mark.skipif(_visible_gpus() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_rccl_ranks_subset_exchanges_and_overlapped_hops(tmp_path):
    """the parts of the sharded step that had only ever run on the in-process / host-staged backends, over RCCL between
    two devices -- the subset exchanges of knob lazy_halo (bitmaps by ncclSend / ncclRecv, the per-step counts through RcclComm::sync)
    and the overlapped hops on the plan's second stream (one communicator, two streams, event-ordered).  Forced on for a test-sized
    graph through the job-wide environment knobs; the result is held to the single-GPU trainer's."""
    import os
    import socket
    import subprocess
    import sys
    g = load_golden("train_py_n200_d16")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emb_path = tmp_path / "in.embs.txt"
    emb_path.write_bytes(bytes(g["in_embs_txt"]))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    common = ["--emb-file", str(emb_path), "--num-layers", "2", "--hidden-units", "16", "--k", "5", "--epochs", "3", "--lr", "0.0003",
              "--beta-percentile", "98", "--batch-size", "64", "--seed", "7"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "train.py")] + common + ["--ngpus", "2", "--out", str(tmp_path / "two.txt")],
                         capture_output=True, text=True, env=dict(env, GSS_OPTIONS="lazy_halo=1", GSS_SPLIT="1"), timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    one = subprocess.run([sys.executable, os.path.join(root, "train.py")] + common + ["--out", str(tmp_path / "one.txt")],
                         capture_output=True, text=True, env=env, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    a, b = np.loadtxt(str(tmp_path / "two.txt")), np.loadtxt(str(tmp_path / "one.txt"))
    assert np.abs(a - b).max() < 1e-5          # (own-column + boundary-column sums and RCCL's reduction order: rounding-level differences)


@pytest.mark.skipif(_visible_gpus() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_rccl_ranks_bench_line_is_complete():
    """`bench.py --gpus 2` starts its own two ranks; the line must carry what a scaling run is graded on"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--spinup-time", "0.05",
                        "--min-time", "0"], capture_output=True, text=True, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2
    for key in ("roofline", "xgmi", "per_rank", "comm_share", "kernel_ms_per_step", "value_executed", "collectives_per_step"):
        assert key in line, key
    assert len(line["per_rank"]["ms_per_step"]) == 2 and line["roofline"]["bound"] == "hbm"
    assert line["collectives_per_step"]["total"] <= 3.0          # L = 2: M_1's boundary rows, the batch rows, the weight gradients
    assert np.isfinite(line["config"]["final_loss"])


def _host_env():
    import os
    return dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GSS_COMM_BACKEND="host")


@pytest.mark.parametrize("ranks", [2, 3])
def test_multi_process_trainer_on_one_gpu_host_staged_backend(tmp_path, ranks):
    """The multi-PROCESS job on a one-GPU box: `train.py --ngpus N` under torch.distributed.run, every rank its own process with its own
    native sharded plan, collectives through the host-staged backend (gss_comm_create_host, gloo underneath) because RCCL refuses
    two ranks on one device.  Same assertions as the two-RCCL-rank test: the single-GPU trainer's embeddings and the
    reference-generated fixture."""
    import os
    import subprocess
    import sys
    g = load_golden("train_py_n200_d16")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emb_path = tmp_path / "in.embs.txt"
    emb_path.write_bytes(bytes(g["in_embs_txt"]))
    common = ["--emb-file", str(emb_path), "--num-layers", "2", "--hidden-units", "16", "--k", "5", "--epochs", "3", "--lr", "0.0003",
              "--beta-percentile", "98", "--batch-size", "64", "--seed", "7"]
    env = _host_env()
    if ranks == 3:      # the subset exchanges of knob lazy_halo (bitmap requests, per-step counts) across processes as well
        env["GSS_OPTIONS"] = "lazy_halo=1"
    multi = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                            "--master-port", str(_free_port()), os.path.join(root, "train.py")] + common
                           + ["--ngpus", str(ranks), "--out", str(tmp_path / "multi.txt")], capture_output=True, text=True, env=env, timeout=900)
    assert multi.returncode == 0, multi.stderr[-3000:]
    one = subprocess.run([sys.executable, os.path.join(root, "train.py")] + common + ["--out", str(tmp_path / "one.txt")],
                         capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    a, b = np.loadtxt(str(tmp_path / "multi.txt")), np.loadtxt(str(tmp_path / "one.txt"))
    assert np.abs(a - b).max() < 1e-5          # weight gradients summed over the ranks instead of over the single GPU's slices
    ref = np.loadtxt(bytes(g["graph_embs_txt"]).decode().splitlines())
    assert np.abs(a - ref).max() < T.TRAJ_CLI_EMB_ABS


def test_multi_process_bench_on_one_gpu_host_staged_backend():
    """`bench.py --gpus 2` as two processes sharing the GPU (rehearsal of the driver's multi-GPU launch: rendezvous, sharding by rank,
    barrier, max-over-ranks, the N>1 keys) -- the line is marked a rehearsal, its loss must equal the single-GPU run's"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--spinup-time", "0",
                        "--min-time", "0"], capture_output=True, text=True,
                       env=dict(_host_env(), GSS_OPTIONS="lazy_halo=1", GSS_LOCAL_TRANSPOSE="0"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["rehearsal"] is True
    sub = line["xgmi"]["subset_exchange_u"]        # the knob reached both processes: the backward hop fetched a subset of u's halo
    assert all(0 < f <= h for f, h in zip(sub["rows_fetched_last_step_by_rank"], sub["rows_of_the_whole_halo_by_rank"]))
    assert "REHEARSAL" in line["config"]["parallelism"]
    for key in ("roofline", "xgmi", "per_rank", "comm_share", "kernel_ms_per_step", "value_executed", "collectives_per_step"):
        assert key in line, key
    # L = 2 with the subset exchange of u forced on and the exchange-free last hop switched off (GSS_LOCAL_TRANSPOSE=0, so that the
    # subset exchange runs across processes): M_1's boundary rows (1), u's (bitmaps + rows: 2), the batch rows (1), the weight gradients
    # (1); X_1's boundary rows are recomputed, the batch rows' input gradients are computed on every rank
    cps = line["collectives_per_step"]
    assert cps["batch_row_allreduces"] == 1 and cps["weight_gradient_allreduces"] == 1 and cps["boundary_row_exchanges"] == 3, cps
    assert len(line["per_rank"]["ms_per_step"]) == 2 and sum(line["per_rank"]["rows"]) == 29960
    ref = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--spinup-time", "0", "--min-time", "0",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert ref.returncode == 0, ref.stderr[-3000:]
    one = json.loads([l for l in ref.stdout.splitlines() if l.startswith("{")][-1])
    assert abs(line["config"]["final_loss"] - one["config"]["final_loss"]) <= T.SPREAD_LOSS_REL * 10 * abs(one["config"]["final_loss"])
    # the defaults: 3 collectives per step (M_1's boundary rows, the batch rows, the weight gradients)
    r3 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--spinup-time", "0",
                         "--min-time", "0"], capture_output=True, text=True, env=_host_env(), timeout=900)
    assert r3.returncode == 0, r3.stderr[-3000:]
    l3 = json.loads([l for l in r3.stdout.splitlines() if l.startswith("{")][-1])
    assert l3["collectives_per_step"]["total"] == 3.0, l3["collectives_per_step"]
    assert abs(l3["config"]["final_loss"] - one["config"]["final_loss"]) <= T.SPREAD_LOSS_REL * 10 * abs(one["config"]["final_loss"])


def test_multi_process_job_with_a_dying_rank_ends_instead_of_hanging():
    """Three ranks as processes (host-staged backend); rank 1 disappears in the middle of training (tests/mp_fail_job.py).  The survivors'
    next collective must fail -- the transport reports the dead peer, the communicator turns it into GSS_ECOMM, the caller exits -- and
    the job ends with a non-zero code well inside the deadline instead of hanging."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "tests", "mp_fail_job.py")],
                       capture_output=True, text=True, env=_host_env(), timeout=600)
    took = time.time() - t0
    assert r.returncode != 0, "a job that lost a rank reported success"
    assert "step 2" in r.stdout and "step 5" not in r.stdout, r.stdout[-2000:]
    assert took < 300, f"the job took {took:.0f} s to notice a dead rank"


def test_host_staged_backend_collectives_single_rank():
    """gss_comm_create_host with one rank in this process: the three collectives are copies through the pinned staging buffers"""
    import torch.distributed as dist
    from gcn_drug_repurposing_amd.dist import host_comm
    import os
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    started = not dist.is_initialized()
    if started:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        comm = host_comm(1, 0)
        assert comm.count() == 1
        t = torch.randn(1000, device="cuda")
        ref = t.clone()
        comm.all_reduce_sum_(t)
        g = comm.allgather_bytes(ref)
        send = torch.randn(7, 16, device="cuda")
        recv = torch.zeros(7, 16, device="cuda")
        comm.exchange_rows(16, send, np.array([0, 0]), recv, np.array([0, 0]))      # the own range is empty by contract
        comm.sync(10)
        assert torch.equal(t, ref) and torch.equal(g, ref) and not bool(recv.any())
    finally:
        if started:
            dist.destroy_process_group()


def test_bench_sharded_path_over_rccl_single_rank(tmp_path):
    """bench.py's multi-GPU branch (the native sharded plan over an RCCL communicator) with one rank, launched the way the driver
    launches it; must agree with the single-GPU plan's loss after the same steps and carry the keys a scaling run is graded on."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSS_FORCE_SHARDED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
           "--spinup-time", "0", "--min-time", "0"]     # (both runs must take the SAME number of steps: no time-based spin-up / long run)
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    sharded = json.loads(line)
    assert "node-range shards" in sharded["config"]["parallelism"]
    assert sharded["rccl_ranks"] == 1 and len(sharded["per_rank"]["ms_per_step"]) == 1
    for key in ("roofline", "xgmi", "comm_share", "kernel_ms_per_step", "value_executed", "event_overhead_us_per_launch"):
        assert key in sharded, key
    assert sharded["roofline"]["bound"] == "hbm" and 0 < sharded["roofline"]["frac"] < 1
    ref = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--spinup-time", "0", "--min-time", "0"],
                         capture_output=True, text=True, timeout=600)
    assert ref.returncode == 0, ref.stderr[-2000:]
    single = json.loads([l for l in ref.stdout.splitlines() if l.startswith("{")][-1])
    assert abs(sharded["config"]["final_loss"] - single["config"]["final_loss"]) < 10 * T.TRAJ_LOSS_RTOL * abs(single["config"]["final_loss"])


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
    for mode, env_extra in (("single", {}), ("sharded", {"GSS_FORCE_SHARDED": "1"})):
        out = tmp_path / f"{mode}.txt"
        cmd = [sys.executable, os.path.join(root, "train.py"), "--emb-file", str(emb_path), "--num-layers", "2", "--hidden-units", "16",
               "--k", "5", "--epochs", "3", "--lr", "0.0003", "--beta-percentile", "98", "--batch-size", "64", "--seed", "7", "--out", str(out)]
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **env_extra), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.loadtxt(str(out)))
    assert np.abs(outs[0] - outs[1]).max() < 1e-5
    ref = np.loadtxt(bytes(g["graph_embs_txt"]).decode().splitlines())
    assert np.abs(outs[1] - ref).max() < T.TRAJ_CLI_EMB_ABS


def test_trainer_checkpoint_resume_on_the_sharded_path(tmp_path):
    """--checkpoint / --resume through the native sharded plan (one RCCL rank): 4 epochs in one run == 2 epochs + resume, byte for
    byte, and equal to the single-GPU trainer's file"""
    import os
    import subprocess
    import sys
    from gcn_drug_repurposing_amd import embio
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.RandomState(3)
    n, d = 500, 32
    emb = tmp_path / "in.embs.txt"
    embio.write_embs(str(emb), [f"n{i}" for i in range(n)], rng.randn(n, d) / 6)
    common = [sys.executable, os.path.join(root, "train.py"), "--emb-file", str(emb), "--num-layers", "2", "--hidden-units", str(d), "--k", "5",
              "--lr", "0.001", "--beta-percentile", "95", "--batch-size", "128", "--seed", "11"]
    env = dict(os.environ, GSS_FORCE_SHARDED="1")

    def run(extra, env_):
        r = subprocess.run(common + extra, capture_output=True, text=True, env=env_, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]

    ck = tmp_path / "ck.npz"
    run(["--epochs", "4", "--out", str(tmp_path / "a.txt")], env)
    run(["--epochs", "2", "--out", str(tmp_path / "b2.txt"), "--checkpoint", str(ck)], env)
    run(["--epochs", "4", "--out", str(tmp_path / "b.txt"), "--resume", str(ck)], env)
    run(["--epochs", "4", "--out", str(tmp_path / "s.txt")], dict(os.environ))
    assert (tmp_path / "a.txt").read_bytes() == (tmp_path / "b.txt").read_bytes()
    assert (tmp_path / "a.txt").read_bytes() == (tmp_path / "s.txt").read_bytes()       # world = 1 shard == plain plan, bit for bit


@pytest.mark.parametrize("world,case,recompute,slab", [(2, "edge_n600_d128_L2", -1, -1), (3, "edge_n600_d128_L2", 0, -1), (3, "knn_n2000_d64_L3", -1, -1),
                                                       (2, "knn_n200_d16_L2", -1, -1), (3, "edge_n600_d128_L2", -1, 1), (4, "knn_n2000_d64_L3", -1, 1),
                                                       (8, "knn_n200_d16_L2", -1, 1)])
def test_sharded_step_enqueues_the_announced_collectives(world, case, recompute, slab):
    """Round 4: a sharded step at L layers is 2L - 3 exchanges of A_hat's boundary rows (2L - 2 without
    halo_recompute: layer 2's boundary input rows are recomputed from layer 1's constant AX / AM, fetched once) + 2L - 3 of A_hat^T's,
    ONE batch-row all-reduce ([E_B | P_B | inv_B]; the [2B][d] input gradients are computed on every rank by the loss kernel's tail
    -- widths the tail does not cover, d = 16 here, keep the second all-reduce) and one weight-gradient all-reduce: 4 collectives at
    L = 2 where round 3 had 6 -- and 3 with A_hat's shard transposed in place (gss_shard_desc.a_loc_t, what build_shard passes by default): the
    last backward hop then multiplies this shard's rows of u into own + boundary rows and needs no exchange.  gss_plan_comm_stats counts what the plan enqueued; the first step's embeddings and loss still equal
    the single-GPU plan's bit for bit, the later ones to the trajectory tolerances.  slab = 1: the row-slab form of the loss sweep
    (knob loss_slab, automatic from B = 8192: rank r sweeps the i tiles r, r + P, ...; the ranks' rows of dE and shares of the loss
    are summed by one more all-reduce) -- more ranks than i tiles included; its loss agrees to rounding."""
    import gcn_drug_repurposing_amd as pkg
    from gcn_drug_repurposing_amd.dist import local_comms, sharded_plan_engine
    from gcn_drug_repurposing_amd.engine import GssEngine
    from gcn_drug_repurposing_amd.graph import GssGraph
    g = load_golden(case)
    n, d, L = (int(v) for v in g["meta"])
    adj, X, p0 = golden_csr(g, "A"), g["X"], golden_params(g, "init")
    batches = golden_batches(g)
    kw = dict(num_layers=L, layer_decay=float(g["decay"]), alpha=float(g["alpha"]), lr=float(g["lr"]))
    ref = GssEngine(GssGraph(adj, need_transpose=L > 1), torch.from_numpy(X).cuda(),
                    [torch.from_numpy(p0[k].copy()).cuda() for k in ("W1", "b1", "W2", "b2")], **kw)
    ref_out = []
    for k, idx in enumerate(batches):
        ref.step(torch.from_numpy(idx.astype(np.int32)).cuda(), float(g["beta"]))
        if k == 2:
            ref.forward()                # (the shards run a lazy step there and recompute all rows afterwards, with the updated weights)
        ref_out.append((ref.loss.item(), ref.emb.cpu().numpy().copy()))
    lib = pkg.load()
    comms = local_comms(world)
    results, errors = [None] * world, []
    gate = threading.Barrier(world)

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                eng = sharded_plan_engine(adj, X, p0, comms[rank], device=torch.device("cuda:0"), **kw)   # (snapshots the knob)
                gate.wait(60)
                out = []
                for k, idx in enumerate(batches):
                    t = torch.from_numpy(idx.astype(np.int32)).cuda()
                    (eng.step_lazy if k == 2 else eng.step)(t, float(g["beta"]))
                    stats = eng.comm_stats()
                    eng.check_guards()
                    if k == 2:
                        eng.forward()            # a lazy step leaves the embeddings valid on the batch rows only
                        eng.comm_stats()
                    out.append((eng.loss.item(), eng.gather_embeddings().cpu().numpy(), stats))
                results[rank] = out
        except Exception as e:  # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))
            comms[rank].abort()
            gate.abort()

    assert lib.gss_debug_set_option(b"halo_recompute", recompute) == 0 and lib.gss_debug_set_option(b"loss_slab", slab) == 0
    try:
        ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
        [t.start() for t in ts]
        [t.join(600) for t in ts]
    finally:
        lib.gss_debug_set_option(b"halo_recompute", -1)
        lib.gss_debug_set_option(b"loss_slab", -1)
    assert not errors, errors
    rec = 1 if (recompute != 0 and L > 1) else 0
    tail = d in (64, 128, 256)
    tloc = 1 if (rec and L >= 2) else 0          # the last backward hop on A_hat's shard transposed in place, no exchange of u
    steady = ((2 * L - 2 - rec) + max(0, 2 * L - 3) - tloc, (1 if (tail or L == 1) else 2) + (1 if slab == 1 else 0), 1)
    for res in results:
        for k, (loss, emb, stats) in enumerate(res):
            if k == 0:     # one forward: a row's result does not depend on the shard, nor on who computed a boundary row
                if slab == 1:      # the ranks' shares of the loss are summed in rank order
                    assert abs(loss - ref_out[k][0]) <= T.TRAJ_LOSS_RTOL * abs(ref_out[k][0]), (k, loss, ref_out[k][0])
                else:
                    assert loss == ref_out[k][0], (k, loss, ref_out[k][0])
                np.testing.assert_array_equal(emb, ref_out[k][1])
            else:          # the ranks' weight gradients are summed in rank order, the single GPU's slabs in slice order: rounding
                assert abs(loss - ref_out[k][0]) <= T.TRAJ_LOSS_RTOL * abs(ref_out[k][0]), (k, loss, ref_out[k][0])
                assert np.abs(emb - ref_out[k][1]).max() <= T.TRAJ_EMB_REL * np.abs(ref_out[k][1]).max(), k
            first = (steady[0] + (2 + 2 * rec if L > 1 else 1), steady[1], steady[2]) if k == 0 else steady
            assert stats == first, (k, stats, first)


def test_large_batches_sweep_the_loss_as_row_slabs_by_themselves():
    """knob loss_slab left at its default: from B = 8192 on a sharded plan sweeps the B x B loss as row slabs (rank r the i tiles r, r + P,
    ...; the ranks' rows of dE and shares of the loss summed by a second batch all-reduce), below it the sweep is replicated.  Config 2's
    graph at world 2, d = 64: B = 2048 (replicated), B = 8192 and the reference's default --batch-size 0, i.e. ONE batch of all N =
    29,960 nodes (1873 i tiles: 937 / 936 per rank) -- losses and embeddings against the single-GPU plan."""
    from gcn_drug_repurposing_amd.dist import local_comms, sharded_plan_engine
    from gcn_drug_repurposing_amd.engine import GssEngine
    from gcn_drug_repurposing_amd.graph import GssGraph
    from gcn_drug_repurposing_amd.synth import whole_graph_standin
    adj = whole_graph_standin(seed=1)[0]
    n, d, L, world = adj.shape[0], 64, 2, 2
    rng = np.random.RandomState(6)
    X = (rng.randn(n, d) / 4).astype(np.float32)
    w = (rng.randn(d, d) * 1e-2).astype(np.float32)
    np.fill_diagonal(w, 1.0)
    p0 = {"W1": w, "b1": np.zeros(d, np.float32), "W2": w.copy(), "b2": np.zeros(d, np.float32)}
    batches = [rng.permutation(n)[:b].astype(np.int32) for b in (2048, 8192, n, 8200)]
    kw = dict(num_layers=L, layer_decay=0.3, alpha=1.0, lr=1e-3, max_batch=n)
    ref = GssEngine(GssGraph(adj), torch.from_numpy(X).cuda(), [torch.from_numpy(p0[k].copy()).cuda() for k in ("W1", "b1", "W2", "b2")], **kw)
    ref_out = []
    for idx in batches:
        ref.step(torch.from_numpy(idx).cuda(), 0.2)
        ref_out.append((ref.loss.item(), ref.emb.cpu().numpy().copy()))
    comms = local_comms(world)
    results, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                eng = sharded_plan_engine(adj, X, p0, comms[rank], device=torch.device("cuda:0"), **kw)
                out = []
                for idx in batches:
                    eng.comm_stats()
                    eng.step(torch.from_numpy(idx).cuda(), 0.2)
                    stats = eng.comm_stats()
                    eng.check_guards()
                    out.append((eng.loss.item(), eng.gather_embeddings().cpu().numpy(), stats))
                results[rank] = out
        except Exception as e:  # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))
            comms[rank].abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(900) for t in ts]
    assert not errors, errors
    for res in results:
        assert res is not None
        for k, (loss, emb, stats) in enumerate(res):
            assert stats[1] == (1 if len(batches[k]) < 8192 else 2), (k, stats)          # batch-row all-reduces: the slab adds one
            assert abs(loss - ref_out[k][0]) <= T.TRAJ_LOSS_RTOL * abs(ref_out[k][0]), (k, loss, ref_out[k][0])
            assert np.abs(emb - ref_out[k][1]).max() <= T.TRAJ_EMB_REL * np.abs(ref_out[k][1]).max(), k
    assert results[0][2][0] == results[1][2][0]                                             # replicas agree bit for bit
