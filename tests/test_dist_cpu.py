"""CPU: the PRODUCT's node-range sharding as separate processes (world_size 2 and 3 over gloo).

Each rank runs gcn_drug_repurposing_amd/shards.py build_shard (row source -> nnz-balanced ranges -> D^-1/2 all-gather -> Halo.exchange
-> operand-row CSRs, with and without the hub-first relabelling) and dist.ShardLayout exactly as on the GPU box; only the device side is
replaced (tests/cpu_ops.py: numpy kernels, a gloo communicator with dist.Comm's methods).  The step is tests/shard_step_mirror.py, a
restatement of the sharded step csrc/plan.hip enqueues, on those product objects: boundary rows packed by Halo.send_rows and moved
by exchange_rows(send_off, recv_off), the batch maps through gid2op_t / node_map, all-reduces of the batch rows and the gradients.
It must reproduce the reference-generated fixture trajectories, keep the replicated state identical on every rank, and exchange
exactly the rows the halo layout announces."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp

import tolerances as T
from conftest import golden_batches, golden_csr, golden_params, load_golden

torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, relabel, out_dir, slab=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from cpu_ops import GlooComm, NumpyShardOps
        from shard_step_mirror import ShardStepMirror
        from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_rows
        g = load_golden(case)
        n, d, L = (int(v) for v in g["meta"])
        comm = GlooComm()
        shard = build_shard(ScipySource(golden_csr(g, "A")), comm, need_transpose=L > 1, device="cpu", relabel=relabel, ops=NumpyShardOps())
        lay = shard.layout
        lo, hi = shard.part.rows(rank)
        # the shard's normalised values are the reference's preprocess_graph values (fp32 cast of helper.py:95), entry for entry
        ahat = golden_csr(g, "Ahat")
        ahat.sort_indices()
        rows = np.arange(lo, hi) if shard.relabel is None else shard.relabel.perm[lo:hi]
        ref = sp.csr_matrix(ahat[rows])
        ref.sort_indices()
        np.testing.assert_array_equal(np.asarray(shard.a.m.data, np.float32), ref.data.astype(np.float32))
        # the same matrices once more, split by column for the overlapped hops: own-column + boundary-column entries = the matrix,
        # every row's entries in their original order
        split = build_shard(ScipySource(golden_csr(g, "A")), comm, need_transpose=L > 1, device="cpu", relabel=relabel, ops=NumpyShardOps(), split=True)
        for full, halves in ((split.a, split.split_a), (split.at, split.split_at)):
            if full is None:
                continue
            assert (halves is not None) == (world > 1)
            if halves is None:
                continue
            own, hal = halves
            assert own.nnz + hal.nnz == full.nnz and abs((own.m + hal.m) - full.m).max() == 0
            assert (own.m.indices < (hi - lo)).all() and (hal.m.indices >= (hi - lo)).all()
            for r in range(hi - lo):
                both = np.concatenate([own.m.indices[own.m.indptr[r]:own.m.indptr[r + 1]], hal.m.indices[hal.m.indptr[r]:hal.m.indptr[r + 1]]])
                assert sorted(both.tolist()) == sorted(full.m.indices[full.m.indptr[r]:full.m.indptr[r + 1]].tolist())
        eng = ShardStepMirror(shard, shard_rows(shard, g["X"]), golden_params(g, "init"), comm, num_layers=L, layer_decay=float(g["decay"]),
                              alpha=float(g["alpha"]), lr=float(g["lr"]), slab=slab)
        losses, ex_per_step, rows_per_step, ar_per_step = [], [], [], []
        for idx in golden_batches(g):
            e0, r0, a0 = comm.exchanges, comm.rows_received, comm.allreduces
            eng.step(idx, float(g["beta"]))
            losses.append(float(eng.loss.item()))
            ex_per_step.append(comm.exchanges - e0)
            rows_per_step.append(comm.rows_received - r0)
            ar_per_step.append(comm.allreduces - a0)
        emb = eng.gather_embeddings()
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), losses=np.array(losses), emb=emb, lo=lo, hi=hi, ex=np.array(ex_per_step),
                 rows=np.array(rows_per_step), ar=np.array(ar_per_step), halo_a=lay.halo_a.n_halo, halo_t=lay.halo_at.n_halo if lay.halo_at is not None else 0,
                 send_a=int(lay.halo_a.send_off[-1]), send_t=int(lay.halo_at.send_off[-1]) if lay.halo_at is not None else 0,
                 **{k: p.numpy() for k, p in zip(("W1", "b1", "W2", "b2"), eng.params)})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case,relabel,slab", [(2, "edge_n600_d128_L2", False, False), (2, "edge_n600_d128_L2", True, False),
                                                     (3, "knn_n200_d16_L2", False, False), (2, "knn_n2000_d64_L3", True, False),
                                                     (3, "toy_sif_d64_L2", False, False), (3, "edge_n600_d128_L2", True, True)])
def test_product_shards_as_gloo_processes_match_reference_trajectory(tmp_path, world, case, relabel, slab):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), case, relabel, str(tmp_path), slab), nprocs=world, join=True)
    g = load_golden(case)
    n, d, L = (int(v) for v in g["meta"])
    outs = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert outs[0]["lo"] == 0 and outs[-1]["hi"] == n
    for r in range(1, world):
        assert outs[r]["lo"] == outs[r - 1]["hi"]
        for k in ("losses", "emb", "W1", "b1", "W2", "b2"):     # replicated state stays identical across ranks
            np.testing.assert_array_equal(outs[r][k], outs[0][k])
    # the mirror computes every op in fp64 and rounds to fp32 between ops: inside the spread between fp32 and fp64 executions of the
    # reference's ops that tests/tolerances.py records
    np.testing.assert_allclose(outs[0]["losses"], g["losses"], rtol=T.TRAJ_LOSS_RTOL, atol=1e-9)
    assert np.abs(outs[0]["emb"] - g["emb_last"]).max() / np.abs(g["emb_last"]).max() < T.TRAJ_EMB_REL
    for k in ("W1", "b1", "W2", "b2"):
        assert np.abs(outs[0][k] - g["final_" + k]).max() < T.TRAJ_WEIGHT_LR * float(g["lr"])
    # what the step moves is what the halo layout announces (include/gssgcn.h, gss_plan_create_sharded): per step 2L - 2 exchanges of
    # A_hat's halo -- one less with halo_recompute (these graphs: layer 2's boundary input rows are computed, not fetched) -- + 2L - 3
    # of A_hat^T's, less the last one (gss_shard_desc.a_loc_t); the first step also fetches the constant boundary rows of X_0 and M_0 and, with halo_recompute, of AX_0 and AM_0
    rec = 1 if L > 1 else 0
    hops_a = 2 * L - 2 - rec
    hops_t = max(0, 2 * L - 3) - (1 if L >= 2 else 0)      # the last backward hop runs on A_hat's shard transposed in place: no exchange
    steady = hops_a + hops_t
    for o in outs:
        assert o["ex"][0] == steady + 2 + 2 * rec and all(int(e) == steady for e in o["ex"][1:])
        assert int(o["rows"][1]) == hops_a * int(o["halo_a"]) + hops_t * int(o["halo_t"])
        # all-reduces per step: [E_B | P_B | inv_B], the four weight gradients (+ the ranks' rows of dE with the row-slab loss sweep)
        assert all(int(a) == (3 if slab else 2) for a in o["ar"])
    # every row a shard receives is a row a peer sends
    assert sum(int(o["halo_a"]) for o in outs) == sum(int(o["send_a"]) for o in outs)
    assert sum(int(o["halo_t"]) for o in outs) == sum(int(o["send_t"]) for o in outs)


def test_partition_ranges_balance_stored_entries():
    from gcn_drug_repurposing_amd.dist import Partition, nnz_balanced_ranges
    rng = np.random.RandomState(0)
    a = sp.random(500, 500, density=0.02, random_state=rng, format="csr")
    a[7, :] = 1.0   # a hub row
    a = sp.csr_matrix(a)
    b = nnz_balanced_ranges(a.indptr, 4)
    assert b[0] == 0 and b[-1] == 500 and np.all(np.diff(b) >= 0)
    per = [a.indptr[b[i + 1]] - a.indptr[b[i]] for i in range(4)]
    assert max(per) < 2.2 * (a.nnz / 4)
    part = Partition(b)
    ids = np.arange(500)
    own = part.owner(ids)
    assert np.all((ids >= part.bounds[own]) & (ids < part.bounds[own + 1]))
    assert [part.rows(r) for r in range(4)] == [(int(b[r]), int(b[r + 1])) for r in range(4)]


def test_halo_layout_reproduces_the_global_product():
    """Halo (dist.py): own rows first, then the referenced rows of other shards grouped by owner; a shard's CSR with
    operand-row column ids times the assembled operand equals its rows of the global product.  The send side is derived
    here the way Halo.exchange derives it on the device: every rank's request list, regrouped by owner."""
    from gcn_drug_repurposing_amd.dist import Halo, Partition, nnz_balanced_ranges
    rng = np.random.RandomState(1)
    n, P, d = 400, 4, 3
    a = sp.random(n, n, density=0.01, random_state=rng, format="csr") + sp.eye(n, format="csr")
    a = sp.csr_matrix(a)
    a[5, :] = 0.5                      # a hub row that references every node
    a = sp.csr_matrix(a)
    part = Partition(nnz_balanced_ranges(a.indptr, P))
    x = rng.randn(n, d)
    halos = []
    for r in range(P):
        lo, hi = part.rows(r)
        halos.append(Halo(a[lo:hi].indices, part, r))
    for r, h in enumerate(halos):
        lo, hi = part.rows(r)
        assert h.recv_off[0] == 0 and h.recv_off[-1] == h.n_halo and h.recv_off[r + 1] == h.recv_off[r]      # nothing from itself
        assert np.all(np.diff(h.remote) > 0) and not np.any((h.remote >= lo) & (h.remote < hi))
        for q in range(P):                                                 # the block of owner q holds rows of q only
            blk = h.remote[h.recv_off[q]:h.recv_off[q + 1]]
            assert np.all((blk >= part.bounds[q]) & (blk < part.bounds[q + 1]))
        # what the peers would pack for this shard: rows (ids - their lo) of their local x, in this shard's request order
        operand = np.concatenate([x[lo:hi]] + [x[part.bounds[q]:part.bounds[q + 1]][h.remote[h.recv_off[q]:h.recv_off[q + 1]] - part.bounds[q]]
                                               for q in range(P)])
        sub = a[lo:hi]
        loc = sp.csr_matrix((sub.data, h.local_cols(sub.indices), sub.indptr), shape=(hi - lo, (hi - lo) + h.n_halo))
        np.testing.assert_allclose(loc @ operand, (a @ x)[lo:hi], rtol=1e-12, atol=1e-12)
        assert np.array_equal(h.gid2op[lo:hi], np.arange(hi - lo)) and (h.gid2op >= 0).sum() == (hi - lo) + h.n_halo
    # the hub's shard reads every node; shards that share no edge exchange nothing
    hub_rank = int(part.owner(np.array([5]))[0])
    assert halos[hub_rank].n_halo == n - halos[hub_rank].nl


def _host_transport_worker(rank, world, port):
    """the transport callback dist.host_comm hands to gss_comm_create_host, called directly on host buffers (no GPU: the C side only
    stages device memory through such buffers)"""
    import ctypes as C
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gcn_drug_repurposing_amd.dist import host_comm
        comm = host_comm()
        assert (comm.world, comm.rank) == (world, rank)
        cb = comm._keep
        i64 = C.POINTER(C.c_int64)
        # kind 0: all-gather of `count` bytes
        count = 37
        send = (C.c_uint8 * count)(*[(rank * 50 + k) % 251 for k in range(count)])
        recv = (C.c_uint8 * (count * world))()
        assert cb(None, 0, C.addressof(send), i64(), C.addressof(recv), i64(), count) == 0
        got = np.frombuffer(recv, dtype=np.uint8).reshape(world, count)
        for r in range(world):
            np.testing.assert_array_equal(got[r], [(r * 50 + k) % 251 for k in range(count)])
        assert cb(None, 0, C.addressof(send), i64(), C.addressof(recv), i64(), 0) == 0          # nothing to move
        # kind 1: all-to-all-v with byte offsets; rank r sends 3 * (r + q + 1) bytes to q != r, nothing to itself; an empty pair as well
        def n_bytes(src, dst):
            return 0 if src == dst or (src, dst) == (0, world - 1) else 3 * (src + dst + 1)
        soff = np.zeros(world + 1, np.int64)
        roff = np.zeros(world + 1, np.int64)
        for q in range(world):
            soff[q + 1] = soff[q] + n_bytes(rank, q)
            roff[q + 1] = roff[q] + n_bytes(q, rank)
        sbuf = np.zeros(max(int(soff[-1]), 1), np.uint8)
        for q in range(world):
            sbuf[soff[q]:soff[q + 1]] = (100 * rank + 10 * q + np.arange(n_bytes(rank, q))) % 256
        rbuf = np.full(max(int(roff[-1]), 1), 255, np.uint8)
        assert cb(None, 1, sbuf.ctypes.data, soff.ctypes.data_as(i64), rbuf.ctypes.data, roff.ctypes.data_as(i64), 0) == 0
        for q in range(world):
            np.testing.assert_array_equal(rbuf[roff[q]:roff[q + 1]], (100 * q + 10 * rank + np.arange(n_bytes(q, rank))) % 256)
        assert cb(None, 9, sbuf.ctypes.data, soff.ctypes.data_as(i64), rbuf.ctypes.data, roff.ctypes.data_as(i64), 0) != 0   # unknown kind
        assert comm.count() == world
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_host_staged_transport_callback_over_gloo(world):
    """GSS_COMM_BACKEND=host (include/gssgcn.h gss_comm_create_host): the byte transport underneath the host-staged communicator, as
    separate processes on CPU -- all-gather and all-to-all-v land every rank's bytes where the C side expects them"""
    import torch.multiprocessing as mp
    mp.spawn(_host_transport_worker, args=(world, _free_port()), nprocs=world, join=True)


@pytest.mark.parametrize("relabel", [False, True])
def test_edgelist_source_rows_equal_the_scipy_path(relabel):
    """shards.EdgelistSource is train.py's --adj-file path for sharded runs and every graph of >= RELABEL_MIN_NODES nodes.
    Its rows of A + I and of (A + I)^T must be the rows ScipySource(edgelist_adj(...)) has -- the path it replaced -- on a directed edgelist
    with repeated (u, v) (last weight wins), self loops (added to the diagonal's 1), zero weights and a self loop of weight -1 (both
    dropped, as scipy's `adj + eye` drops them): same stored entries, same nnz(A_hat), same row work, over several row windows."""
    from gcn_drug_repurposing_amd.graph import edgelist_adj
    from gcn_drug_repurposing_amd.shards import EdgelistSource, Relabel, ScipySource
    rng = np.random.RandomState(5)
    n, m = 120, 900
    src, dst = rng.randint(0, n, m), rng.randint(0, n, m)
    w = rng.uniform(0.1, 2.0, m)
    src[:40], dst[:40] = src[40:80], dst[40:80]                     # repeated (u, v): the later line wins
    w[100:130] = 0.0                                                # zero weights
    src[200:215] = dst[200:215]                                     # self loops ...
    w[200:205] = -1.0                                               # ... five of them cancel the diagonal's 1
    ref = ScipySource(edgelist_adj(src, dst, w, n))
    got = EdgelistSource(src, dst, w, n, device="cpu")
    assert got.nnz == ref.nnz and ref.nnz < n + len(set(zip(src.tolist(), dst.tolist())))
    np.testing.assert_array_equal(got.work(None, "cpu"), ref.work(None, "cpu"))
    rl = Relabel(ref.work(None, "cpu")) if relabel else None
    for lo, hi in ((0, n), (0, 37), (37, 90), (90, n), (50, 50)):
        for name in ("rows", "rows_t"):
            a = getattr(got, name)(lo, hi, "cpu", relabel=rl)
            b = getattr(ref, name)(lo, hi, "cpu", relabel=rl)
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x.numpy(), y.numpy(), err_msg=f"{name} [{lo}, {hi})")


def _bad_rowsum_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        import warnings
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from cpu_ops import GlooComm, NumpyShardOps
        from gcn_drug_repurposing_amd.graph import NonPositiveRowSum
        from gcn_drug_repurposing_amd.shards import ScipySource, build_shard
        g = load_golden("knn_negative_rowsum_n96_d8")
        n = g["X"].shape[0]
        adj = sp.csr_matrix((g["A_data"], g["A_indices"], g["A_indptr"]), shape=(n, n))
        msg = ""
        try:
            build_shard(ScipySource(adj), GlooComm(), device="cpu", relabel=bool(rank >= 0 and world == 3), ops=NumpyShardOps(), name_of=lambda i: f"n{i}")
        except NonPositiveRowSum as e:
            msg = f"{e.count} {e.first} {e.args[0]}"
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            shard = build_shard(ScipySource(adj), GlooComm(), device="cpu", relabel=False, ops=NumpyShardOps(), allow_nan=True)
        lo, hi = shard.part.rows(rank)
        ahat = sp.csr_matrix((g["Ahat_data"], g["Ahat_indices"], g["Ahat_indptr"]), shape=(n, n))[lo:hi]
        same_nan = bool(np.array_equal(np.isnan(np.asarray(shard.a.m.data)), np.isnan(ahat.data)))
        with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as f:
            f.write(f"{msg}\n{len(w)} {same_nan}\n")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_non_positive_row_sum_is_refused_on_every_rank(tmp_path, world):
    """the guard helper.py:85 lacks, on shards: the rank that owns the bad row finds it, EVERY rank raises the same NonPositiveRowSum
    (naming the node by its original id, relabelled or not) -- no rank is left waiting in a collective; under allow_nan every rank
    warns and the shard's values have the reference's NaNs"""
    import torch.multiprocessing as mp
    mp.spawn(_bad_rowsum_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = load_golden("knn_negative_rowsum_n96_d8")
    bad = int(g["bad_rows"][0])
    lines = [open(tmp_path / f"r{r}.txt").read().splitlines() for r in range(world)]
    for l in lines:
        assert l[0].startswith(f"1 {bad} 1 of 96 rows") and f"node {bad} = 'n{bad}'" in l[0], l
        assert l[1] == "1 True", l
