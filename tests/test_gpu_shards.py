"""-m gpu: building shards without the whole graph on any rank (gcn_drug_repurposing_amd/shards.py) -- device normalisation
of a shard is bit-identical to the single-GPU A_hat (helpers/helper.py:82-95 semantics), the RMAT row source gives the same
graph whatever the number of ranks, and a sharded step on it equals the single-GPU plan."""
import threading

import numpy as np
import pytest
import scipy.sparse as sp

import tolerances as T
from conftest import golden_csr, load_golden
from oracle import gss_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _threaded(world, fn, comms=None):
    out, errors = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                out[rank] = fn(rank)
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))
            if comms is not None:
                comms[rank].abort()           # release the peers now instead of after the barrier timeout

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(900) for t in ts]
    assert not errors, errors
    assert all(o is not None for o in out), "a rank thread did not finish"
    return out


def _assemble(out, n):
    """per-rank (lo, indptr, global cols, vals) -> scipy CSR [n, n]"""
    rows, cols, vals = [], [], []
    for lo, ip, gc, v in out:
        cnt = np.diff(ip)
        rows.append(np.repeat(np.arange(lo, lo + len(cnt)), cnt))
        cols.append(gc)
        vals.append(v)
    m = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    m.sort_indices()
    return m


def _shard_dump(shard, rank, which="a"):
    csr = shard.a if which == "a" else shard.at
    halo = shard.layout.halo_a if which == "a" else shard.layout.halo_at
    lo, hi = shard.part.rows(rank)
    op2gid = np.concatenate([np.arange(lo, hi), halo.remote]).astype(np.int64)
    col = csr.col.cpu().numpy()[:csr.nnz]
    return lo, csr.h_indptr.copy(), op2gid[col], csr.val.cpu().numpy()[:csr.nnz].copy()


@pytest.mark.parametrize("world", [1, 3])
def test_device_normalised_shards_are_bit_identical_to_single_gpu_a_hat(world):
    from gcn_drug_repurposing_amd.dist import local_comms
    from gcn_drug_repurposing_amd.graph import GssGraph
    from gcn_drug_repurposing_amd.shards import ScipySource, build_shard
    g = load_golden("edge_n600_d128_L2")           # an asymmetric weighted edgelist adjacency
    adj = golden_csr(g, "A")
    n = adj.shape[0]
    ref = GssGraph(adj)
    comms = local_comms(world)

    def fn(rank):
        shard = build_shard(ScipySource(adj), comms[rank], need_transpose=True, device="cuda:0")
        return _shard_dump(shard, rank, "a"), _shard_dump(shard, rank, "at"), shard.rowsum.cpu().numpy(), shard.layout.halo_fraction()

    out = _threaded(world, fn, comms)
    a = _assemble([o[0] for o in out], n)
    at = _assemble([o[1] for o in out], n)
    ra, rat = ref.a.to_scipy(), ref.at.to_scipy()
    ra.sort_indices(), rat.sort_indices()
    for got, want in ((a, ra), (at, rat)):
        assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
        assert np.array_equal(got.data, want.data)                                   # the same bits as gss_normalize_adj
    assert np.array_equal(np.concatenate([o[2] for o in out]), ref.rowsum.cpu().numpy())
    a_hat, _ = O.preprocess_graph(adj)                                               # and the reference's values to 1 ulp
    assert np.abs(a.toarray() - a_hat.toarray().astype(np.float32)).max() <= np.spacing(np.float32(np.abs(a.data).max()))
    if world > 1:
        assert all(0.0 < o[3][0] <= 1.0 for o in out)


def test_rmat_source_is_the_same_graph_on_any_number_of_ranks_and_trains_like_one_gpu():
    from gcn_drug_repurposing_amd.dist import local_comms
    from gcn_drug_repurposing_amd.engine import GssEngine
    from gcn_drug_repurposing_amd.graph import GssGraph
    from gcn_drug_repurposing_amd.shards import RmatSource, build_shard, gaussian_rows, shard_engine
    n, m, d, L, B = 6000, 70000, 64, 2, 512
    np.random.seed(3)
    p = O.init_layer_weights(d, 1e-2)
    idx = np.random.RandomState(2).permutation(n)[:B].astype(np.int32)
    graphs, runs = {}, {}
    for world in (1, 3):
        comms = local_comms(world)

        def fn(rank):
            src = RmatSource(n, m, seed=4, device="cuda:0", chunk=1 << 15)        # several rounds
            shard = build_shard(src, comms[rank], need_transpose=True, device="cuda:0")
            lo, hi = shard.part.rows(rank)
            eng = shard_engine(shard, gaussian_rows(lo, hi, d, 5), p, comms[rank], num_layers=L, layer_decay=0.3, alpha=1.0, lr=1e-3, max_batch=B)
            t = torch.from_numpy(idx).cuda()
            eng.forward()
            eng.loss_backward(t, 0.25)
            res = dict(dump=_shard_dump(shard, rank, "a"), dump_t=_shard_dump(shard, rank, "at"), nnz=shard.nnz_global, rounds=src.rounds,
                       emb=eng.gather_embeddings().cpu().numpy(), loss=eng.loss.item(), grads=[g_.cpu().numpy() for g_ in eng.grads])
            return res

        out = _threaded(world, fn, comms)
        graphs[world] = (_assemble([o["dump"] for o in out], n), _assemble([o["dump_t"] for o in out], n))
        runs[world] = out[0]
        assert all(o["nnz"] == m + n for o in out) and out[0]["rounds"] >= 2
    a1, at1 = graphs[1]
    assert a1.nnz == m + n                                                       # exactly m entries + n self loops
    assert (a1 != at1.T).nnz == 0                                                # the transposed shards are the transpose
    for k in (0, 1):
        assert np.array_equal(graphs[3][k].indptr, graphs[1][k].indptr) and np.array_equal(graphs[3][k].indices, graphs[1][k].indices)
        assert np.array_equal(graphs[3][k].data, graphs[1][k].data)
    assert a1.diagonal().min() > 0
    deg = np.diff(a1.indptr)
    assert deg.max() > 20 * np.median(deg)                                       # RMAT skew
    # the sharded step equals the one-rank step bit for bit in everything row-wise, and the plain single-GPU plan on the
    # assembled matrix
    np.testing.assert_array_equal(runs[3]["emb"], runs[1]["emb"])
    assert runs[3]["loss"] == runs[1]["loss"]
    for a, b in zip(runs[3]["grads"], runs[1]["grads"]):
        assert np.abs(a - b).max() < 1e-5 * np.abs(b).max() + 1e-12
    graph = GssGraph.from_normalized(a1)
    params = [torch.from_numpy(p[k].copy()).cuda() for k in ("W1", "b1", "W2", "b2")]
    eng = GssEngine(graph, torch.from_numpy(gaussian_rows(0, n, d, 5)).cuda(), params, num_layers=L, layer_decay=0.3, alpha=1.0, lr=1e-3, max_batch=B)
    eng.forward()
    eng.loss_backward(torch.from_numpy(idx).cuda(), 0.25)
    np.testing.assert_array_equal(eng.emb.cpu().numpy(), runs[1]["emb"])
    assert eng.loss.item() == runs[1]["loss"]


@pytest.mark.parametrize("world", [1, 3])
def test_hub_first_relabelling_is_invisible_in_the_results(world):
    """build_shard(relabel=True): nodes renumbered by descending degree (a gather-locality measure for graphs far larger than
    the caches).  Rows keep the original order of their entries, batches keep naming original ids and the embeddings come back
    in original order -> embeddings and loss are BIT-identical to the un-relabelled single-GPU plan, gradients equal up to the
    order in which rows are summed."""
    from gcn_drug_repurposing_amd.dist import local_comms
    from gcn_drug_repurposing_amd.engine import GssEngine
    from gcn_drug_repurposing_amd.graph import GssGraph
    from gcn_drug_repurposing_amd.shards import RmatSource, ScipySource, build_shard, gaussian_rows, shard_engine, shard_rows
    from conftest import golden_batches, golden_params
    g = load_golden("knn_n2000_d64_L3")
    n, d, L = (int(v) for v in g["meta"])
    adj, X, p0 = golden_csr(g, "A"), g["X"], golden_params(g, "init")
    batches = golden_batches(g)
    kw = dict(num_layers=L, layer_decay=float(g["decay"]), alpha=float(g["alpha"]), lr=float(g["lr"]))
    ref = GssEngine(GssGraph(adj), torch.from_numpy(X).cuda(), [torch.from_numpy(p0[k].copy()).cuda() for k in ("W1", "b1", "W2", "b2")], **kw)
    ref.forward()
    ref.loss_backward(torch.from_numpy(batches[0].astype(np.int32)).cuda(), float(g["beta"]))
    ref_emb, ref_loss, ref_grads = ref.emb.cpu().numpy().copy(), ref.loss.item(), [t.cpu().numpy().copy() for t in ref.grads]
    comms = local_comms(world)

    def fn(rank):
        shard = build_shard(ScipySource(adj), comms[rank], need_transpose=True, device="cuda:0", relabel=True)
        assert shard.relabel is not None and shard.node_map is not None
        eng = shard_engine(shard, shard_rows(shard, X), p0, comms[rank], **kw)
        t = torch.from_numpy(batches[0].astype(np.int32)).cuda()
        eng.forward()
        eng.loss_backward(t, float(g["beta"]))
        res = dict(emb=eng.gather_embeddings().cpu().numpy(), loss=eng.loss.item(), grads=[x.cpu().numpy() for x in eng.grads],
                   beta=eng.percentile(float(g["beta_pct"])), perm=shard.relabel.perm.copy())
        eng.adam()
        losses = [res["loss"]]
        for idx in batches[1:]:
            eng.step(torch.from_numpy(idx.astype(np.int32)).cuda(), float(g["beta"]))
            losses.append(eng.loss.item())
        res["losses"] = losses
        return res

    out = _threaded(world, fn, comms)
    deg = np.diff(sp.csr_matrix(adj).indptr) + np.diff(sp.csr_matrix(adj.T).indptr)
    assert np.all(np.diff(deg[out[0]["perm"]]) <= 0)                       # hubs first
    for o in out:
        np.testing.assert_array_equal(o["emb"], ref_emb)
        assert o["loss"] == ref_loss
        assert abs(o["beta"] - float(g["beta"])) < 2e-6
        for a, b in zip(o["grads"], ref_grads):
            assert np.abs(a - b).max() < 1e-5 * np.abs(b).max() + 1e-12
    np.testing.assert_allclose(out[0]["losses"], g["losses"], rtol=T.TRAJ_LOSS_RTOL, atol=1e-9)
    # the RMAT source relabelled = the RMAT source as generated
    n2, m2, d2 = 5000, 50000, 32
    np.random.seed(1)
    p2 = O.init_layer_weights(d2, 1e-2)
    idx = np.random.RandomState(3).permutation(n2)[:300].astype(np.int32)
    embs = []
    for rl in (False, True):
        c1 = local_comms(1)[0]
        shard = build_shard(RmatSource(n2, m2, seed=4, device="cuda:0", chunk=1 << 14), c1, need_transpose=True, device="cuda:0", relabel=rl)
        x_all = gaussian_rows(0, n2, d2, 5)
        eng = shard_engine(shard, shard_rows(shard, x_all), p2, c1, num_layers=2, lr=1e-3, max_batch=300)
        eng.forward()
        eng.loss_backward(torch.from_numpy(idx).cuda(), 0.25)
        embs.append((eng.gather_embeddings().cpu().numpy(), eng.loss.item(), shard.nnz_global))
    np.testing.assert_array_equal(embs[0][0], embs[1][0])
    assert embs[0][1] == embs[1][1] and embs[0][2] == embs[1][2] == m2 + n2


@pytest.mark.parametrize("world,L", [(1, 2), (1, 3), (3, 2)])
def test_nonzero_row_bitmaps_of_the_sparse_backward_change_nothing(world, L):
    """Plans over huge operands (>= sparse_bits_rows rows, 100k by default) keep two bitmaps: batch membership in front of the position map of
    the top layer's sparse backward hop, and -- written by that hop -- the rows of its result u that can be non-zero, which the hop
    after it follows exclusively (B << N: nearly all rows of u are zero).  Forced on here at test size (knob) with a small batch, so
    that most rows ARE skipped: parameters, losses and embeddings after 4 steps equal the plan without the bitmaps bit for bit."""
    import gcn_drug_repurposing_amd as pkg
    from gcn_drug_repurposing_amd.dist import local_comms
    from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_engine, shard_rows
    from conftest import golden_params
    lib = pkg.load()
    g = load_golden("knn_n2000_d64_L3")
    n, d, _ = (int(v) for v in g["meta"])
    adj, X, p0 = golden_csr(g, "A"), g["X"], golden_params(g, "init")
    rng = np.random.RandomState(5)
    batches = [rng.choice(n, size=s, replace=False).astype(np.int32) for s in (40, 7, 64, 1)]
    kw = dict(num_layers=L, layer_decay=float(g["decay"]), alpha=float(g["alpha"]), lr=float(g["lr"]), max_batch=64)

    def run(bits_rows):
        assert lib.gss_debug_set_option(b"sparse_bits_rows", bits_rows) == 0
        assert lib.gss_debug_set_option(b"spmm_list_blocks", 1 if bits_rows == 1 else 2048) == 0     # (round 6) and the list form of the filtered hop
        try:
            comms = local_comms(world)

            def fn(rank):
                shard = build_shard(ScipySource(adj), comms[rank], need_transpose=True, device="cuda:0", relabel=False)
                eng = shard_engine(shard, shard_rows(shard, X), p0, comms[rank], **kw)
                losses = []
                for idx in batches:
                    eng.step(torch.from_numpy(idx).cuda(), float(g["beta"]))
                    losses.append(eng.loss.item())
                eng.forward()
                eng.check_guards()
                return dict(losses=losses, emb=eng.gather_embeddings().cpu().numpy(), params=[t.cpu().numpy().copy() for t in eng.params])

            return _threaded(world, fn, comms)
        finally:
            lib.gss_debug_set_option(b"sparse_bits_rows", 100000)
            lib.gss_debug_set_option(b"spmm_list_blocks", 2048)

    plain, bits = run(100000), run(1)
    for a, b in zip(plain, bits):
        assert a["losses"] == b["losses"]
        np.testing.assert_array_equal(a["emb"], b["emb"])
        for x, y in zip(a["params"], b["params"]):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("world,case,relabel", [(2, "edge_n600_d128_L2", False), (3, "knn_n2000_d64_L3", True), (8, "knn_n200_d16_L2", False)])
def test_overlapped_hops_agree_with_the_plain_sharded_plan(world, case, relabel):
    """gss_shard_desc a_own / a_halo / at_own / at_halo: every hop's boundary exchange on a second stream under the product over the
    shard's own columns, the boundary-column entries added afterwards (forward and backward hops, full and lazy steps, two and three
    layers, an empty shard at world 8).  A row is summed as (own entries) + (boundary entries) instead of in column order, so the
    results agree with the plain sharded plan -- which equals the single-GPU plan bit for bit -- to rounding, not bit for bit: the bound
    is the same multiple of the CPU spread as for the fixtures (tests/tolerances.py).  Replicated state stays identical on all ranks."""
    from gcn_drug_repurposing_amd.dist import local_comms
    from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_engine, shard_rows
    from conftest import golden_batches, golden_params
    g = load_golden(case)
    n, d, L = (int(v) for v in g["meta"])
    adj, X, p0 = golden_csr(g, "A"), g["X"], golden_params(g, "init")
    batches = golden_batches(g)
    kw = dict(num_layers=L, layer_decay=float(g["decay"]), alpha=float(g["alpha"]), lr=float(g["lr"]), max_batch=max(len(b) for b in batches))

    def run(split):
        comms = local_comms(world)

        def fn(rank):
            shard = build_shard(ScipySource(adj), comms[rank], need_transpose=True, device="cuda:0", relabel=relabel, split=split)
            assert shard.layout.overlapped == bool(split)
            eng = shard_engine(shard, shard_rows(shard, X), p0, comms[rank], **kw)
            losses = []
            for k, idx in enumerate(batches):
                (eng.step_lazy if k % 2 else eng.step)(torch.from_numpy(idx.astype(np.int32)).cuda(), float(g["beta"]))
                losses.append(eng.loss.item())
            eng.forward()
            eng.check_guards()
            return dict(losses=losses, emb=eng.gather_embeddings().cpu().numpy(), params=[t.cpu().numpy().copy() for t in eng.params])

        return _threaded(world, fn, comms)

    plain, over = run(False), run(True)
    for r in range(1, world):
        assert over[r]["losses"] == over[0]["losses"]
        np.testing.assert_array_equal(over[r]["emb"], over[0]["emb"])
        for x, y in zip(over[r]["params"], over[0]["params"]):
            np.testing.assert_array_equal(x, y)
    a, b = plain[0], over[0]
    np.testing.assert_allclose(b["losses"], a["losses"], rtol=T.TRAJ_LOSS_RTOL, atol=1e-9)
    assert np.abs(b["emb"] - a["emb"]).max() / np.abs(a["emb"]).max() < T.TRAJ_EMB_REL
    for x, y in zip(a["params"], b["params"]):
        assert np.abs(x - y).max() < T.TRAJ_WEIGHT_LR * float(g["lr"])
    np.testing.assert_allclose(b["losses"], g["losses"], rtol=T.TRAJ_LOSS_RTOL, atol=1e-9)


@pytest.mark.parametrize("world,L", [(3, 2), (2, 3)])
def test_overlapped_hops_with_the_nonzero_row_bitmaps(world, L):
    """the two round-2/3 mechanisms together: plans over huge operands filter the second backward hop by the bitmap of non-zero rows of u
    (forced on here, sparse_bits_rows = 1) AND the hops are overlapped with their exchange -- the own-column pass then is a plain product
    with that bitmap as its gather filter, the boundary-column pass the sparse epilogue.  Small batches, so most rows are skipped."""
    import gcn_drug_repurposing_amd as pkg
    from gcn_drug_repurposing_amd.dist import local_comms
    from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_engine, shard_rows
    from conftest import golden_params
    lib = pkg.load()
    g = load_golden("knn_n2000_d64_L3")
    n, d, _ = (int(v) for v in g["meta"])
    adj, X, p0 = golden_csr(g, "A"), g["X"], golden_params(g, "init")
    rng = np.random.RandomState(21)
    batches = [rng.choice(n, size=s, replace=False).astype(np.int32) for s in (40, 7, 64, 1)]
    kw = dict(num_layers=L, layer_decay=float(g["decay"]), alpha=float(g["alpha"]), lr=float(g["lr"]), max_batch=64)

    def run(split, bits_rows):
        assert lib.gss_debug_set_option(b"sparse_bits_rows", bits_rows) == 0
        try:
            comms = local_comms(world)

            def fn(rank):
                shard = build_shard(ScipySource(adj), comms[rank], need_transpose=True, device="cuda:0", relabel=False, split=split)
                eng = shard_engine(shard, shard_rows(shard, X), p0, comms[rank], **kw)
                losses = []
                for k, idx in enumerate(batches):
                    (eng.step_lazy if k % 2 else eng.step)(torch.from_numpy(idx).cuda(), float(g["beta"]))
                    losses.append(eng.loss.item())
                eng.forward()
                eng.check_guards()
                return dict(losses=losses, emb=eng.gather_embeddings().cpu().numpy(), params=[t.cpu().numpy().copy() for t in eng.params])

            return _threaded(world, fn, comms)
        finally:
            lib.gss_debug_set_option(b"sparse_bits_rows", 100000)

    ref, both = run(False, 100000), run(True, 1)
    for r in range(1, world):
        assert both[r]["losses"] == both[0]["losses"]
        np.testing.assert_array_equal(both[r]["emb"], both[0]["emb"])
    a, b = ref[0], both[0]
    np.testing.assert_allclose(b["losses"], a["losses"], rtol=T.TRAJ_LOSS_RTOL, atol=1e-9)
    assert np.abs(b["emb"] - a["emb"]).max() / np.abs(a["emb"]).max() < T.TRAJ_EMB_REL
    for x, y in zip(a["params"], b["params"]):
        assert np.abs(x - y).max() < T.TRAJ_WEIGHT_LR * float(g["lr"])


def test_zero_pieces_of_a_live_row_are_written_not_skipped():
    """with the non-zero-row bitmap on, the sparse backward hop leaves all-zero rows of u / t unwritten (every reader
    consults the bitmap).  That decision must be per ROW: a row whose bit is set but one of whose 4-feature pieces sums to exactly zero
    has to store that zero, or the piece keeps the previous step's values and the next hop reads them.  Here a 4-column block of both
    weight matrices is zeroed on every other step (the compact input gradients g_ax / g_am then have an exactly-zero piece, and so has
    every row of the hop's sum), after a step that left non-zero values there; losses and parameters must equal the plan without
    bitmaps bit for bit."""
    import gcn_drug_repurposing_amd as pkg
    from gcn_drug_repurposing_amd.dist import local_comms
    from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_engine, shard_rows
    from conftest import golden_params
    lib = pkg.load()
    g = load_golden("knn_n2000_d64_L3")
    n, d, _ = (int(v) for v in g["meta"])
    adj, X, p0 = golden_csr(g, "A"), g["X"], golden_params(g, "init")
    rng = np.random.RandomState(11)
    batches = [rng.choice(n, size=s, replace=False).astype(np.int32) for s in (48, 48, 31, 48, 5, 48)]

    def run(bits_rows):
        assert lib.gss_debug_set_option(b"sparse_bits_rows", bits_rows) == 0
        try:
            comm = local_comms(1)[0]
            shard = build_shard(ScipySource(adj), comm, need_transpose=True, device="cuda:0", relabel=False)
            eng = shard_engine(shard, shard_rows(shard, X), p0, comm, num_layers=2, layer_decay=float(g["decay"]), alpha=float(g["alpha"]),
                               lr=float(g["lr"]), max_batch=64)
            out = []
            for k, idx in enumerate(batches):
                if k % 2 == 1:                      # an exactly-zero 4-feature piece in this step's input gradients
                    eng.params[0][:, 8:12] = 0.0
                    eng.params[2][:, 8:12] = 0.0
                else:
                    eng.params[0][:, 8:12] = torch.from_numpy(p0["W1"][:, 8:12]).cuda()
                    eng.params[2][:, 8:12] = torch.from_numpy(p0["W2"][:, 8:12]).cuda()
                lib.gss_plan_set_step(eng.handle, k)  # weights changed behind the plan's back: its transposed copies are stale
                eng.step(torch.from_numpy(idx).cuda(), float(g["beta"]))
                out.append((eng.loss.item(), [t.cpu().numpy().copy() for t in eng.params], [t.cpu().numpy().copy() for t in eng.grads]))
            eng.check_guards()
            return out
        finally:
            lib.gss_debug_set_option(b"sparse_bits_rows", 100000)

    plain, bits = run(100000), run(1)
    for k, (a, b) in enumerate(zip(plain, bits)):
        assert a[0] == b[0], k
        for x, y in zip(a[1] + a[2], b[1] + b[2]):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("world,relabel", [(2, False), (3, True), (8, False)])
def test_lazy_step_on_shards_equals_the_full_sharded_step(world, relabel):
    """gss_plan_step_lazy on a node-range sharded plan: every shard evaluates the top layer on the batch rows it owns (the members of
    other shards are -1 in its row list and skipped; an empty shard evaluates nothing).  Losses, parameters
    and the embeddings of a full forward afterwards equal the full sharded step's bit for bit, over batches of changing size that
    leave some shards without a member."""
    from gcn_drug_repurposing_amd.dist import local_comms
    from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_engine, shard_rows
    from conftest import golden_params
    g = load_golden("knn_n200_d16_L2" if world == 8 else "knn_n2000_d64_L3")
    n, d, L = (int(v) for v in g["meta"])
    adj, X, p0 = golden_csr(g, "A"), g["X"], golden_params(g, "init")
    rng = np.random.RandomState(9)
    batches = [rng.choice(n, size=s, replace=False).astype(np.int32) for s in (min(n, 150), 3, 1, 64, min(n, 200))]
    kw = dict(num_layers=L, layer_decay=float(g["decay"]), alpha=float(g["alpha"]), lr=float(g["lr"]), max_batch=200)

    def run(lazy):
        comms = local_comms(world)

        def fn(rank):
            shard = build_shard(ScipySource(adj), comms[rank], need_transpose=True, device="cuda:0", relabel=relabel)
            eng = shard_engine(shard, shard_rows(shard, X), p0, comms[rank], **kw)
            losses = []
            lo, hi = shard.part.rows(rank)
            eng.forward()
            for idx in batches:
                before = eng.emb.clone()
                (eng.step_lazy if lazy else eng.step)(torch.from_numpy(idx).cuda(), float(g["beta"]))
                losses.append(eng.loss.item())
                if lazy:
                    # the lazy top layer writes the batch rows this shard owns and nothing else: members of other shards are skipped
                    # (row-list entry -1), not clamped onto one of this shard's rows
                    rows = idx.astype(np.int64) if shard.node_map is None else shard.node_map.cpu().numpy().astype(np.int64)[idx]
                    own = rows[(rows >= lo) & (rows < hi)] - lo
                    untouched = np.ones(hi - lo, dtype=bool)
                    untouched[own] = False
                    sel = torch.from_numpy(np.flatnonzero(untouched)).cuda()
                    assert torch.equal(eng.emb.index_select(0, sel), before.index_select(0, sel)), "a lazy step wrote a row outside its batch"
            eng.forward()
            eng.check_guards()
            return dict(losses=losses, emb=eng.gather_embeddings().cpu().numpy(), params=[t.cpu().numpy().copy() for t in eng.params])

        return _threaded(world, fn, comms)

    full, lazy = run(False), run(True)
    for a, b in zip(full, lazy):
        assert a["losses"] == b["losses"]
        np.testing.assert_array_equal(a["emb"], b["emb"])
        for x, y in zip(a["params"], b["params"]):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("world,case,relabel,split", [(2, "edge_n600_d128_L2", False, False), (3, "knn_n2000_d64_L3", True, True),
                                                       (8, "knn_n200_d16_L2", False, False), (4, "knn_n2000_d64_L3", False, True)])
def test_lazy_halo_fetches_only_what_the_batch_rows_read(world, case, relabel, split):
    """Knob lazy_halo (automatic from 262,144 nodes; forced here): in a lazy step the top layer's `A_hat M` is evaluated on the batch
    rows a shard owns, so of M's boundary rows only the columns of THOSE rows are fetched -- requests as bitmaps over the halo slots,
    both sides listing the set bits in the same order, counts through one device -> host copy.  The unfetched boundary rows keep stale
    values from earlier steps (the weights change every step, so a row wrongly left out would show).  Losses, parameters and
    embeddings equal the plan that exchanges the whole halo bit for bit (same split, so the same summation order); the number of
    rows fetched is exactly the number of distinct boundary columns of the owned batch rows, and what the ranks fetch in total is
    what they send in total.  Round 5: the request phase of M's exchange runs ahead on the plan's request stream -- the host waits for
    its event, never for the caller's stream (gss_plan_sync_stats: no drain); u's sender-driven exchange still drains once per step, and
    with lazy_halo_u = 0 a step's PLAN CODE drains nothing at all, with the same bits.  (gss_plan_sync_stats counts the plan's own
    comm->sync calls.  The in-process backend used here also synchronises inside every collective -- exchange_rows on the request stream
    waits for ev_rq_in, i.e. for the caller's stream -- so the "no drain" property is a property of DEVICE transports (RCCL), which this
    counter can assert but this backend cannot exhibit)  Since round 6 the subset exchange is opt-in over RCCL."""
    import gcn_drug_repurposing_amd as pkg
    from gcn_drug_repurposing_amd.dist import local_comms
    from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_engine, shard_rows
    from conftest import golden_params
    lib = pkg.load()
    g = load_golden(case)
    n, d, L = (int(v) for v in g["meta"])
    adj, X, p0 = golden_csr(g, "A"), g["X"], golden_params(g, "init")
    rng = np.random.RandomState(11)
    batches = [rng.choice(n, size=s, replace=False).astype(np.int32) for s in (min(n, 100), 3, 1, 48, min(n, 128), 17)]
    kw = dict(num_layers=L, layer_decay=float(g["decay"]), alpha=float(g["alpha"]), lr=float(g["lr"]), max_batch=128)

    def run(knob, knob_u=-1):
        assert lib.gss_debug_set_option(b"lazy_halo", knob) == 0 and lib.gss_debug_set_option(b"lazy_halo_u", knob_u) == 0
        try:
            comms = local_comms(world)

            def fn(rank):
                # (local_transpose=False: at two layers the default runs the last backward hop without any exchange of u -- here its
                #  subset exchange is what is under test)
                shard = build_shard(ScipySource(adj), comms[rank], need_transpose=True, device="cuda:0", relabel=relabel, split=split,
                                    local_transpose=False)
                eng = shard_engine(shard, shard_rows(shard, X), p0, comms[rank], **kw)
                lo, hi = shard.part.rows(rank)
                indptr, col = shard.a.h_indptr, shard.a.col.cpu().numpy()[:shard.a.nnz]
                losses, moved, moved_u = [], [], []
                eng.sync_stats()
                for k, idx in enumerate(batches):
                    (eng.step if k == 3 else eng.step_lazy)(torch.from_numpy(idx).cuda(), float(g["beta"]))   # one full step in between
                    losses.append(eng.loss.item())
                    fetched, sent, halo, u_fetched, u_sent, u_halo = eng.lazy_halo_rows()
                    # host-side waits of the step: (drains of the caller's stream, event waits on the request stream)
                    u_on = knob == 1 and knob_u != 0
                    assert eng.sync_stats() == (1 if u_on else 0, 1 if (knob == 1 and k != 3) else 0), (knob, knob_u, k)
                    if knob == 0:
                        assert (fetched, sent, u_fetched, u_sent) == (-1, -1, -1, -1)
                    elif not u_on:
                        assert (u_fetched, u_sent) == (-1, -1)
                    else:
                        assert 0 <= u_fetched <= u_halo == shard.layout.halo_at.n_halo
                        moved_u.append((u_fetched, u_sent))
                    if knob == 1 and k != 3:
                        rows = idx.astype(np.int64) if shard.node_map is None else shard.node_map.cpu().numpy().astype(np.int64)[idx]
                        own = rows[(rows >= lo) & (rows < hi)] - lo
                        cols = np.concatenate([col[indptr[r]:indptr[r + 1]] for r in own]) if len(own) else np.zeros(0, np.int64)
                        assert fetched == len(np.unique(cols[cols >= hi - lo])) and fetched <= halo == shard.layout.halo_a.n_halo
                        moved.append((fetched, sent))
                eng.forward()
                eng.check_guards()
                return dict(losses=losses, moved=moved, moved_u=moved_u, emb=eng.gather_embeddings().cpu().numpy(),
                            params=[t.cpu().numpy().copy() for t in eng.params], halo=shard.layout.halo_a.n_halo, halo_t=shard.layout.halo_at.n_halo)

            return _threaded(world, fn, comms)
        finally:
            lib.gss_debug_set_option(b"lazy_halo", -1)
            lib.gss_debug_set_option(b"lazy_halo_u", -1)

    whole, needed, rccl_like = run(0), run(1), run(1, 0)
    for other in (needed, rccl_like):
        for a, b in zip(whole, other):
            assert a["losses"] == b["losses"]
            np.testing.assert_array_equal(a["emb"], b["emb"])
            for x, y in zip(a["params"], b["params"]):
                np.testing.assert_array_equal(x, y)
    assert [r["moved"] for r in rccl_like] == [r["moved"] for r in needed]
    for k in range(len(needed[0]["moved"])):
        assert sum(r["moved"][k][0] for r in needed) == sum(r["moved"][k][1] for r in needed)
    for k in range(len(batches)):
        assert sum(r["moved_u"][k][0] for r in needed) == sum(r["moved_u"][k][1] for r in needed)
    # a batch of one or three rows reads a small part of the halo, and few rows of u are non-zero
    assert sum(r["moved"][1][0] for r in needed) < sum(r["halo"] for r in needed)
    assert sum(r["moved_u"][2][0] for r in needed) < sum(r["halo_t"] for r in needed)


def test_subset_exchanges_at_a_halo_of_several_compaction_trips():
    """RMAT 600k nodes / 6M edges on two ranks: the request bitmaps are several thousand words long, so the workgroup that lists their set
    bits (bits_compact_kernel) makes several trips of 1024 words with its running total carried across them and across the peers' ranges.
    Lazy and full steps with the subset exchanges forced on equal the whole-halo plan's bit for bit."""
    import gcn_drug_repurposing_amd as pkg
    from gcn_drug_repurposing_amd.dist import local_comms
    from gcn_drug_repurposing_amd.shards import RmatSource, build_shard, gaussian_rows, shard_engine
    lib = pkg.load()
    n, m, d, L, B, world = 600000, 6000000, 32, 2, 1024, 2
    np.random.seed(3)
    p = O.init_layer_weights(d, 1e-2)
    rng = np.random.RandomState(2)
    batches = [rng.permutation(n)[:B].astype(np.int32) for _ in range(3)]

    def run(knob):
        assert lib.gss_debug_set_option(b"lazy_halo", knob) == 0
        try:
            comms = local_comms(world)

            def fn(rank):
                shard = build_shard(RmatSource(n, m, seed=4, device="cuda:0"), comms[rank], need_transpose=True, device="cuda:0", split=False,
                                    local_transpose=False)
                lo, hi = shard.part.rows(rank)
                eng = shard_engine(shard, gaussian_rows(lo, hi, d, 5), p, comms[rank], num_layers=L, layer_decay=0.3, alpha=1.0, lr=1e-3, max_batch=B)
                losses, moved = [], []
                for k, idx in enumerate(batches):
                    (eng.step if k == 1 else eng.step_lazy)(torch.from_numpy(idx).cuda(), 0.25)
                    losses.append(eng.loss.item())
                    moved.append(eng.lazy_halo_rows())
                eng.check_guards()
                sel = torch.from_numpy(np.arange(0, hi - lo, max(1, (hi - lo) // 5000))).cuda()
                return dict(losses=losses, moved=moved, emb=eng.emb.index_select(0, sel).cpu().numpy(), params=[t.cpu().numpy().copy() for t in eng.params],
                            slots=(int(shard.layout.halo_a.send_off[-1]), shard.layout.halo_a.n_halo))

            return _threaded(world, fn, comms)
        finally:
            lib.gss_debug_set_option(b"lazy_halo", -1)

    whole, needed = run(0), run(1)
    assert max(max(r["slots"]) for r in needed) > 32 * 4096, "the halo is too small for several compaction trips"
    for a, b in zip(whole, needed):
        assert a["losses"] == b["losses"]
        np.testing.assert_array_equal(a["emb"], b["emb"])
        for x, y in zip(a["params"], b["params"]):
            np.testing.assert_array_equal(x, y)
    for k in range(3):
        assert sum(r["moved"][k][3] for r in needed) == sum(r["moved"][k][4] for r in needed) > 0
        if k != 1:
            assert sum(r["moved"][k][0] for r in needed) == sum(r["moved"][k][1] for r in needed) > 0


@pytest.mark.parametrize("world,relabel", [(1, False), (2, True), (3, False)])
def test_knn_row_source_builds_the_rows_of_the_reference_graph(world, relabel):
    """Round 4: train.py's kNN adjacency as a ROW SOURCE -- every rank computes the top-k of its own row window
    (gss_knn_topk_rows), the [N][k] table is all-gathered, and a rank assembles only its rows of A + I.  Against the whole-graph builder
    (graph.knn_descriptor_adj_device -> ScipySource): the same work array, and for every range, with and without relabelling, the same
    rowptr / column ids / values entry for entry -- so build_shard gives the same shard whichever source it is handed."""
    import threading
    from gcn_drug_repurposing_amd.dist import local_comms
    from gcn_drug_repurposing_amd.graph import knn_descriptor_adj_device
    from gcn_drug_repurposing_amd.shards import KnnSource, Relabel, ScipySource
    rng = np.random.RandomState(8)
    n, d, k = 777, 20, 5
    X = rng.randn(n, d)
    X[5] = X[9]                                   # a tie between two rows' similarities
    ref = ScipySource(knn_descriptor_adj_device(X, k))
    comms = local_comms(world)
    errors, done = [], [0]

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                src = KnnSource(X, k, device="cuda:0")
                work = src.work(comms[rank], torch.device("cuda:0"))
                assert np.array_equal(work, ref.work(None, None))
                assert src.nnz == ref.nnz
                rl = Relabel(work) if relabel else None
                for lo, hi in ((0, n), (100, 431), (n - 1, n), (0, 0)):
                    for f in ("rows", "rows_t"):
                        got = getattr(src, f)(lo, hi, "cuda:0", relabel=rl)
                        want = getattr(ref, f)(lo, hi, "cuda:0", relabel=rl)
                        for a, b in zip(got, want):
                            assert torch.equal(a.cpu(), b.cpu()), (rank, lo, hi, f)
                torch.cuda.current_stream().synchronize()
                done[0] += 1
        except Exception as e:  # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))
            comms[rank].abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errors, errors
    assert done[0] == world


def test_a_rank_whose_job_wide_knob_differs_is_refused_by_name():
    """lazy_halo / lazy_halo_u / halo_recompute / loss_slab decide WHICH collectives a step enqueues; ranks that
    disagree would wait for each other in different collectives.  gss_plan_create_sharded compares them over the job's communicator and
    fails on every rank, naming the knob -- world 3 as threads: rank 0's plan snapshots halo_recompute = 0 and waits in the all-gather,
    then the process default goes back to -1 and ranks 1, 2 create theirs."""
    import time
    import gcn_drug_repurposing_amd as pkg
    from gcn_drug_repurposing_amd.dist import local_comms
    from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_engine, shard_rows
    lib = pkg.load()
    g = load_golden("edge_n600_d128_L2")
    adj = golden_csr(g, "A")
    n, d = adj.shape[0], 128
    X = np.random.RandomState(0).randn(n, d).astype(np.float32)
    np.random.seed(3)
    p0 = O.init_layer_weights(d, 1e-2)
    world = 3
    comms = local_comms(world)
    shards = _threaded(world, lambda rank: build_shard(ScipySource(adj), comms[rank], need_transpose=True, device="cuda:0"), comms)
    errors, started = [None] * world, threading.Event()

    def create(rank):
        torch.cuda.set_device(0)
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                if rank == 0:
                    started.set()
                shard_engine(shards[rank], shard_rows(shards[rank], X), p0, comms[rank], num_layers=2, layer_decay=0.3, alpha=1.0, lr=1e-3, max_batch=64)
        except Exception as e:  # noqa: BLE001
            errors[rank] = str(e)

    try:
        assert lib.gss_debug_set_option(b"halo_recompute", 0) == 0
        t0 = threading.Thread(target=create, args=(0,))
        t0.start()
        started.wait(60)
        time.sleep(5.0)              # rank 0 has taken its snapshot (the first thing plan creation does) and waits for its peers
    finally:
        assert lib.gss_debug_set_option(b"halo_recompute", -1) == 0
    rest = [threading.Thread(target=create, args=(r,)) for r in (1, 2)]
    [t.start() for t in rest]
    [t.join(300) for t in [t0] + rest]
    assert all(e is not None for e in errors), errors
    assert all("halo_recompute" in e and "same value on every rank" in e for e in errors), errors
    # the same job with agreeing knobs still builds and steps
    comms2 = local_comms(world)
    shards2 = _threaded(world, lambda rank: build_shard(ScipySource(adj), comms2[rank], need_transpose=True, device="cuda:0"), comms2)

    def ok(rank):
        eng = shard_engine(shards2[rank], shard_rows(shards2[rank], X), p0, comms2[rank], num_layers=2, layer_decay=0.3, alpha=1.0, lr=1e-3, max_batch=64)
        eng.step(torch.from_numpy(np.arange(64, dtype=np.int32)).cuda(), 0.2)
        return eng.loss.item()

    losses = _threaded(world, ok, comms2)
    assert len(set(losses)) == 1 and np.isfinite(losses[0])
