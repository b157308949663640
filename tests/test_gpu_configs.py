"""-m gpu: BASELINE.json configs 3, 4 and 5 on their own workloads at full (config 3/4) or 1/10 (config 5) size.

config 3: whole_graph + the 324 NodeCovid<->pathway edge pairs (config_gcn_pathway.json), d = 256, L = 3, B = 2048, 1 GPU
config 4: the same workload node-range sharded (native sharded plan; the ranks are threads on the one GPU of the box,
          the collectives the in-process backend -- the 8-GPU run itself is the driver's)
config 5: RMAT (0.57, 0.19, 0.19, 0.05), d = 128, L = 2, B = 2048: 500k nodes / 10M edges against the full CPU port, and the FULL
          10M nodes / 200M edges on sampled rows

Reference semantics: modules/model.py:152-221 through oracle/torch_cpu_path.py (the reference's torch op sequence on
CPU, fp32) and oracle/gss_oracle.py (numpy, fp64).  Gradient tolerance: gss_loss's gradient is DISCONTINUOUS at S_ij = 0
(relu, model.py:219: G_ij jumps from 0 to alpha beta / B^2), and among the B^2 = 4.2M similarities of a batch a few lie
within fp32 rounding of zero, so any two fp32 forwards (torch's own included) can put them on different sides and then
differ by ~1e-4 of the gradient (tools/accuracy_probe.py: 2 such entries at config 2; the loss kernel itself agrees with
fp64 to 1e-12 on identical inputs).  The gradients are therefore checked to BASELINE.md's 1e-5 against the fp64 backward
fed with dLoss/dE evaluated (in fp64) on the embeddings the device produced -- the backward is linear in dLoss/dE --
and to 1e-3 against the all-fp64 result.
"""
import threading

import numpy as np
import pytest

import tolerances as T
from conftest import record_measured
from oracle import gss_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

DECAY, ALPHA, LR, BETA = 0.3, 1.0, 3e-4, 0.25


def _init(d, seed):
    np.random.seed(seed)
    return O.init_layer_weights(d, 1e-2)


def _cpu_reference(adj, X, p, L, idx):
    """-> (emb fp32 of the torch-CPU port, its loss, its grads, fp64 oracle grads, fp64 emb)"""
    from oracle.torch_cpu_path import TorchCpuPath
    a_hat, _ = O.preprocess_graph(adj)
    a32 = O.to_fp32_csr(a_hat)
    cpu = TorchCpuPath(a32, X, p, L, DECAY, ALPHA, LR)
    e_cpu = cpu.forward()
    l_cpu = cpu.loss(e_cpu, BETA, idx.astype(np.int64))
    cpu.opt.zero_grad()
    l_cpu.backward()
    emb64, cache = O.forward(X, a32, p, L, DECAY, dtype=np.float64)
    g64 = O.backward(cache, O.loss_grad_emb(emb64, BETA, idx, ALPHA))
    return (e_cpu.detach().numpy(), float(l_cpu.detach()), {k: cpu.p[k].grad.numpy().copy() for k in ("W1", "b1", "W2", "b2")},
            (g64, cache, idx), emb64, a_hat)


def _check_grads(got, emb_dev, ref64, tag):
    """got: device gradients; emb_dev: the embeddings the device produced (the input of its loss)"""
    g64, cache, idx = ref64
    g_ref = O.backward(cache, O.loss_grad_emb(np.asarray(emb_dev, np.float64), BETA, idx, ALPHA))
    for k in ("W1", "b1", "W2", "b2"):
        scale = np.abs(g64[k]).max()
        record_measured("configs._check_grads", lin=np.abs(got[k] - g_ref[k]).max() / scale, allfp64=np.abs(got[k] - g64[k]).max() / scale)
        assert np.abs(got[k] - g_ref[k]).max() < 1e-5 * scale + 1e-12, (tag, k, np.abs(got[k] - g_ref[k]).max(), scale)
        # all-fp64: measured <= 2.5e-7 of scale on these workloads (gpurun_out r4a, GSS_RECORD_PARITY); 1e-5 = 40 x that, the same bound
        # as the linearised check -- a similarity within fp32 rounding of the relu's kink would show up here first
        assert np.abs(got[k] - g64[k]).max() < 1e-5 * scale + 1e-12, (tag, k, np.abs(got[k] - g64[k]).max(), scale)


@pytest.fixture(scope="module")
def config3():
    from gcn_drug_repurposing_amd import synth
    adj, _, _ = synth.whole_graph_standin(seed=1, pathway_edges=True)
    n, d, L, B = adj.shape[0], 256, 3, 2048
    assert n == 29960 and adj.nnz + n == 988676
    X = synth.gaussian_features(n, d, seed=3)
    p = _init(d, 7)
    idx = np.random.RandomState(0).permutation(n)[:B]
    ref = _cpu_reference(adj, X, p, L, idx)
    return dict(adj=adj, X=X, p=p, L=L, B=B, idx=idx, ref=ref, n=n, d=d)


def _single_gpu(c, a_hat=None):
    from gcn_drug_repurposing_amd.engine import GssEngine
    from gcn_drug_repurposing_amd.graph import GssGraph
    graph = GssGraph(c["adj"]) if a_hat is None else GssGraph.from_normalized(a_hat)   # same A_hat values as the shards get
    params = [torch.from_numpy(c["p"][k].copy()).cuda() for k in ("W1", "b1", "W2", "b2")]
    eng = GssEngine(graph, torch.from_numpy(c["X"]).cuda(), params, num_layers=c["L"], layer_decay=DECAY, alpha=ALPHA, lr=LR, max_batch=c["B"])
    eng.forward()
    eng.loss_backward(torch.from_numpy(c["idx"].astype(np.int32)).cuda(), BETA)
    torch.cuda.synchronize()
    return eng, graph


def test_config3_full_size_step_matches_cpu_port(config3):
    c = config3
    e_cpu, l_cpu, g_cpu, g64, emb64, _ = c["ref"]
    runs = []
    for _ in range(2):
        eng, _ = _single_gpu(c)
        runs.append((eng.emb.clone(), eng.loss.clone(), [g.clone() for g in eng.grads]))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])     # bitwise reproducible
    for a, b in zip(runs[0][2], runs[1][2]):
        assert torch.equal(a, b)
    emb, loss, grads = runs[0]
    assert torch.equal(grads[1], grads[3])                                          # b1 and b2 share their gradient
    np.testing.assert_allclose(torch.linalg.norm(emb, dim=1).cpu().numpy(), 1.0, atol=2e-6)
    assert np.abs(emb.cpu().numpy() - e_cpu).max() < 1e-5
    assert np.abs(emb.cpu().numpy() - emb64).max() < 1e-5
    assert abs(loss.item() - l_cpu) < 1e-5 * abs(l_cpu)
    _check_grads({k: g.cpu().numpy() for k, g in zip(("W1", "b1", "W2", "b2"), grads)}, emb.cpu().numpy(), g64, "config3")
    # the fused step (the path bench.py times) gives the same loss and moves the weights by ~lr
    eng, _ = _single_gpu(c)
    w0 = eng.params[0].clone()
    eng.step(torch.from_numpy(c["idx"].astype(np.int32)).cuda(), BETA)
    eng.check_guards()
    assert eng.loss.item() == loss.item()
    assert 0 < (eng.params[0] - w0).abs().max().item() < 2.5 * LR


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


@pytest.mark.parametrize("world", [4, 8])
def test_config4_sharded_plan_equals_single_gpu_plan(config3, world):
    """config 4's workload through gss_plan_create_sharded.  Row results do not depend on the shard, so embeddings and
    loss must be BIT-identical to the single-GPU plan's; weight gradients differ only in fp32 summation order."""
    from gcn_drug_repurposing_amd.dist import local_comms, sharded_plan_engine
    c = config3
    _, _, g_cpu, g64, _, a_hat = c["ref"]
    single, _ = _single_gpu(c)                # GssGraph and the shard builder normalise on the device: the same A_hat bits
    emb1, loss1 = single.emb.cpu().numpy(), single.loss.item()
    comms = local_comms(world)
    idx32 = c["idx"].astype(np.int32)

    def rank_fn(rank):
        eng = sharded_plan_engine(c["adj"], c["X"], c["p"], comms[rank], num_layers=c["L"], layer_decay=DECAY, alpha=ALPHA, lr=LR,
                                  max_batch=c["B"], device=torch.device("cuda:0"))
        t = torch.from_numpy(idx32).cuda()
        eng.forward()
        eng.loss_backward(t, BETA)
        res = dict(emb=eng.gather_embeddings().cpu().numpy(), loss=eng.loss.item(), grads=[g.cpu().numpy() for g in eng.grads],
                   rows=eng.n, bounds=eng.part.bounds.copy())
        eng.adam()
        eng.check_guards()
        eng.step(t, BETA)                      # the fused step with its grouped gradient all-reduce
        eng.check_guards()
        res["loss2"] = eng.loss.item()
        res["w1"] = eng.params[0].cpu().numpy()
        return res

    out = _threaded(world, rank_fn, comms)
    assert sum(o["rows"] for o in out) == c["n"]
    for o in out:
        np.testing.assert_array_equal(o["emb"], emb1)
        assert o["loss"] == loss1
        assert o["loss2"] == out[0]["loss2"] and np.array_equal(o["w1"], out[0]["w1"])      # replicas stay bit-identical
        for a, b in zip(o["grads"], out[0]["grads"]):
            np.testing.assert_array_equal(a, b)
    _check_grads(dict(zip(("W1", "b1", "W2", "b2"), out[0]["grads"])), emb1, g64, f"config4 world {world}")
    single.adam()
    single.step(torch.from_numpy(idx32).cuda(), BETA)
    assert abs(out[0]["loss2"] - single.loss.item()) < 1e-6 * abs(single.loss.item())


def test_config4_overlapped_hops_at_full_size(config3):
    """config 4's workload (whole_graph + pathway edges, d = 256, L = 3) at world 4 with every hop overlapped with its boundary exchange
    (split CSRs, second stream; all five kinds of hop occur at three layers).  A row is then summed as own-column entries + boundary-column
    entries: embeddings, loss and gradients agree with the single-GPU plan to rounding instead of bit for bit; replicas stay identical."""
    import tolerances as T
    from gcn_drug_repurposing_amd.dist import local_comms, sharded_plan_engine
    c = config3
    _, _, g_cpu, g64, _, a_hat = c["ref"]
    single, _ = _single_gpu(c)
    emb1, loss1 = single.emb.cpu().numpy(), single.loss.item()
    world = 4
    comms = local_comms(world)
    idx32 = c["idx"].astype(np.int32)

    def rank_fn(rank):
        eng = sharded_plan_engine(c["adj"], c["X"], c["p"], comms[rank], num_layers=c["L"], layer_decay=DECAY, alpha=ALPHA, lr=LR,
                                  max_batch=c["B"], device=torch.device("cuda:0"), split=True)
        assert eng.layout.overlapped
        t = torch.from_numpy(idx32).cuda()
        eng.forward()
        eng.loss_backward(t, BETA)
        res = dict(emb=eng.gather_embeddings().cpu().numpy(), loss=eng.loss.item(), grads=[g.cpu().numpy() for g in eng.grads])
        eng.adam()
        eng.step_lazy(t, BETA)
        eng.check_guards()
        res["loss2"] = eng.loss.item()
        res["w1"] = eng.params[0].cpu().numpy()
        return res

    out = _threaded(world, rank_fn, comms)
    for o in out[1:]:
        np.testing.assert_array_equal(o["emb"], out[0]["emb"])
        assert o["loss"] == out[0]["loss"] and o["loss2"] == out[0]["loss2"] and np.array_equal(o["w1"], out[0]["w1"])
    assert np.abs(out[0]["emb"] - emb1).max() < T.TRAJ_EMB_REL * np.abs(emb1).max()
    assert abs(out[0]["loss"] - loss1) < T.TRAJ_LOSS_RTOL * abs(loss1)
    _check_grads(dict(zip(("W1", "b1", "W2", "b2"), out[0]["grads"])), emb1, g64, "config4 world 4, overlapped hops")
    single.adam()
    single.step(torch.from_numpy(idx32).cuda(), BETA)
    assert abs(out[0]["loss2"] - single.loss.item()) < T.TRAJ_LOSS_RTOL * abs(single.loss.item())


@pytest.fixture(scope="module")
def rmat_mid():
    from gcn_drug_repurposing_amd import synth
    n, m, d, L, B = 500_000, 10_000_000, 128, 2, 2048
    adj = synth.rmat_adj(n, m, seed=4)
    X = synth.gaussian_features(n, d, seed=5)
    return dict(adj=adj, X=X, p=_init(d, 7), L=L, B=B, idx=np.random.RandomState(1).permutation(n)[:B], n=n, d=d)


def test_config5_rmat_500k_10m_properties_and_step_vs_cpu_port(rmat_mid):
    """RMAT at 1/20 of config 5 (the gathered operand, 256 MB, already exceeds the caches): SpMM adjointness <A x, y> = <x, A^T y>, row
    sums of A_hat within 1e-5 of the host preprocess_graph, and one FULL step against the CPU port and the fp64 oracle.  (The full-size
    graph is checked on sampled rows in the next test: a host oracle of 10M nodes takes minutes.)"""
    from gcn_drug_repurposing_amd import _lib
    c = rmat_mid
    eng, graph = _single_gpu(c)
    lib, st = _lib.load(), _lib.current_stream()
    n, d = c["n"], c["d"]
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(n, d, device="cuda", generator=g)
    y = torch.randn(n, d, device="cuda", generator=g)
    ax, aty = torch.empty_like(x), torch.empty_like(x)
    _lib.check(lib.gss_spmm(graph.a.handle, d, x.data_ptr(), ax.data_ptr(), None, None, st))
    _lib.check(lib.gss_spmm(graph.at.handle, d, y.data_ptr(), aty.data_ptr(), None, None, st))
    lhs, rhs = (ax.double() * y.double()).sum().item(), (x.double() * aty.double()).sum().item()
    assert abs(lhs - rhs) < 1e-5 * max(abs(lhs), abs(rhs), 1.0)
    ones, rs = torch.ones(n, 16, device="cuda"), torch.empty(n, 16, device="cuda")
    _lib.check(lib.gss_spmm(graph.a.handle, 16, ones.data_ptr(), rs.data_ptr(), None, None, st))
    e_cpu, l_cpu, g_cpu, g64, emb64, a_hat = _cpu_reference(c["adj"], c["X"], c["p"], c["L"], c["idx"])
    ref_rs = np.asarray(a_hat.sum(1)).reshape(-1)
    assert np.abs(rs[:, 0].cpu().numpy() - ref_rs).max() < 1e-5 * max(1.0, np.abs(ref_rs).max())
    emb = eng.emb.cpu().numpy()
    np.testing.assert_allclose(np.sqrt((emb.astype(np.float64) ** 2).sum(1)), 1.0, atol=2e-6)
    assert np.abs(emb - e_cpu).max() < 1e-5 and np.abs(emb - emb64).max() < 1e-5
    assert abs(eng.loss.item() - l_cpu) < 1e-5 * abs(l_cpu)
    _check_grads({k: g.cpu().numpy() for k, g in zip(("W1", "b1", "W2", "b2"), eng.grads)}, emb, g64, "rmat 500k")


def test_config5_rmat_10m_200m_full_size_sampled_rows():
    """BASELINE config 5 at FULL size on one GPU (the device row source builds it in ~20 s): exactly 210M stored entries, SpMM
    adjointness over the whole graph, and a sample of rows of A_hat's row sums, of AX = A_hat X, of AM = A_hat (AX (.) X) (a dozen of them over
    both hops from X alone) and of the first layer's P recomputed on the host in fp64 from the device's own CSR rows (a full host oracle at
    this size would take minutes); the four weight gradients against an fp64 re-summation of the plan's own dP / AX / AM over all 10M rows; and the two
    backward hops of a whole step (sampled rows of u, t and dP_0 against fp64 from the plan's own batch gradients / u / t / P_0)."""
    from gcn_drug_repurposing_amd import _lib
    from gcn_drug_repurposing_amd.dist import local_comms
    from gcn_drug_repurposing_amd.shards import RmatSource, build_shard, gaussian_rows, shard_engine
    n, m, d, L, B = 10_000_000, 200_000_000, 128, 2, 2048
    comm = local_comms(1)[0]
    shard = build_shard(RmatSource(n, m, seed=4, device="cuda:0"), comm, need_transpose=True, device="cuda:0")
    assert shard.nnz_global == m + n and shard.a.nnz == m + n and shard.at.nnz == m + n and shard.relabel is not None
    lib, st = _lib.load(), _lib.current_stream()
    gen = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(n, 16, device="cuda", generator=gen)
    y = torch.randn(n, 16, device="cuda", generator=gen)
    ax, aty = torch.empty_like(x), torch.empty_like(x)
    _lib.check(lib.gss_spmm(shard.a.handle, 16, x.data_ptr(), ax.data_ptr(), None, None, st))
    _lib.check(lib.gss_spmm(shard.at.handle, 16, y.data_ptr(), aty.data_ptr(), None, None, st))
    lhs, rhs = (ax.double() * y.double()).sum().item(), (x.double() * aty.double()).sum().item()
    assert abs(lhs - rhs) < 1e-5 * max(abs(lhs), abs(rhs), 1.0)                     # <A x, y> = <x, A^T y>
    del x, y, ax, aty
    p = _init(d, 7)
    X = gaussian_rows(0, n, d, 5)                                                    # rows of the relabelled graph
    eng = shard_engine(shard, X, p, comm, num_layers=L, layer_decay=DECAY, alpha=ALPHA, lr=LR, max_batch=B)
    idx = np.random.RandomState(1).permutation(n)[:B].astype(np.int32)
    eng.forward()
    eng.loss_backward(torch.from_numpy(idx).cuda(), BETA)
    torch.cuda.synchronize()
    assert np.isfinite(eng.loss.item()) and all(torch.isfinite(g).all().item() for g in eng.grads)
    emb_norm = torch.linalg.norm(eng.emb, dim=1)
    assert (emb_norm - 1).abs().max().item() < 2e-6
    # sampled rows (hubs, a stride through the rest, the tail) against fp64 on the host
    rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(1000, n, n // 300), np.arange(n - 40, n)]))
    rp = shard.a.h_indptr
    col = shard.a.col
    val = shard.a.val
    AX = eng.activation(0, "AX")
    AM = eng.activation(0, "AM")
    P = eng.activation(0, "P")
    w1, w2 = p["W1"].astype(np.float64), p["W2"].astype(np.float64)
    worst = 0.0
    for r in rows:
        e0, e1 = int(rp[r]), int(rp[r + 1])
        c = col[e0:e1].cpu().numpy().astype(np.int64)
        v = val[e0:e1].cpu().numpy().astype(np.float64)
        ref_ax = v @ X[c].astype(np.float64)
        got_ax = AX[r].cpu().numpy().astype(np.float64)
        assert np.abs(got_ax - ref_ax).max() < 1e-5 * max(1.0, np.abs(ref_ax).max()), r
        # P = AX W1^T + AM W2^T (+ zero biases) from the device's AX / AM rows: checks the projection kernel at this size
        ref_p = got_ax @ w1.T + AM[r].cpu().numpy().astype(np.float64) @ w2.T
        worst = max(worst, np.abs(P[r].cpu().numpy() - ref_p).max() / max(1.0, np.abs(ref_p).max()))
    assert worst < 1e-5
    # (round 6) the SECOND hop at full size: AM = A_hat (AX (.) X), modules/model.py:168-169.
    #  (i) every sampled row from the device's AX rows of its neighbours (M = AX (.) X in fp32 as the kernel's epilogue forms it, the sum in fp64);
    #  (ii) a dozen rows from X ALONE: both hops recomputed in fp64 on the host from the device's CSR rows (the neighbours' rows included)
    Xd = eng.x
    worst_am = 0.0
    for r in rows:
        e0, e1 = int(rp[r]), int(rp[r + 1])
        c = col[e0:e1].long()
        v = val[e0:e1].cpu().numpy().astype(np.float64)
        m_rows = (AX[c] * Xd[c]).cpu().numpy().astype(np.float64)
        ref_am = v @ m_rows
        worst_am = max(worst_am, np.abs(AM[r].cpu().numpy() - ref_am).max() / max(1.0, np.abs(ref_am).max()))
    assert worst_am < 1e-5, worst_am
    two_hop = [int(r) for r in rows if rp[r + 1] - rp[r] <= 64][::max(1, len(rows) // 12)][:12]
    assert len(two_hop) >= 8
    worst_2 = 0.0
    for r in two_hop:
        e0, e1 = int(rp[r]), int(rp[r + 1])
        c = col[e0:e1].cpu().numpy().astype(np.int64)
        v = val[e0:e1].cpu().numpy().astype(np.float64)
        ref_am = np.zeros(d)
        for cj, vj in zip(c, v):
            f0, f1 = int(rp[cj]), int(rp[cj + 1])
            cc = col[f0:f1].cpu().numpy().astype(np.int64)
            vv = val[f0:f1].cpu().numpy().astype(np.float64)
            ax_c = vv @ X[cc].astype(np.float64)
            ref_am += vj * (ax_c * X[cj].astype(np.float64))
        worst_2 = max(worst_2, np.abs(AM[r].cpu().numpy() - ref_am).max() / max(1.0, np.abs(ref_am).max()))
    assert worst_2 < 1e-5, worst_2
    # the weight gradients at full size (train.py:183): dW1 = dP_top^T AX_1[batch rows] + dP_0^T AX_0 (dW2 with AM, db = column sums of both dP),
    # recomputed in fp64 from the plan's own dP / AX / AM buffers in chunks of 1M rows (torch on the device as an independent summation):
    # checks the 10M-row reductions of wgrad_tn_kernel + wgrad_reduce_kernel, which the 500k case does not reach
    del AX, AM, P
    dP0 = eng.activation(0, "dP")
    dPb = eng.activation(0, "dP_batch")[:B].double()
    brow = eng.node_map.long()[torch.from_numpy(idx).cuda().long()]
    ref_w1 = dPb.t() @ eng.activation(L - 1, "AX")[brow].double()
    ref_w2 = dPb.t() @ eng.activation(L - 1, "AM")[brow].double()
    ref_b = dPb.sum(0)
    for which, acc in (("AX", ref_w1), ("AM", ref_w2)):
        Z = eng.activation(0, which)
        for s0 in range(0, n, 1 << 20):
            acc += dP0[s0:s0 + (1 << 20)].double().t() @ Z[s0:s0 + (1 << 20)].double()
        del Z
    for s0 in range(0, n, 1 << 20):
        ref_b += dP0[s0:s0 + (1 << 20)].double().sum(0)
    for got, ref, name in ((eng.grads[0], ref_w1, "dW1"), (eng.grads[2], ref_w2, "dW2"), (eng.grads[1], ref_b, "db1"), (eng.grads[3], ref_b, "db2")):
        err = (got.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
        assert err < 1e-5, (name, err)
    # the two backward hops at full size (train.py:183), through the WHOLE-STEP path -- the batch-sparse first hop over the listed workgroups of
    # the batch rows' neighbourhood, the second hop under the bitmap of the rows the first one wrote:
    #   u = (A_hat^T g_am) (.) x_in (+ g_ax on batch rows), t = (A_hat^T g_am) (.) AX_1 on sampled WRITTEN rows, from the plan's own batch gradients;
    #   dP_0 = (t + A_hat^T u) (.) ELU'(P_0) on sampled rows, from the plan's own u / t / P_0 (rows the first hop did not write are zero by contract)
    del dP0, dPb, ref_w1, ref_w2
    eng.step(torch.from_numpy(idx).cuda(), BETA)
    torch.cuda.synchronize()
    assert np.isfinite(eng.loss.item())
    written = eng.written_rows_bitmap()
    assert written is not None and 0 < int(written.sum().item()) < n // 10        # a batch of 2,048 touches a small part of 10M rows
    u, t_, P0, dP0, AX1 = eng.activation(0, "u"), eng.activation(0, "t"), eng.activation(0, "P"), eng.activation(0, "dP"), eng.activation(L - 1, "AX")
    gb = eng.activation(0, "g_batch")
    g_ax_b, g_am_b = gb[:B].double(), gb[B:2 * B].double()
    pos = torch.full((n,), -1, dtype=torch.int64, device="cuda")
    pos[brow] = torch.arange(B, device="cuda")
    rp_t, col_t, val_t = shard.at.h_indptr, shard.at.col, shard.at.val
    live = torch.nonzero(written).reshape(-1).cpu().numpy()
    pick_live = live[::max(1, len(live) // 150)][:150]
    scale_u, scale_t = u[torch.from_numpy(live).cuda()].abs().max().item(), t_[torch.from_numpy(live).cuda()].abs().max().item()
    assert scale_u > 0 and scale_t > 0
    worst_u = worst_t = 0.0
    for r in pick_live:
        e0, e1 = int(rp_t[r]), int(rp_t[r + 1])
        c = col_t[e0:e1].long()
        v = val_t[e0:e1].double()
        hit = pos[c] >= 0
        dm = (v[hit].view(-1, 1) * g_am_b[pos[c][hit]]).sum(0) if bool(hit.any()) else torch.zeros(d, dtype=torch.float64, device="cuda")
        p0r = P0[r].double()
        x_in = torch.where(p0r > 0, p0r, torch.exp(p0r) - 1.0)                        # the top layer's input row: ELU(P_0) (model.py:173,201-203)
        ref_u = dm * x_in + (g_ax_b[pos[r]] if int(pos[r]) >= 0 else 0.0)
        ref_t = dm * AX1[r].double()
        worst_u = max(worst_u, (u[r].double() - ref_u).abs().max().item() / scale_u)
        worst_t = max(worst_t, (t_[r].double() - ref_t).abs().max().item() / scale_t)
    assert worst_u < 1e-5 and worst_t < 1e-5, (worst_u, worst_t)
    scale_dp = dP0.abs().max().item()
    assert scale_dp > 0
    not_batch = np.setdiff1d(rows, brow.cpu().numpy())
    worst_dp = 0.0
    for r in not_batch[::2]:
        e0, e1 = int(rp_t[r]), int(rp_t[r + 1])
        c = col_t[e0:e1].long()
        v = val_t[e0:e1].double()
        urows = torch.where(written[c].view(-1, 1), u[c].double(), torch.zeros(1, dtype=torch.float64, device="cuda"))
        s_ = (v.view(-1, 1) * urows).sum(0)
        t_r = t_[r].double() if bool(written[r]) else torch.zeros(d, dtype=torch.float64, device="cuda")
        p0r = P0[r].double()
        ref_dp = (t_r + s_) * torch.where(p0r > 0, torch.ones_like(p0r), torch.exp(p0r))
        worst_dp = max(worst_dp, (dP0[r].double() - ref_dp).abs().max().item() / scale_dp)
    assert worst_dp < 1e-5, worst_dp
    # row sums of A_hat of the sampled rows against D^-1/2 (A + I) D^-1/2 recomputed from the unit-weight structure
    deg = np.diff(rp).astype(np.float64)                                             # row sums of A + I (unit weights)
    for r in rows[::8]:
        e0, e1 = int(rp[r]), int(rp[r + 1])
        c = col[e0:e1].cpu().numpy().astype(np.int64)
        ref = (deg[r] ** -0.5) * (deg[c] ** -0.5)
        assert np.abs(val[e0:e1].cpu().numpy() - ref.astype(np.float32)).max() <= 1.2e-7 * ref.max() + 1e-12, r


def test_config2_downstream_auc_with_the_real_drug_indication_pairs():
    """north star: 'embeddings whose downstream evaluate_auc.py scores match the CPU reference within 1e-4' -- at FULL size, on the
    real labels.  Two epochs (30 steps) of config 2 on the whole_graph stand-in from the same start on the device and through the
    torch-CPU port of the reference's op sequence; then evaluate_auc.py:156-170's statistic (consumer.indication_aucs: one ROC-AUC
    per indication over the 1,661 drugs, positives = the 5,926 pairs of the reference's data/drug_indication_df.tsv) on both
    embeddings.  What evaluate_auc.py reports -- the median and the mean over the 840 indications -- must agree within 1e-4.  A
    single indication's AUC moves by 1 / (n_pos n_neg) ~ 1.2e-4 when one positive and one negative drug with near-equal scores
    swap places, which fp32 rounding does to a handful of the 840: those are bounded separately."""
    from gcn_drug_repurposing_amd import consumer, synth
    from gcn_drug_repurposing_amd.engine import GssEngine
    from gcn_drug_repurposing_amd.graph import GssGraph
    from oracle.torch_cpu_path import TorchCpuPath
    adj, ntype, names = synth.whole_graph_standin(seed=1)
    n, d, L, B = adj.shape[0], 128, 2, 2048
    X = synth.gaussian_features(n, d, seed=2)
    np.random.seed(7)
    p = O.init_layer_weights(d, 1e-5)                                   # train.py's default --init-weights
    rng = np.random.RandomState(3)
    batches = []
    for _ in range(2):
        perm = rng.permutation(n)
        batches += [perm[i:i + B] for i in range(0, n, B)]
    params = [torch.from_numpy(p[k].copy()).cuda() for k in ("W1", "b1", "W2", "b2")]
    eng = GssEngine(GssGraph(adj), torch.from_numpy(X).cuda(), params, num_layers=L, layer_decay=DECAY, alpha=ALPHA, lr=LR, max_batch=B)
    eng.forward()
    beta = eng.percentile(98.0)                                          # train.sh: --beta-percentile 98
    a_hat, _ = O.preprocess_graph(adj)
    cpu = TorchCpuPath(O.to_fp32_csr(a_hat), X, p, L, DECAY, ALPHA, LR)
    for idx in batches:
        eng.step(torch.from_numpy(idx.astype(np.int32)).cuda(), beta)
        eng.check_guards()
        emb_cpu, loss_cpu = cpu.step(idx.astype(np.int64), beta)
    emb_gpu = eng.emb.cpu().numpy()                                      # the last forward, as train.py:193 writes it
    record_measured("config2_auc.traj", loss_rel=abs(eng.loss.item() - loss_cpu) / abs(loss_cpu), emb_abs=np.abs(emb_gpu - emb_cpu.numpy()).max())
    # measured after the 30 steps (round 4, GSS_RECORD_PARITY): loss identical to the CPU port's, max |emb difference| 5.4e-7 (8.2e-7 in
    # round 3); the bounds are tests/tolerances.py's for a trajectory (10 x the spread between CPU executions), 1e-5 = 12-18 x measured
    assert abs(eng.loss.item() - loss_cpu) < T.TRAJ_LOSS_RTOL * abs(loss_cpu)
    assert np.abs(emb_gpu - emb_cpu.numpy()).max() < 1e-5
    drugs = [names[i] for i in np.nonzero(ntype == 0)[0]]
    inds = [names[i] for i in np.nonzero(ntype == 1)[0] if names[i] != "NodeCovid"]
    positives = synth.standin_drug_indications()
    auc_gpu, used_g = consumer.indication_aucs(emb_gpu, names, drugs, inds, positives)
    auc_cpu, used_c = consumer.indication_aucs(emb_cpu.numpy(), names, drugs, inds, positives)
    assert used_g == used_c and len(auc_gpu) == 840
    assert abs(np.median(auc_gpu) - np.median(auc_cpu)) < 1e-4 and abs(auc_gpu.mean() - auc_cpu.mean()) < 1e-4
    delta = np.abs(auc_gpu - auc_cpu)
    record_measured("config2_auc.auc", max_delta=delta.max(), frac_above_1e4=(delta > 1e-4).mean(), median=abs(np.median(auc_gpu) - np.median(auc_cpu)))
    # The north star's 1e-4 is on the two scores evaluate_auc.py prints (median and mean, asserted above).  Per indication the honest
    # unit is a SWAPPED PAIR: embeddings that differ by 6e-7 order two near-equal scores differently now and then, and one positive
    # and one negative drug swapping places moves that indication's AUC by exactly 1 / (n_pos n_neg) -- 3.0e-4 for an indication with
    # two positive drugs.  History: round 3 no indication differed by more than 3.6e-5; round 4 one at 3.0e-4 (a pair at n_pos = 2).
    # Round 5 traced that to HOW the row is normalised, not to the order of the norm's additions (unchanged since round 3): the
    # kernels multiplied by 1 / max(||x||, eps) -- two roundings -- where F.normalize divides; with the division
    # (csrc/common.h unit4) the count is back to round 3's: 1-2 indications differ at all, by one pair each, none by more than 3.6e-5
    # (tools/norm_order_probe.py on two batch orders, profiles/r05_norm_probe_*.txt).  Asserted: no indication differs by more than
    # 1e-4, none by more than one pair.
    dset = set(drugs)
    n_pos = np.array([sum(1 for dname in positives.get(ind, ()) if dname in dset) for ind in used_g], dtype=np.float64)
    pairs = delta * n_pos * (len(drugs) - n_pos)
    record_measured("config2_auc.pairs", max_pairs=pairs.max())
    assert pairs.max() < 1.5, (pairs.max(), delta.max())
    assert delta.max() < 1e-4, (delta.max(), (delta > 1e-4).sum())
    # predict_drug.py:55-73: the drugs ranked for the COVID node
    r_gpu, _ = consumer.rank_by_query(emb_gpu, names, "NodeCovid", drugs)
    r_cpu, _ = consumer.rank_by_query(emb_cpu.numpy(), names, "NodeCovid", drugs)
    assert len(set(r_gpu[:20]) & set(r_cpu[:20])) >= 18
