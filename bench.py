#!/usr/bin/env python3
"""bench.py -- the reference's headline workload on MI355X.

    python bench.py --gpus 1 --steps 60 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
           bench.py --gpus N --steps K --warmup W

A "step" is one iteration of train.py:155-184 (full-graph forward of the L-layer GSS GCN, gss_loss on one
batch of B node ids, backward, Adam) on the whole_graph stand-in (BASELINE.json configs[1]: N = 29,960,
nnz(A_hat) = 988,028, d = 128, L = 2, B = 2048, synthetic -- the real edgelist is not shipped).

Prints ONE JSON line.  `value` = ALGORITHMIC SpMM edge traversals per second over the whole job: the reference's step
runs 2L forward + 2(L-1) backward SpMMs over nnz(A_hat) stored entries and that count is what is divided by the wall
time.  Layer-1 results are NOT cached; one of the backward SpMMs (the top layer's, whose operand is non-zero on the B
batch rows only) visits just the entries whose neighbour is a batch row -- exact, but fewer gathers than nnz, so the
figure is reference-equivalent work per second, not executed gathers (`spmm_edges_executed_per_step` has those).
`epoch_time_s` is the other half of BASELINE.json's metric.  `roofline` prices the profile's dominant kernel (the
forward SpMM with the fused Hadamard epilogue) from HIP events recorded around every launch during a second pass over
the same steps, `roofline_plain` the plain forward SpMM; `cpu_baseline` times the reference's torch-CPU op sequence
(oracle/torch_cpu_path.py, kind "port") on the host cores for a few steps of the same workload.

`python bench.py --gpus N` without a torch.distributed environment starts its own N ranks (one per GPU, RCCL) before
anything touches the GPU and relays rank 0's line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
PMC_FILE = "r06_spmm_pmc.json"             # tools/pmc_pack_r06.py: counters per SpMM mode at configs 2, 3 and 5 + the spmm.hip hash they were taken with
MFMA_F32_PEAK_TFLOPS = 157.3
GUIDE_HBM_STREAM_TBS = 6.3   # MI355X_MICROARCH.md: "8 TB/s peak (spec); ~6.3 TB/s achievable"
GUIDE_MALL_STREAM_TBS = 8.6  # MI355X_MICROARCH.md: 38 MB table, uniformly random 512-B rows served by the Infinity Cache
# the bare gather stream of a workload's own col[] array (col[] streamed, one 512-B / 1-KB row gathered per entry as 256-B slices pinned to XCDs, nothing else): the rate at which
# the cache hierarchy serves this graph's gathers, measured with tools/micro/gather_hub_lds.py (H = 0 rows) on the GPU box
GATHER_CEILING = {("whole_graph", 128): {"TBs": 19.42, "source": "profiles/r05_gather_ceiling.txt: 506 MB of gathers in 26.1 us (r03: 19.37)"},
                  ("whole_graph_pathway", 256): {"TBs": 19.02, "source": "profiles/r05_gather_ceiling.txt: 1,012 MB of gathers in 53.2 us"}}
XGMI_LINK_GBS = 153.0        # one xGMI link, one direction (7 links per GPU, one per peer in an 8-GPU node)
COLLECTIVE_TIMEOUT_S = 300.0 # a stream that does not drain for this long = a peer stopped taking part: abort and exit non-zero


def spmm_bytes(nnz, n, d, extra_rows=0):
    """8 nnz + 4 (N+1) + 4 N d (X once) + 4 N d (Y) [+ 4 N d per extra operand]"""
    return 8 * nnz + 4 * (n + 1) + 8 * n * d + 4 * n * d * extra_rows


def free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def self_launch(args):
    """--gpus N outside a torch.distributed job: start N ranks of this script with torch.distributed.run as a CHILD
    process (nothing in this process has touched the GPU yet; a process that has must never exec) and relay the
    child's JSON line and exit code."""
    import subprocess
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    raise SystemExit(res.returncode if res.returncode != 0 or lines else 1)


def build_workload(name, d_override=None):
    from gcn_drug_repurposing_amd import synth
    if name == "whole_graph":
        adj, _, _ = synth.whole_graph_standin(seed=1)
        d, L, B = 128, 2, 2048
        x = synth.gaussian_features(adj.shape[0], d_override or d, seed=2)
    elif name == "whole_graph_knn":
        # train.py's own adjacency: the "descriptor" kNN graph (k = 5) of the input features (helpers/helper.py:25-58),
        # built on the device (gss_knn_topk); same N, d, L, B as whole_graph
        from gcn_drug_repurposing_amd.graph import knn_descriptor_adj_device
        d, L, B = 128, 2, 2048
        x = synth.gaussian_features(29960, d_override or d, seed=2)
        adj = knn_descriptor_adj_device(x.astype(np.float64), 5)
    elif name == "whole_graph_pathway":
        adj, _, _ = synth.whole_graph_standin(seed=1, pathway_edges=True)
        d, L, B = 256, 3, 2048
        x = synth.gaussian_features(adj.shape[0], d_override or d, seed=3)
    elif name.startswith("rmat"):
        # rmat:<nodes>:<edges>, default a 1/10-scale version of BASELINE config 5.  No matrix on the host: the edge stream is
        # generated on the device and every rank keeps the rows it owns (shards.RmatSource); features per row block
        from gcn_drug_repurposing_amd.shards import RmatSource
        parts = name.split(":")
        n = int(parts[1]) if len(parts) > 1 else 1_000_000
        m = int(parts[2]) if len(parts) > 2 else 20_000_000
        d, L, B = 128, 2, 2048
        return RmatSource(n, m, seed=4), None, (d_override or d), L, B
    else:
        raise SystemExit(f"unknown workload {name}")
    return adj, x, (d_override or d), L, B


_RESULT_FD = 1
_PG_DEVICE = "cuda"          # where the tensors of torch.distributed collectives live ("cpu" under GSS_COMM_BACKEND=host: gloo)


def emit(result):
    os.write(_RESULT_FD, (json.dumps(result) + "\n").encode())


def bench_diffusion(args, rank, world, local_rank):
    """--workload diffusion: all diffusion profiles of the whole-graph stand-in.  A step = the
    complete batched power iteration for every drug and indication; start nodes are independent, so N ranks take
    N contiguous slices of them with no collective (strong scaling: the total is fixed)."""
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist
    import gcn_drug_repurposing_amd as pkg
    from gcn_drug_repurposing_amd import synth
    from gcn_drug_repurposing_amd.diffusion import PprEngine, PprProblem
    pkg.load()
    adj, ntype, _ = synth.whole_graph_standin(seed=1)
    m0 = sp.csr_matrix(adj, dtype=np.float64)
    di = np.flatnonzero(ntype <= 1)
    prot = {int(s): m0.indices[m0.indptr[s]:m0.indptr[s + 1]].tolist() for s in di}
    alpha, max_iter, tol = 0.8595436247434408, 1000, 1e-6          # evaluate_auc.py:80-83
    mine = np.array_split(di, world)[rank]
    prob = PprProblem(m0, mine, prot)
    eng = PprEngine(prob)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        x, it = eng.run(alpha, tol, max_iter)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x, it = eng.run(alpha, tol, max_iter)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=_PG_DEVICE)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    sums = x[:, :prob.k].sum(0)
    if not bool(((sums - 1).abs() < 1e-9).all()):
        raise SystemExit("a diffusion profile does not sum to 1")
    n, nnz = prob.n, int(prob.mt.nnz)
    out = {"metric": "diffusion_profiles_per_s", "value": len(di) * args.steps / elapsed, "unit": "profiles/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           **({"knobs": list(args.set)} if args.set else {}),
           "config": {"workload": f"diffusion profiles of the whole_graph stand-in: N={n}, nnz(M')={nnz}, {len(di)} start nodes "
                                  f"(all drugs and indications), alpha={alpha}, tol={tol}; {int(it.max())} power iterations",
                      "parallelism": "single" if world == 1 else f"start nodes split over {world} ranks, no collective"}}
    if rank == 0:
        y = torch.empty_like(x)
        for _ in range(3):
            eng.spmm(x, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # launched on torch's current stream
        e0.record()
        for _ in range(10):
            eng.spmm(x, y)
        e1.record()
        torch.cuda.synchronize()
        avg_s = e0.elapsed_time(e1) / 10 * 1e-3
        alg = 12 * nnz + 4 * (n + 1) + 2 * 8 * n * prob.kpad
        out["roofline"] = {"bound": "hbm", "kernel": "ppr_spmm_kernel (y = M'^T x, fp64)", "achieved": alg / avg_s / 1e9, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": alg / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": None, "alg_bytes_per_launch": alg,
                           "avg_launch_us": avg_s * 1e6, "launches": 10}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import diffusion_oracle as O
            pick = di[np.linspace(0, len(di) - 1, 64).astype(int)]
            t1 = time.perf_counter()
            worst = 0.0
            xs = x[:, :prob.k].t().contiguous().cpu().numpy()
            for s in pick:
                ref, _ = O.diffusion_profile(m0, int(s), prot, alpha, max_iter, tol)
                worst = max(worst, float(np.abs(ref - xs[np.searchsorted(di, s)]).max()))
            t_cpu = (time.perf_counter() - t1) / len(pick)
            out["cpu_baseline"] = {"value": 1.0 / t_cpu, "unit": "profiles/s", "cores": 1, "kind": "port",
                                   "sample": f"{len(pick)} of the {len(di)} start nodes, one scipy power iteration each "
                                             f"(oracle/diffusion_oracle.py), max |device - oracle| = {worst:.1e}"}
        emit(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="whole_graph")
    ap.add_argument("--hidden-units", type=int, default=None)
    ap.add_argument("--cache-layer1", action="store_true", help="also report the step time with layer-1 SpMMs cached")
    ap.add_argument("--no-lazy-top", dest="lazy_top", action="store_false", help="do not also time gss_plan_step_lazy (the top layer on the "
                    "batch rows only: the same loss, gradients and parameters bit for bit; reported as `lazy_top`, never part of `value`)")
    ap.add_argument("--pipeline", action="store_true", help="cross-step layer-1 software pipelining on a second HIP stream "
                                                             "(measured: no gain, off by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=20, help="timed steps of the CPU baseline (~0.5 s each at config 2)")
    ap.add_argument("--min-time", type=float, default=0.5, help="after the K timed steps of the contract, keep stepping until this many "
                    "seconds have been timed and report that steadier figure as `long_run` (0 = off)")
    ap.add_argument("--spinup-time", type=float, default=0.3, help="seconds of untimed steps BEFORE the W warm-up steps, to bring the GPU to "
                    "its working clocks (a 5-step warm-up is 1.6 ms at this size); reported as spinup_steps")
    ap.add_argument("--set", action="append", default=[], metavar="KNOB=VALUE", help="gss_debug_set_option before anything runs (A/B runs of "
                    "a kernel variant, e.g. --set gemm_variant=3); listed in the JSON line as `knobs`")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)          # never returns

    import torch
    import torch.distributed as dist

    # stdout carries exactly one JSON line.  RCCL prints a version banner to the C stdout when a communicator is
    # created, so file descriptor 1 is pointed at stderr for the whole run and the result goes to the saved descriptor.
    sys.stdout.flush()
    global _RESULT_FD
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # GSS_COMM_BACKEND=host: the ranks are still separate processes with a plan each, but their collectives are staged through host
    # memory over gloo and they may share a GPU -- how a ONE-GPU box rehearses `--gpus N` (RCCL refuses two ranks on a device).
    # Never a measurement: the JSON line says so in config.parallelism and carries "rehearsal": true.
    host_backend = os.environ.get("GSS_COMM_BACKEND", "rccl").lower() == "host"
    global _PG_DEVICE
    _PG_DEVICE = "cpu" if host_backend else "cuda"
    if host_backend:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        if host_backend:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=2 * COLLECTIVE_TIMEOUT_S))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=2 * COLLECTIVE_TIMEOUT_S))

    import gcn_drug_repurposing_amd as pkg
    from gcn_drug_repurposing_amd.graph import GssGraph
    lib_ = pkg.load()
    for kv in args.set:
        name, _, val = kv.partition("=")
        if lib_.gss_debug_set_option(name.encode(), int(val)) != 0:
            raise SystemExit(f"--set {kv}: {lib_.gss_last_error().decode()}")

    if args.workload == "diffusion":
        bench_diffusion(args, rank, world, local_rank)
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    adj, x_host, d, L, B = build_workload(args.workload, args.hidden_units)
    from_source = x_host is None           # a row source instead of a matrix in memory (RMAT)
    n = adj.n if from_source else adj.shape[0]
    B = min(B, n)
    steps_per_epoch = (n + B - 1) // B
    rng = np.random.RandomState(1234)
    batches = []
    while len(batches) < args.warmup + args.steps:
        perm = rng.permutation(n).astype(np.int32)
        batches += [perm[i:i + B] for i in range(0, n, B)]
    batches = batches[:args.warmup + args.steps]
    beta, alpha, lr, decay = 0.25, 1.0, 3e-4, 0.3
    np.random.seed(7)
    w = np.random.randn(d, d) * 1e-5
    np.fill_diagonal(w, 1.0)
    params_host = {"W1": w.astype(np.float32), "b1": np.zeros(d, np.float32), "W2": w.astype(np.float32).copy(),
                   "b2": np.zeros(d, np.float32)}

    sharded = world > 1 or os.environ.get("GSS_FORCE_SHARDED") == "1"   # the env knob lets a 1-GPU box exercise the RCCL path
    if sharded and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if host_backend:
            dist.init_process_group("gloo", rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    halo_info = None
    shard = None
    # one GPU, a matrix in memory: the same builder as the shards (it relabels the nodes hub-first: -1.7 % of a step at config 2);
    # the A/B options of the plain plan (--cache-layer1, --pipeline) keep the GssGraph path
    single_via_shard = not sharded and not from_source and not args.cache_layer1 and not args.pipeline
    if from_source or single_via_shard or sharded:
        # the native sharded path (also world = 1 for a row source): a gss_plan per rank that owns an RCCL communicator
        # (gss_plan_create_sharded); torch.distributed only hands the communicator's 128-byte id around and takes the MAX of
        # the timings.  Every rank builds its own rows only (shards.build_shard).
        from gcn_drug_repurposing_amd.dist import job_comm as make_job_comm, local_comms
        from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, gaussian_rows, shard_engine
        comm = make_job_comm(world, rank) if sharded else local_comms(1)[0]
        t_setup = time.perf_counter()
        from gcn_drug_repurposing_amd.shards import row_weight_for
        shard = build_shard(adj if from_source else ScipySource(adj), comm, need_transpose=L > 1, row_weight=row_weight_for(d, L))
        lo_, hi_ = shard.part.rows(rank)
        from gcn_drug_repurposing_amd.shards import shard_rows
        x_rows = gaussian_rows(lo_, hi_, d, 5) if from_source else shard_rows(shard, x_host)
        engine = shard_engine(shard, x_rows, params_host, comm, num_layers=L, layer_decay=decay, alpha=alpha, lr=lr, max_batch=B)
        torch.cuda.synchronize()
        nnz = engine.global_nnz
        fa, ft = shard.layout.halo_fraction()
        halo_info = {"rows": int(hi_ - lo_), "halo_rows_a": int(shard.layout.halo_a.n_halo),
                     "halo_rows_at": int(shard.layout.halo_at.n_halo) if shard.layout.halo_at is not None else 0,
                     "halo_fraction_a": fa, "halo_fraction_at": ft, "setup_s": time.perf_counter() - t_setup,
                     "plan_bytes": engine.device_bytes()}
        parallelism = (f"node-range shards x{world}, native plan, boundary rows by grouped ncclSend/ncclRecv per SpMM hop" if sharded
                       else "single")
        if sharded and host_backend:
            parallelism = (f"REHEARSAL (GSS_COMM_BACKEND=host): node-range shards x{world}, one process and one native plan per rank, "
                           f"collectives staged through host memory over gloo, {torch.cuda.device_count()} GPU(s) shared by the ranks -- not a measurement")
        if shard.relabel is not None:
            parallelism += "; nodes relabelled hub-first"
        if shard.layout.overlapped:
            parallelism += "; hops overlapped with their boundary exchange (own-column product on the main stream, exchange on a second one)"
        halo_info["overlapped_hops"] = bool(shard.layout.overlapped)
    else:
        from gcn_drug_repurposing_amd.engine import GssEngine
        graph = GssGraph(adj, need_transpose=L > 1)
        nnz = graph.nnz
        feats = torch.from_numpy(x_host).cuda()
        params = [torch.from_numpy(params_host[k].copy()).cuda() for k in ("W1", "b1", "W2", "b2")]
        engine = GssEngine(graph, feats, params, num_layers=L, layer_decay=decay, alpha=alpha, lr=lr, max_batch=B,
                           pipeline_layer1=args.pipeline)
        parallelism = "single" if not args.pipeline else "single GPU; next step's layer-1 SpMMs on a 2nd HIP stream"

    idx_all = torch.from_numpy(np.concatenate(batches)).cuda()
    offs = np.concatenate([[0], np.cumsum([len(b) for b in batches])]).astype(np.int64)
    job_comm = comm if sharded else None     # the RCCL communicator the plans of this job hold

    def run(lo, hi):
        for s in range(lo, hi):
            engine.step(idx_all, beta, count=int(offs[s + 1] - offs[s]), offset=int(offs[s]))

    def barrier():
        # sharded: wait for the stream with a deadline and the RCCL error poll (gss_comm_sync) -- a peer that stopped taking part
        # aborts the communicator and this rank exits non-zero instead of hanging until the driver's timeout
        if job_comm is not None:
            job_comm.sync(COLLECTIVE_TIMEOUT_S)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if world == 1:
            return float(v)
        t = torch.tensor([v], dtype=torch.float64, device=_PG_DEVICE)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks(vals):
        """[world][len(vals)] of every rank's floats"""
        t = torch.tensor(list(vals), dtype=torch.float64, device=_PG_DEVICE)
        if world == 1:
            return t.cpu().numpy()[None]
        out = torch.empty(world * t.numel(), dtype=torch.float64, device=_PG_DEVICE)
        dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy().reshape(world, -1)

    # spin-up: untimed steps on the warm-up batches until the clocks have settled (the same count on every rank: the step is a
    # collective), then the contract's W warm-up steps and K timed steps
    spinup_steps = 0
    if args.spinup_time > 0 and args.warmup > 0:
        run(0, args.warmup)
        barrier()
        t_ = time.perf_counter()
        run(0, args.warmup)
        barrier()
        per = max((time.perf_counter() - t_) / args.warmup, 1e-6)
        reps = int(min(args.spinup_time / per / args.warmup, 2000))
        if world > 1:
            t = torch.tensor([reps], dtype=torch.int64, device=_PG_DEVICE)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            reps = int(t.item())
        for _ in range(reps):
            run(0, args.warmup)
        spinup_steps = (2 + reps) * args.warmup
    run(0, args.warmup)
    barrier()
    t0 = time.perf_counter()
    run(args.warmup, args.warmup + args.steps)
    barrier()
    elapsed_own = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed_own)
    loss_end = float(engine.loss.item())
    if not np.isfinite(loss_end):
        raise SystemExit(f"non-finite loss {loss_end} after the timed steps")
    # the contract's K steps are a few milliseconds at this size; a longer region over the same batches gives a figure
    # that can be tracked from round to round
    long_run = None
    if args.min_time > 0:
        reps = int(np.ceil(args.min_time / max(elapsed, 1e-6)))
        reps = max(1, min(reps, 100000))
        barrier()
        t1 = time.perf_counter()
        for _ in range(reps):
            run(args.warmup, args.warmup + args.steps)
        barrier()
        el2 = max_over_ranks(time.perf_counter() - t1)
        long_run = {"steps": reps * args.steps, "seconds": el2, "ms_per_step": el2 / (reps * args.steps) * 1e3}

    lazy_top = None
    if args.lazy_top and world == 1 and hasattr(engine, "step_lazy"):
        try:
            # opt-in extra: the same steps through gss_plan_step_lazy (NOT the reported value: the reference's step computes all N rows of
            # the top layer; this one computes the B rows anything reads).  Same batches, timed the same way.
            def run_lazy(lo, hi):
                for s in range(lo, hi):
                    engine.step_lazy(idx_all, beta, count=int(offs[s + 1] - offs[s]), offset=int(offs[s]))
            run_lazy(0, args.warmup)
            reps = max(1, int(np.ceil(args.min_time / max(elapsed, 1e-6)))) if args.min_time > 0 else 1
            barrier()
            t1 = time.perf_counter()
            for _ in range(reps):
                run_lazy(args.warmup, args.warmup + args.steps)
            barrier()
            lz = (time.perf_counter() - t1) / (reps * args.steps)
            trainer_ms = None
            if not from_source and not sharded and shard is not None:
                # what train.py runs: lazy steps on a plan that also keeps layer 1's two SpMM results (their inputs, A_hat and X, are
                # constants; tests: bitwise neutral).  A second plan over the same shard.
                eng2 = shard_engine(shard, x_rows, params_host, comm, num_layers=L, layer_decay=decay, alpha=alpha, lr=lr, max_batch=B,
                                    cache_layer1=True)
                def run2(lo, hi):
                    for s in range(lo, hi):
                        eng2.step_lazy(idx_all, beta, count=int(offs[s + 1] - offs[s]), offset=int(offs[s]))
                run2(0, args.warmup)
                barrier()
                t2 = time.perf_counter()
                for _ in range(reps):
                    run2(args.warmup, args.warmup + args.steps)
                barrier()
                trainer_ms = (time.perf_counter() - t2) / (reps * args.steps) * 1e3
                del eng2
            lazy_top = {"ms_per_step": lz * 1e3, "ms_per_step_with_layer1_kept": trainer_ms, "steps": reps * args.steps,
                        "final_loss": float(engine.loss.item()),
                        "note": "gss_plan_step_lazy: top layer's A_hat M / projection / ELU / normalise on the batch rows only; loss, gradients "
                                "and parameters bit-identical to the full step (tests/test_gpu_train.py); not part of `value`.  ms_per_step_with_layer1_kept: the same "
                                "on a plan that keeps layer 1's SpMM results (constant inputs) -- the step train.py runs"}
        except Exception as e:  # noqa: BLE001  (an extra must never cost the run its headline line)
            lazy_top = {"error": repr(e)}

    spmm_per_step = 2 * L + 2 * (L - 1)
    ms_per_step = elapsed / args.steps * 1e3
    value = spmm_per_step * nnz * args.steps / elapsed
    # gathers actually executed: the top layer's backward SpMM only follows entries whose neighbour is one of the B batch rows
    # (expected nnz * B / N of them for a random batch); every other SpMM follows all nnz
    edges_executed = ((spmm_per_step - 1) * nnz + nnz * B / n) if L > 1 else spmm_per_step * nnz

    out = {
        "metric": "gcn_spmm_edges_per_s", "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "spinup_steps": spinup_steps, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "epoch_time_s": ms_per_step * 1e-3 * steps_per_epoch,
        # `value` counts the reference's work (every SpMM at nnz); value_executed counts the entries the kernels actually visited
        "value_executed": edges_executed * args.steps / elapsed,
        "config": {"workload": f"{args.workload} stand-in (value = {value:.4g} edges/s counts every SpMM at nnz as the reference executes it; "
                               f"value_executed = {edges_executed * args.steps / elapsed:.4g} counts the entries the kernels visited): "
                               f"N={n}, nnz(A_hat)={nnz}, d={d}, L={L}, B={B}, "
                               f"{steps_per_epoch} steps/epoch; full train step (fwd+gss_loss+bwd+Adam), "
                               f"{spmm_per_step} SpMMs/step counted at nnz each (algorithmic; the top layer's first backward SpMM "
                               f"visits only entries whose neighbour is a batch row"
                               + (", its second one only entries whose neighbour row is non-zero (graphs of >= 100k nodes)" if n >= 100000 and L > 1 else "")
                               + "; value_executed has the executed count)",
                   "parallelism": parallelism, "final_loss": loss_end},
    }
    if args.set:
        out["knobs"] = list(args.set)
    if lazy_top:
        out["lazy_top"] = lazy_top
    if L > 1:
        out["spmm_edges_executed_per_step"] = edges_executed
        if n >= 100000:
            # from 100k nodes on the hop after it skips neighbours whose row of its operand is all zeros, too (how many depends on the
            # batch's neighbourhood and is not counted): the figure above is an upper bound there
            out["spmm_edges_executed_is_upper_bound"] = True
    if long_run:
        out["long_run"] = long_run
        out["long_run"]["edges_per_s"] = spmm_per_step * nnz / (long_run["ms_per_step"] * 1e-3)

    # ---- roofline leg: HIP events around every kernel class over the same steps, on every rank ----
    # An event pair around a launch also brackets the gap the marker packets open between kernels, so the bracketed times sum to
    # more than the un-instrumented step (round 2: 0.340 vs 0.307 ms).  The per-launch overhead is measured -- (sum of the bracketed
    # times - the wall time of the same steps without events) / launches -- and subtracted, so the class times use the SAME timer as
    # the headline and sum to ms_per_step; both the raw and the corrected launch time are reported.
    if hasattr(engine, "comm_stats"):
        engine.comm_stats()                  # reset: the counts below are those of the profiled steps
        engine.sync_stats()
    engine.profile(True)
    run(args.warmup, args.warmup + args.steps)
    prof = engine.profile_read()
    engine.profile(False)
    barrier()
    launches = sum(v[1] for v in prof.values())
    raw_ms_per_step = sum(v[0] for v in prof.values()) / args.steps
    own_ms_per_step = elapsed_own / args.steps * 1e3
    ev_over_us = max(0.0, (raw_ms_per_step - own_ms_per_step) * 1e3 / max(launches / args.steps, 1))

    def class_us(cls):
        """average launch time of a class on the headline's timer, or None when the class is too short for the correction to mean
        anything: the overhead is an AVERAGE over all launches of a step (a bracket around a 5 us launch does not open the same gap as
        one around a 15 ms launch), so a class whose raw bracket is below 3 x the overhead is reported as null instead of a difference
        of two comparable numbers (the RMAT line carried negative class times and an MFMA fraction of 0.94)"""
        ms, cnt = prof[cls]
        if not cnt:
            return None
        raw = ms / cnt * 1e3
        return (raw - ev_over_us) if raw >= 3.0 * ev_over_us else None

    out["kernel_us"] = {k: class_us(k) for k, v in prof.items() if v[1]}
    out["kernel_us_raw_event_bracket"] = {k: v[0] / v[1] * 1e3 for k, v in prof.items() if v[1]}
    out["kernel_ms_per_step"] = {k: (class_us(k) * v[1] / args.steps * 1e-3 if class_us(k) is not None else None) for k, v in prof.items() if v[1]}
    out["event_overhead_us_per_launch"] = ev_over_us
    out["kernel_us_null_below_us"] = 3.0 * ev_over_us
    out["launches_per_step"] = launches / args.steps

    # the shard this rank computes on (the whole graph at world 1): rows, stored entries, operand rows incl. the boundary rows
    n_loc = engine.n
    nnz_loc = int(engine.graph.a.nnz)
    rows_op = n_loc + (int(shard.layout.halo_a.n_halo) if shard is not None else 0)

    def roof(cls, kernel, extra_rows, what):
        us = class_us(cls)
        if us is None or us <= 0:
            return None
        avg = us * 1e-6
        # 8 nnz + 4 (N+1) + 4 (operand rows) d + 4 N d (+ 4 N d per extra operand / result), on THIS rank's shard
        alg = 8 * nnz_loc + 4 * (n_loc + 1) + 4 * rows_op * d + 4 * n_loc * d * (1 + extra_rows)
        ach = alg / avg / 1e9
        return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": None, "traffic_source": None, "alg_bytes_per_launch": alg, "alg_bytes_model": what,
                "avg_launch_us": us, "avg_launch_us_raw_event_bracket": prof[cls][0] / prof[cls][1] * 1e3, "launches": prof[cls][1],
                "shard": {"rank": rank, "rows": n_loc, "nnz": nnz_loc, "operand_rows": rows_op}}

    # the dominant kernel of the step (profiles/*_kernel_stats.csv): AX = A_hat X with the Hadamard epilogue M = AX (.) X
    out["roofline"] = roof("spmm_fwd_hadamard", "spmm_balanced_kernel<FWD1> (AX = A_hat . X, M = AX (.) X, forward)", 1,
                           "8 nnz + 4 (N+1) + 4 N d (X, also the Hadamard operand) + 4 N d (AX) + 4 N d (M)")
    out["roofline_plain"] = roof("spmm_fwd", "spmm_balanced_kernel<PLAIN> (AM = A_hat . M, forward)", 0,
                                 "8 nnz + 4 (N+1) + 4 N d (M) + 4 N d (AM)")
    single = world == 1 and not sharded
    # ---- the backward SpMMs (SURVEY 8(d)(i) "fwd and bwd reported separately") ----
    # spmm_bwd2 = the top layer's second backward hop  dP = c (t + A_hat^T u) (.) ELU'(P) + residual on the batch rows  (SPMM_BWD2S): a full
    #   product over A_hat^T whose epilogue streams three more N x d operands: 8 nnz + 4 (N+1) + 4 N d (u, gathered) + 4 N d (t) + 4 N d (P) +
    #   4 N d (dP) + 4 B d (the compact residual).  (From 100k nodes on it skips neighbours whose row of u is zero: fewer bytes than this.)
    # spmm_bwd1 = the top layer's first backward hop, batch-sparse (SPMM_BWD1S): the kernel streams the whole index (it must see every column
    #   id) but gathers only for entries whose neighbour is a batch row.  Algorithmic bytes = what a product restricted to those entries needs:
    #   8 hits + 4 (N+1) + 2 * 4 B d (the compact g_am / g_ax) + 4 streams of d floats over the R rows that have such an entry or are batch rows
    #   (x_in and ax read, u and t written).  hits and R are counted on the device for the last timed batch.
    # spmm_bwd1_dense / spmm_bwd2_dense (L >= 3): the N-row hops of the layers below: 8 nnz + 4 (N+1) + 6 N d / 5-6 N d streams.
    bwd_hits = bwd_rows = None
    if L > 1 and world == 1 and getattr(engine.graph, "at", None) is not None:
        at_ = engine.graph.at
        ids = idx_all[int(offs[args.warmup + args.steps - 1]):int(offs[args.warmup + args.steps])].long()
        if getattr(engine, "node_map", None) is not None:
            ids = engine.node_map.long()[ids]
        mark = torch.zeros(n_loc, dtype=torch.bool, device="cuda")
        mark[ids] = True
        hit = mark[at_.col[:at_.nnz].long()]
        bwd_hits = int(hit.sum().item())
        cnt_ = (at_.rowptr[1:] - at_.rowptr[:-1]).long()
        row_hit = torch.zeros(n_loc, dtype=torch.bool, device="cuda")
        row_hit[torch.repeat_interleave(torch.arange(n_loc, device="cuda"), cnt_)[hit]] = True
        bwd_rows = int((row_hit | mark).sum().item())
        del mark, hit, row_hit, cnt_

    def roof_bytes(cls, kernel, alg, what, extra=None):
        us = class_us(cls) if prof.get(cls, (0, 0))[1] else None
        if us is None or us <= 0:
            return None
        ach = alg / (us * 1e-6) / 1e9
        r = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
             "traffic_source": None, "alg_bytes_per_launch": alg, "alg_bytes_model": what, "avg_launch_us": us,
             "avg_launch_us_raw_event_bracket": prof[cls][0] / prof[cls][1] * 1e3, "launches": prof[cls][1]}
        if extra:
            r.update(extra)
        return r

    if L > 1 and world == 1:
        nnz_t = int(engine.graph.at.nnz)
        base_t = 8 * nnz_t + 4 * (n_loc + 1)
        out["roofline_bwd2"] = roof_bytes("spmm_bwd2", "spmm_balanced_kernel<BWD2S> (dP = c (t + A_hat^T u) (.) ELU'(P) + batch-row residual; the top layer's second backward hop)",
                                          base_t + 4 * n_loc * d * 4 + 4 * B * d, "8 nnz + 4 (N+1) + 4 N d (u) + 4 N d (t) + 4 N d (P) + 4 N d (dP) + 4 B d (residual)"
                                          + ("; the kernel skips neighbours whose row of u is zero (graphs of >= 100k nodes): it moves fewer bytes than this" if n >= 100000 else ""))
        if bwd_hits is not None:
            alg1 = 8 * bwd_hits + 4 * (n_loc + 1) + 8 * B * d + 16 * bwd_rows * d
            out["roofline_bwd1"] = roof_bytes("spmm_bwd1", "spmm_balanced_kernel<BWD1S> (u, t from A_hat^T restricted to batch-row neighbours; the top layer's first backward hop)",
                                              alg1, "8 hits + 4 (N+1) + 2 * 4 B d (compact g_am, g_ax) + 4 * 4 R d (x_in, ax read; u, t written on the R live rows)",
                                              {"hits": bwd_hits, "live_rows": bwd_rows,
                                               "executed_bytes_model": 8 * nnz_t + 4 * (n_loc + 1) + 8 * B * d + 8 * n_loc * d + 8 * bwd_rows * d,
                                               "executed_bytes_note": "what the kernel streams: the whole index (8 nnz), u and t written for every row (zeros "
                                                                      "included, unless the plan tracks non-zero rows: graphs of >= 100k nodes), x_in / ax for the live rows"})
        if L > 2:
            out["roofline_bwd1_dense"] = roof_bytes("spmm_bwd1_dense", "spmm_balanced_kernel<BWD1> (u = g_ax + (A_hat^T g_am) (.) x, t = (A_hat^T g_am) (.) ax; N-row hop)",
                                                    base_t + 4 * n_loc * d * 6, "8 nnz + 4 (N+1) + 4 N d x (g_am gathered, g_ax, x_in, ax read; u, t written)")
            out["roofline_bwd2_dense"] = roof_bytes("spmm_bwd2_dense", "spmm_balanced_kernel<BWD2> (dP = c (t + A_hat^T u) (.) ELU'(P) [+ residual]; N-row hop)",
                                                    base_t + 4 * n_loc * d * 4, "8 nnz + 4 (N+1) + 4 N d x (u gathered, t, P read; dP written) [+ 4 N d per residual read / gx written at L > 3]")
    if world == 1:
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per SpMM mode on this workload, corrected as MI355X_MICROARCH.md prescribes.  Counters cannot be read
        # from inside bench.py: the figures come from an OFFLINE profile (tools/profile_r06.sh -> tools/pmc_pack_r06.py), which records the sha256
        # of the spmm.hip it measured.  A file taken with another spmm.hip than this library's is NOT used (traffic stays null and says why).
        pmc_path = os.path.join(ROOT, "profiles", PMC_FILE)
        wl_key = {"whole_graph": "config2", "whole_graph_pathway": "config3", "rmat:10000000:200000000": "config5"}.get(args.workload)
        want_d = {"config2": 128, "config3": 256, "config5": 128}.get(wl_key)
        lib_hash = lib_.gss_source_hash(b"spmm.hip")
        lib_hash = lib_hash.decode() if lib_hash else None
        if wl_key and d == want_d and os.path.exists(pmc_path):
            z = json.load(open(pmc_path))
            rec_hash = z.get("source_hash", {}).get("spmm.hip")
            for key, name in (("roofline", "fwd1"), ("roofline_plain", "plain"), ("roofline_bwd1", "bwd1s"), ("roofline_bwd2", "bwd2s"),
                              ("roofline_bwd1_dense", "bwd1"), ("roofline_bwd2_dense", "bwd2")):
                r = out.get(key)
                c = z.get(wl_key, {}).get(name)
                if not r or not c:
                    continue
                if rec_hash != lib_hash:
                    r["traffic_source"] = (f"profiles/{PMC_FILE} was taken with spmm.hip {str(rec_hash)[:12]}, this library is {str(lib_hash)[:12]}: "
                                           "stale counters are not reported (re-run tools/profile_r06.sh)")
                    continue
                r["traffic"] = c["traffic_bytes_per_launch"]
                r["traffic_over_alg"] = c["traffic_bytes_per_launch"] / r["alg_bytes_per_launch"]
                r["l2_hit_rate"] = c.get("l2_hit_rate")
                r["traffic_source"] = (f"offline rocprofv3 --pmc profile of the same kernel on this workload, spmm.hip {rec_hash[:12]} (= this library's): "
                                       f"profiles/{PMC_FILE} [{wl_key}][{name}]")
        # Which bound binds.  `frac` above is the prescribed one: compulsory bytes against HBM.  A cache-resident
        # graph (config 2 / 3: the operand fits the Infinity Cache) is bound by the rate at which the L2s serve row gathers -- SURVEY
        # 8(d)'s bytes_gather = 8 nnz + 4 (N+1) + 4 nnz d + 4 N d per result, against the bare gather stream of the same col[] array
        # (tools/micro/gather_ceiling.py / gather_hub_lds.py: col[] streamed, 512-B rows gathered, nothing else).  A graph far beyond the
        # caches (RMAT 10M) is bound by what leaves the L2s: the counters' bytes per launch against the guide's streaming rates.
        for key, extra in (("roofline", 1), ("roofline_plain", 0)):
            r = out[key]
            if not r:
                continue
            avg = r["avg_launch_us"] * 1e-6
            bytes_gather = 8 * nnz_loc + 4 * (n_loc + 1) + 4 * nnz_loc * d + 4 * n_loc * d * (1 + extra) + (4 * n_loc * d if extra else 0)
            ceil = GATHER_CEILING.get((args.workload, d))
            r["gather"] = {"bytes_gather": bytes_gather, "achieved": bytes_gather / avg / 1e12, "unit": "TB/s",
                           "ceiling": ceil["TBs"] if ceil else None, "frac": (bytes_gather / avg / 1e12 / ceil["TBs"]) if ceil else None,
                           "ceiling_source": ceil["source"] if ceil else "no bare-stream measurement of this graph on file",
                           "model": "SURVEY 8(d) bytes_gather: 8 nnz + 4 (N+1) + 4 nnz d (one row gather per entry) + 4 N d per result"
                                    + (" + 4 N d (the Hadamard operand's own rows)" if extra else "")}
            if r.get("traffic"):
                r["fabric"] = {"achieved": r["traffic"] / avg / 1e12, "unit": "TB/s", "streaming_hbm": GUIDE_HBM_STREAM_TBS,
                               "infinity_cache_hit": GUIDE_MALL_STREAM_TBS,
                               "frac_of_streaming_hbm": r["traffic"] / avg / 1e12 / GUIDE_HBM_STREAM_TBS,
                               "note": "bytes that left the L2s per launch (roofline.traffic) over the launch time, against the guide's measured "
                                       "streaming rates (MI355X_MICROARCH.md: HBM 6.3 TB/s, Infinity-Cache-resident 8.6 TB/s)"}
        if out["roofline_plain"]:
            out["spmm_kernel_edges_per_s"] = nnz / (out["roofline_plain"]["avg_launch_us"] * 1e-6)
        # nnz / t_SpMM per launch, forward and backward kinds separately.  spmm_bwd1 at L = 2 is the sparsity-aware
        # top-layer hop (it visits only entries whose neighbour is a batch row), counted at nnz like the rest
        out["spmm_kernel_edges_per_s_by_kind"] = {k: nnz / (class_us(k) * 1e-6)
                                                  for k in ("spmm_fwd_hadamard", "spmm_fwd", "spmm_bwd1", "spmm_bwd2", "spmm_bwd1_dense", "spmm_bwd2_dense")
                                                  if prof.get(k, (0, 0))[1] and class_us(k)}
        # fp32-MFMA kernels: achieved TFLOP/s against the 157.3 TF dense fp32-matrix peak
        mf = {}
        dgrad_own = prof.get("dgrad", (0, 0))[1] > 0     # one GPU: finish and input gradient are two launches (knob loss_dgrad)
        for cls, name, fl in (("dense_fwd", "projection (gemm_nt)", 2.0 * n * (2 * d) * d),
                              ("wgrad", "weight gradients (wgrad_tn: L - 1 launches over N rows, the top layer's B rows merged into the first)",
                               2.0 * (n * (L - 1) + B) * d * (2 * d)),
                              ("loss", ("loss: B x B sweep + finish (two launches), 4 B^2 d; the batch rows' input gradient is a launch of its own (class dgrad)"
                                        if dgrad_own else "loss: B x B sweep + finish with the batch rows' input gradient in the same launch, 4 B^2 d + 4 B d^2"),
                               4.0 * B * B * d + (0.0 if dgrad_own else 4.0 * B * d * d))):
            cnt = prof[cls][1]
            if not cnt or class_us(cls) is None:
                continue
            per_step = cnt / args.steps
            # several launches of a class per step (L projections; L - 1 full weight gradients; the loss's two launches): flops of all
            # of them over the time of all of them
            mult = {"dense_fwd": L, "wgrad": 1, "loss": 1}[cls]
            t = class_us(cls) * 1e-6 * per_step
            mf[cls] = {"kernel": name, "achieved": fl * mult / t / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": fl * mult / t / 1e12 / MFMA_F32_PEAK_TFLOPS, "us_per_step": t * 1e6}
        out["mfma"] = mf
        if "dense_fwd" in mf:
            out["mfma_dense_fwd"] = mf["dense_fwd"]
    if sharded:
        # ---- multi-GPU keys: what RCCL saw, per-rank timings, the exchange against the link rate ----
        halo_bytes_a = int(shard.layout.halo_a.n_halo) * d * 4
        halo_bytes_t = (int(shard.layout.halo_at.n_halo) * d * 4) if shard.layout.halo_at is not None else 0
        pair_a = int(np.diff(shard.layout.halo_a.recv_off).max()) * d * 4 if world > 1 else 0
        pair_t = (int(np.diff(shard.layout.halo_at.recv_off).max()) * d * 4) if (world > 1 and shard.layout.halo_at is not None) else 0
        # the three comm classes use their RAW event brackets: a collective's bracket has no neighbouring kernel gap to correct for
        comm_cls = {k: prof[k] for k in ("comm", "comm_batch", "comm_grads") if prof.get(k, (0, 0))[1]}
        comm_ms = sum(v[0] for v in comm_cls.values()) / args.steps
        stats = engine.comm_stats() if hasattr(engine, "comm_stats") else (0, 0, 0)        # since the last read: the profiled steps
        out["collectives_per_step"] = {"boundary_row_exchanges": stats[0] / args.steps, "batch_row_allreduces": stats[1] / args.steps,
                                       "weight_gradient_allreduces": stats[2] / args.steps, "total": sum(stats) / args.steps,
                                       "us_each": {k: v[0] / v[1] * 1e3 for k, v in comm_cls.items()},
                                       "note": "gss_plan_comm_stats over the profiled steps; us_each = event bracket around one collective on this "
                                               "rank's stream (pack kernel + transfer for an exchange), the wait for the slowest peer included"}
        syncs = engine.sync_stats() if hasattr(engine, "sync_stats") else (0, 0)
        out["host_syncs_per_step"] = {"stream_drains": syncs[0] / args.steps, "request_stream_event_waits": syncs[1] / args.steps,
                                      "note": "(counts the plan's own waits; a host-side backend -- in-process, host-staged -- also synchronises inside every collective, so "
                                              "zero here means 'no drain' only over a device transport = RCCL) "
                                              "gss_plan_sync_stats over the profiled (full) steps: host waits that drain the caller's stream (the "
                                              "sender-driven subset exchange of u, knob lazy_halo_u: off over RCCL) / host waits for an event of the "
                                              "plan's request stream while the caller's stream keeps running (the lazy step's subset exchange of M)"}
        mine = [own_ms_per_step, comm_ms, halo_bytes_a, halo_bytes_t, pair_a, pair_t, out["roofline"]["frac"] if out["roofline"] else 0.0,
                out["roofline"]["avg_launch_us"] if out["roofline"] else 0.0, n_loc, nnz_loc]
        allv = all_ranks(mine)
        # (halo_recompute: layer 2's boundary input rows are computed, not fetched; two layers: the last backward hop runs on A_hat's shard
        #  transposed in place and fetches nothing)
        tloc = L >= 2 and getattr(shard.layout, "a_loc_t", None) is not None and (n < 262144 or engine.lazy_halo_rows()[3] < 0)
        hops_a, hops_t = max(0, 2 * L - 2 - (1 if L > 1 else 0)), max(0, 2 * L - 3 - (1 if tloc else 0))
        # a hop is done when its most loaded pair is: every pair has its own xGMI link (8 GPUs fully connected), so the floor per hop
        # is max over pairs of bytes / link rate; the all-reduces (B d, 2 B d and 2 (d^2 + d) floats) are latency-bound and not priced
        ideal_us = (hops_a * allv[:, 4].max() + hops_t * allv[:, 5].max()) / (XGMI_LINK_GBS * 1e9) * 1e6
        out["rccl_ranks"] = job_comm.count()
        if host_backend:
            out["rehearsal"] = True
        out["per_rank"] = {"ms_per_step": allv[:, 0].tolist(), "ms_per_step_min": float(allv[:, 0].min()), "ms_per_step_max": float(allv[:, 0].max()),
                           "comm_ms_per_step": allv[:, 1].tolist(), "rows": allv[:, 8].astype(int).tolist(), "nnz": allv[:, 9].astype(int).tolist(),
                           "spmm_fwd1_us": allv[:, 7].tolist(), "spmm_fwd1_roofline_frac": allv[:, 6].tolist()}
        out["comm_share"] = float(allv[:, 1].max() / max(allv[:, 0].max(), 1e-9))
        out["xgmi"] = {"link_GBs": XGMI_LINK_GBS, "halo_exchanges_per_step": hops_a + hops_t,
                       "bytes_received_per_hop_by_rank_a": allv[:, 2].astype(int).tolist(), "bytes_received_per_hop_by_rank_at": allv[:, 3].astype(int).tolist(),
                       "max_pair_bytes_per_hop_a": int(allv[:, 4].max()), "max_pair_bytes_per_hop_at": int(allv[:, 5].max()),
                       "ideal_exchange_us_per_step": ideal_us,
                       "measured_comm_us_per_step_max": float(allv[:, 1].max() * 1e3),
                       "frac_of_link_rate": (ideal_us / (allv[:, 1].max() * 1e3)) if allv[:, 1].max() > 0 else None,
                       "note": "comm classes = pack kernels + the grouped ncclSend/ncclRecv of every halo hop + the two all-reduces of a step "
                               "([E_B | P_B | inv_B] of the batch, the four weight gradients), event-bracketed on each rank's stream: they "
                               "include the wait for the slowest peer"}
        lh = engine.lazy_halo_rows()
        if lh[3] >= 0:
            # knob lazy_halo (automatic from 262,144 nodes): the backward hop A_hat^T u fetched only the rows of u that can be non-zero
            rows = all_ranks([lh[3], lh[5]])
            out["xgmi"]["subset_exchange_u"] = {"rows_fetched_last_step_by_rank": rows[:, 0].astype(int).tolist(),
                                                "rows_of_the_whole_halo_by_rank": rows[:, 1].astype(int).tolist(),
                                                # the sender-driven subset hop drains the stream once per step for the counts (gss_comm_sync + the
                                                # 144-byte D2H copy): inside ms_per_step, not inside the comm brackets; off over RCCL by default
                                                "stream_drains_per_step": 1,
                                                "note": "ideal_exchange_us_per_step above is priced on whole halos"}
        fa, ft = shard.layout.halo_fraction()
        halo_info.update({"halo_fraction_a": fa, "halo_fraction_at": ft})
    if halo_info is not None:
        # what this rank exchanges per SpMM hop (rank 0's shard; shards are nnz-balanced, so their row counts differ)
        out["shard"] = halo_info

    if single and args.cache_layer1 and not from_source:
        from gcn_drug_repurposing_amd.engine import GssEngine
        eng2 = GssEngine(graph, feats, [p.clone() for p in params], num_layers=L, layer_decay=decay, alpha=alpha, lr=lr,
                         max_batch=B, cache_layer1=True)
        for s in range(args.warmup):
            eng2.step(idx_all, beta, count=int(offs[s + 1] - offs[s]), offset=int(offs[s]))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for s in range(args.warmup, args.warmup + args.steps):
            eng2.step(idx_all, beta, count=int(offs[s + 1] - offs[s]), offset=int(offs[s]))
        torch.cuda.synchronize()
        out["ms_per_step_layer1_cached"] = (time.perf_counter() - t1) / args.steps * 1e3

    # ---- CPU baseline leg (rank 0, N=1 only): the reference's torch-CPU op sequence on the host cores ----
    if rank == 0 and world == 1 and not from_source and not args.no_cpu_baseline:
        from oracle import gss_oracle as O
        from oracle.torch_cpu_path import TorchCpuPath
        a_hat, _ = O.preprocess_graph(adj)
        cpu = TorchCpuPath(O.to_fp32_csr(a_hat), x_host, params_host, L, decay, alpha, lr)
        host_cpus = os.cpu_count() or 1
        # torch's default thread count on a 256-thread host may be oversubscribed or starved: a short sweep picks the count that is
        # fastest on THIS box, the bounded sample then runs at that count
        sweep = {}
        cand = sorted({t for t in (8, 16, 32, 64, 128) if t <= host_cpus} | {min(host_cpus, torch.get_num_threads())})
        for t in cand:
            torch.set_num_threads(t)
            ts = cpu.time_steps([b.astype(np.int64) for b in batches[:3]], beta, warmup=1)
            sweep[t] = float(np.median(ts))
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        cpu_steps = max(2, args.cpu_steps)
        ts = cpu.time_steps([b.astype(np.int64) for b in batches[:cpu_steps + 1]], beta, warmup=1)
        t_step = float(np.median(ts))
        t_spmm = cpu.time_spmm(3)
        out["cpu_baseline"] = {"value": spmm_per_step * nnz / t_step, "unit": "edges/s", "cores": best,
                               "kind": "port",
                               "sample": f"{len(ts)} timed steps (median) of the same workload after 1 warm-up at the fastest thread count of a "
                                         f"sweep ({best} threads), torch {torch.__version__} CPU, torch.sparse.mm COO fp32",
                               "s_per_step": t_step, "epoch_time_s": t_step * steps_per_epoch,
                               "thread_sweep_s_per_step": {str(k): v for k, v in sweep.items()},
                               "spmm_kernel_edges_per_s": nnz / t_spmm, "host_cpus": host_cpus}

    if rank == 0:
        emit(out)
    if job_comm is not None:
        job_comm.sync(COLLECTIVE_TIMEOUT_S)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
