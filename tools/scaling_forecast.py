#!/usr/bin/env python3
"""Per-rank ISOLATED timing of the sharded step on a one-GPU box, and from it a FORECAST (never a measurement) of the 1 -> 8 GPU curve.

`world` ranks run as threads of this process on the in-process backend.  After the warm-up steps one full step and one lazy step are
RECORDED (gss_comm_local_mode 1: every collective keeps what it delivered); then each rank REPLAYS its two steps ALONE on the GPU (mode 2:
the recorded payloads are served by device copies, no peers, no barriers) with the plan's per-class profiler on.  Output per rank: kernel
ms by class for the full and the lazy step, the step's wall time alone, what every collective delivered; per job: the critical rank, the
xGMI floor of every collective (most loaded pair / 153 GB/s), a step forecast = slowest rank's kernels + collectives at their floors + a
stated latency per collective, and a least-squares fit of the kernel time to the partition's cost model (stored entries, own rows, boundary
rows projected under halo_recompute).

usage: scaling_forecast.py <workload> <world>[,<world>...] [--d 128] [--layers 2] [--batch 2048] [--reps 3] [--rccl-default 1]
workload: whole_graph | whole_graph_pathway | rmat:<nodes>:<edges>
--rccl-default 1 (the default): the knobs an RCCL job gets (lazy_halo by graph size, lazy_halo_u = 0: u is never exchanged as a subset)."""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gcn_drug_repurposing_amd as pkg  # noqa: E402
from gcn_drug_repurposing_amd import _lib, synth  # noqa: E402
from gcn_drug_repurposing_amd.dist import local_comms  # noqa: E402
from gcn_drug_repurposing_amd.shards import RmatSource, ScipySource, build_shard, gaussian_rows, shard_engine  # noqa: E402

XGMI_LINK_GBS = 153.0
# cost of one collective between devices beyond its bytes.  Round 6: MEASURED on the box's one real RCCL rank (tools/rccl_world1_latency.py ->
# profiles/r06_rccl_world1_latency.json): the device-side bracket around one all-reduce / one grouped exchange of a world-1 communicator on the
# caller's stream -- a LOWER BOUND (no peer, no rendezvous, no link latency).  Without that file: the 25 us rounds 4-5 assumed.
LATENCY_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_rccl_world1_latency.json")
if os.path.exists(LATENCY_FILE):
    _lat = json.load(open(LATENCY_FILE))["per_collective_floor_us"]
    LATENCY_US = {"exchange": float(_lat["exchange"]), "allreduce": float(_lat["allreduce"])}
    LATENCY_SOURCE = "measured: world-1 RCCL communicator on one MI355X, device-side bracket per collective (profiles/r06_rccl_world1_latency.json) -- a lower bound"
else:
    LATENCY_US = {"exchange": 25.0, "allreduce": 25.0}
    LATENCY_SOURCE = "ASSUMED 25 us per collective (no measurement on file)"
KERNEL_CLASSES = ("spmm_fwd_hadamard", "spmm_fwd", "spmm_bwd1", "spmm_bwd2", "spmm_bwd1_dense", "spmm_bwd2_dense", "dense_fwd", "dgrad", "wgrad", "wgrad_batch", "loss", "elementwise", "rownorm", "adam")

ap = argparse.ArgumentParser()
ap.add_argument("workload")
ap.add_argument("worlds")
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--layers", type=int, default=2)
ap.add_argument("--batch", type=int, default=2048)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--rccl-default", type=int, default=1)
ap.add_argument("--lazy-halo", type=int, default=None, help="knob lazy_halo (over RCCL the subset exchange of the lazy step is opt-in: --rccl-default 1 sets 0 unless this says 1)")
ap.add_argument("--cache-layer1", type=int, default=0, help="1: the plans keep layer 1's two SpMM results (their inputs are constants) -- what train.py runs; its lazy step is then the trainer's step")
ap.add_argument("--row-weight", type=int, default=None, help="the partition's cost of a row besides its entries (default: shards.row_weight_for(d, layers), what train.py and bench.py pass)")
args = ap.parse_args()
lib = pkg.load()
if args.rccl_default:
    assert lib.gss_debug_set_option(b"lazy_halo_u", 0) == 0
    assert lib.gss_debug_set_option(b"lazy_halo", 1 if args.lazy_halo == 1 else 0) == 0     # (an RCCL job's default since round 6: off unless asked for)
elif args.lazy_halo is not None:
    assert lib.gss_debug_set_option(b"lazy_halo", args.lazy_halo) == 0
d, L, B = args.d, args.layers, args.batch

if args.workload.startswith("rmat:"):
    n, m = (int(v) for v in args.workload.split(":")[1:])
    make_source = lambda: RmatSource(n, m, seed=4, device="cuda:0")        # noqa: E731
    feats = lambda lo, hi: gaussian_rows(lo, hi, d, 5)                      # noqa: E731
else:
    adj = synth.whole_graph_standin(seed=1, pathway_edges=args.workload == "whole_graph_pathway")[0]
    n = adj.shape[0]
    X = synth.gaussian_features(n, d, seed=2 if d == 128 else 3)
    make_source = lambda: ScipySource(adj)                                 # noqa: E731
    feats = None
np.random.seed(7)
w = np.random.randn(d, d) * 1e-5
np.fill_diagonal(w, 1.0)
params = {"W1": w.astype(np.float32), "b1": np.zeros(d, np.float32), "W2": w.astype(np.float32).copy(), "b2": np.zeros(d, np.float32)}
rng = np.random.RandomState(1234)
batches = [rng.permutation(n)[:B].astype(np.int32) for _ in range(6)]


def run_world(world):
    comms = local_comms(world)
    out, errors = [None] * world, []
    gate = threading.Barrier(world)
    turn = threading.Condition()
    state = {"turn": 0}

    def worker(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                from gcn_drug_repurposing_amd.shards import shard_rows
                from gcn_drug_repurposing_amd.shards import row_weight_for
                kw = {"row_weight": row_weight_for(d, L) if args.row_weight is None else args.row_weight}
                shard = build_shard(make_source(), comms[rank], need_transpose=L > 1, device="cuda:0", **kw)
                lo, hi = shard.part.rows(rank)
                x_loc = feats(lo, hi) if feats else shard_rows(shard, X)
                eng = shard_engine(shard, x_loc, params, comms[rank], num_layers=L, layer_decay=0.3, alpha=1.0, lr=3e-4, max_batch=B,
                                   cache_layer1=bool(args.cache_layer1))
                idx = [torch.from_numpy(b).cuda() for b in batches]
                sync = torch.cuda.current_stream().synchronize
                for k in range(2):
                    eng.step(idx[k], 0.25)
                eng.step_lazy(idx[2], 0.25)
                sync()
                # all ranks together (NOT a performance figure: one GPU, host-synchronised copies)
                gate.wait()
                t0 = time.perf_counter()
                eng.step(idx[3], 0.25)
                sync()
                together_full = (time.perf_counter() - t0) * 1e3
                # ---- record one full + one lazy step
                gate.wait()
                comms[rank].local_mode(1)
                eng.comm_stats(); eng.sync_stats()
                eng.step(idx[4], 0.25)
                sync()
                n_full = len(comms[rank].local_log())
                coll_full, sync_full = eng.comm_stats(), eng.sync_stats()
                eng.step_lazy(idx[5], 0.25)
                sync()
                coll_lazy, sync_lazy = eng.comm_stats(), eng.sync_stats()
                log = comms[rank].local_log()
                lh = eng.lazy_halo_rows()
                gate.wait()
                # ---- replay alone, rank by rank
                with turn:
                    turn.wait_for(lambda: state["turn"] == rank)
                acc = {"full": {}, "lazy": {}}
                wall = {"full": [], "lazy": []}
                for rep in range(args.reps + 1):
                    comms[rank].local_mode(2)
                    eng.profile(True)
                    for kind, call, b in (("full", eng.step, idx[4]), ("lazy", eng.step_lazy, idx[5])):
                        sync()
                        t0 = time.perf_counter()
                        call(b, 0.25)
                        sync()
                        dt = (time.perf_counter() - t0) * 1e3
                        prof = eng.profile_read()
                        if rep == 0:
                            continue              # the first replay warms the caches
                        wall[kind].append(dt)
                        for cls, (ms, cnt) in prof.items():
                            a = acc[kind].setdefault(cls, [0.0, 0])
                            a[0] += ms
                            a[1] += cnt
                    eng.profile(False)
                comms[rank].local_mode(0)
                with turn:
                    state["turn"] += 1
                    turn.notify_all()
                per = lambda kind: {cls: round(v[0] / args.reps, 4) for cls, v in acc[kind].items() if v[1]}     # noqa: E731
                ha, hat = shard.layout.halo_a, shard.layout.halo_at
                pair_a = int(np.diff(ha.recv_off).max()) if world > 1 else 0
                out[rank] = dict(rank=rank, rows=hi - lo, nnz=int(shard.a.nnz), nnz_t=int(shard.at.nnz) if shard.at is not None else 0,
                                 halo_rows_a=int(ha.n_halo), halo_rows_at=int(hat.n_halo) if hat is not None else 0,
                                 max_pair_rows_a=pair_a, plan_gb=round(eng.device_bytes() / 2 ** 30, 2),
                                 kernel_ms_full=per("full"), kernel_ms_lazy=per("lazy"),
                                 kernels_ms_full=round(sum(v for k, v in per("full").items() if k in KERNEL_CLASSES), 4),
                                 kernels_ms_lazy=round(sum(v for k, v in per("lazy").items() if k in KERNEL_CLASSES), 4),
                                 wall_alone_ms_full=round(float(np.median(wall["full"])), 4), wall_alone_ms_lazy=round(float(np.median(wall["lazy"])), 4),
                                 together_ms_full=round(together_full, 3),
                                 delivered_bytes_full=log[:n_full], delivered_bytes_lazy=log[n_full:],
                                 collectives_full=list(coll_full), collectives_lazy=list(coll_lazy), host_syncs_full=list(sync_full), host_syncs_lazy=list(sync_lazy),
                                 lazy_m_rows_fetched=lh[0], u_rows_fetched=lh[3], exchange_free_last_hop=bool(getattr(shard.layout, "a_loc_t", None) is not None and lh[3] < 0))
        except Exception as e:  # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))
            comms[rank].abort()
            gate.abort()
            with turn:
                state["turn"] = 10 ** 6
                turn.notify_all()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if errors:
        print(errors[0][2], file=sys.stderr)
        raise SystemExit(1)
    return out


def floor_us(bytes_in, rows_total, max_pair_rows):
    """a collective is done when its most loaded pair is (one xGMI link per pair): the rank's delivered bytes x the largest pair's share of
    its halo; without a halo (an all-reduce) the bytes over one link"""
    share = (max_pair_rows / rows_total) if rows_total else 1.0
    return bytes_in * share / (XGMI_LINK_GBS * 1e9) * 1e6


from gcn_drug_repurposing_amd.shards import row_weight_for as _rwf  # noqa: E402
res = {"workload": args.workload, "n": n, "d": d, "layers": L, "batch": B, "reps": args.reps, "cache_layer1": bool(args.cache_layer1), "row_weight": _rwf(d, L) if args.row_weight is None else args.row_weight, "rccl_default_knobs": bool(args.rccl_default),
       "latency_us_per_collective": LATENCY_US, "latency_source": LATENCY_SOURCE, "lazy_halo": args.lazy_halo, "xgmi_link_GBs": XGMI_LINK_GBS,
       "what_this_is": "a FORECAST from per-rank kernel times measured with each rank alone on ONE MI355X (recorded exchange payloads replayed by device "
                       "copies) + collectives priced at the xGMI floor of their most loaded pair + a per-collective latency (see latency_source); no two-device run exists",
       "worlds": {}}
fit_rows = []
for world in [int(v) for v in args.worlds.split(",")]:
    ranks = run_world(world)
    torch.cuda.empty_cache()
    entry = {"ranks": ranks}
    for kind in ("full", "lazy"):
        kt = np.array([r[f"kernels_ms_{kind}"] for r in ranks])
        coll = []
        ncoll = max(len(r[f"delivered_bytes_{kind}"]) for r in ranks)
        for c in range(ncoll):
            fl = [floor_us(r[f"delivered_bytes_{kind}"][c], r["halo_rows_a"], r["max_pair_rows_a"]) if c < len(r[f"delivered_bytes_{kind}"]) else 0.0 for r in ranks]
            coll.append({"max_delivered_bytes": max((r[f"delivered_bytes_{kind}"][c] if c < len(r[f"delivered_bytes_{kind}"]) else 0) for r in ranks),
                         "xgmi_floor_us": round(max(fl), 2)})
        # latency: per COLLECTIVE the plan enqueued (gss_plan_comm_stats; the four weight gradients are one fused all-reduce but four
        # deliveries in the log), + the request phase's bitmap exchange where the lazy step's subset exchange is on
        n_launch = max(sum(r[f"collectives_{kind}"]) for r in ranks)
        n_exch = max(r[f"collectives_{kind}"][0] for r in ranks)            # (boundary-row exchanges, batch all-reduces, weight-gradient all-reduces)
        lat_us = LATENCY_US["exchange"] * n_exch + LATENCY_US["allreduce"] * (n_launch - n_exch)
        comm_us = sum(c["xgmi_floor_us"] for c in coll) + (lat_us if world > 1 else 0.0)
        comm_us_25 = sum(c["xgmi_floor_us"] for c in coll) + (25.0 * n_launch if world > 1 else 0.0)   # the constant rounds 4-5 assumed, for comparison
        entry[kind] = {"kernel_ms_by_rank": kt.round(4).tolist(), "critical_rank": int(kt.argmax()), "kernel_ms_max": round(float(kt.max()), 4),
                       "kernel_ms_mean": round(float(kt.mean()), 4), "imbalance_max_over_mean": round(float(kt.max() / kt.mean()), 3),
                       "collectives": coll, "collectives_enqueued": n_launch, "collectives_us_at_floor_plus_latency": round(comm_us, 1),
                       "forecast_ms_per_step_no_overlap": round(float(kt.max()) + comm_us * 1e-3, 4),
                       "forecast_ms_per_step_at_25us_per_collective": round(float(kt.max()) + comm_us_25 * 1e-3, 4)}
    res["worlds"][str(world)] = entry
    for r in ranks:
        boundary = r["halo_rows_a"] if L > 1 and world > 1 else 0
        fit_rows.append((r["nnz"] + r["nnz_t"], r["rows"], boundary, r["kernels_ms_full"], world))
base = res["worlds"].get("1")
for wkey, e in res["worlds"].items():
    if base:
        for kind in ("full", "lazy"):
            e[kind]["forecast_speedup_vs_world1"] = round(base[kind]["forecast_ms_per_step_no_overlap"] / e[kind]["forecast_ms_per_step_no_overlap"], 3)
# the partition's cost model against the measured kernel times: ms ~ a * entries + b * own rows + c * boundary rows (+ e)
A = np.array([[r[0], r[1], r[2], 1.0] for r in fit_rows], dtype=np.float64)
y = np.array([r[3] for r in fit_rows])
if len(fit_rows) >= 5:
    coef, *_ = np.linalg.lstsq(A, y, rcond=None)
    pred = A @ coef
    res["cost_model_fit"] = {"model": "kernel ms of a full step ~ a * (entries of the shard's A_hat and A_hat^T) + b * own rows + c * boundary rows + e",
                             "a_ns_per_entry": round(coef[0] * 1e6, 4), "b_ns_per_row": round(coef[1] * 1e6, 3), "c_ns_per_boundary_row": round(coef[2] * 1e6, 3),
                             "e_ms": round(coef[3], 4), "row_weight_implied_entries_per_row": round(coef[1] / coef[0], 1) if coef[0] > 0 else None,
                             "boundary_row_weight_implied": round(coef[2] / coef[0], 1) if coef[0] > 0 else None,
                             "max_rel_residual": round(float(np.abs(pred - y).max() / y.max()), 3), "points": len(fit_rows)}
print(json.dumps(res, indent=1))
