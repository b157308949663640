#!/usr/bin/env python3
"""A sharded job on a ONE-GPU box: `world` ranks as threads of this process (gss_comm_create_local), each building only its
own rows of an RMAT graph (shards.RmatSource / build_shard) and running the native sharded plan.
usage: shard_emulation.py <nodes> <edges> <world> [steps] [d] [split: auto | 0 | 1] [lazy_halo: -1 | 0 | 1] [halo_recompute: -1 | 0 | 1] [lazy_halo_u: -1 | 0 | 1]
(lazy_halo_u = 0 with everything else automatic = exactly the knobs an RCCL job gets: the in-process backend would otherwise also take the
sender-driven subset exchange of u, which RCCL jobs leave off)
Reports per rank: rows, stored entries, boundary rows per hop (halo) and their fraction of the other shards' rows, plan
bytes; for the job: host peak RSS, setup time, ms/step (NOT a performance figure: the ranks share one GPU and the exchanges
are host-synchronised copies), and the loss after the steps -- compare it with the world = 1 run of the same command.  After the
full steps the same number of LAZY steps (what train.py runs): their time and the boundary rows of the top layer's M each rank fetched in the
last one, against the whole halo (knob lazy_halo, argument 7: 0 = always the whole halo)."""
import json
import os
import resource
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gcn_drug_repurposing_amd as pkg  # noqa: E402
from gcn_drug_repurposing_amd.dist import local_comms  # noqa: E402
from gcn_drug_repurposing_amd.shards import RmatSource, build_shard, gaussian_rows, shard_engine  # noqa: E402

pkg.load()
n, m, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
d = int(sys.argv[5]) if len(sys.argv) > 5 else 128
split = {"auto": "auto", "0": False, "1": True}[sys.argv[6] if len(sys.argv) > 6 else "auto"]
lazy_halo = int(sys.argv[7]) if len(sys.argv) > 7 else -1
assert pkg.load().gss_debug_set_option(b"lazy_halo", lazy_halo) == 0
recompute = int(sys.argv[8]) if len(sys.argv) > 8 else -1
assert pkg.load().gss_debug_set_option(b"halo_recompute", recompute) == 0
lazy_halo_u = int(sys.argv[9]) if len(sys.argv) > 9 else -1
assert pkg.load().gss_debug_set_option(b"lazy_halo_u", lazy_halo_u) == 0
from gcn_drug_repurposing_amd.shards import row_weight_for  # noqa: E402
L, B = 2, 2048
np.random.seed(7)
w = np.random.randn(d, d) * 1e-5
np.fill_diagonal(w, 1.0)
params = {"W1": w.astype(np.float32), "b1": np.zeros(d, np.float32), "W2": w.astype(np.float32).copy(), "b2": np.zeros(d, np.float32)}
rng = np.random.RandomState(1234)
batches = [rng.permutation(n)[:B].astype(np.int32) for _ in range(2 * steps + 1)]
comms = local_comms(world)
out, errors = [None] * world, []
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss


def worker(rank):
    try:
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            t0 = time.perf_counter()
            shard = build_shard(RmatSource(n, m, seed=4, device="cuda:0"), comms[rank], need_transpose=True, device="cuda:0", split=split,
                                row_weight=row_weight_for(d, 2))
            lo, hi = shard.part.rows(rank)
            eng = shard_engine(shard, gaussian_rows(lo, hi, d, 5), params, comms[rank], num_layers=L, layer_decay=0.3, alpha=1.0, lr=3e-4, max_batch=B)
            torch.cuda.current_stream().synchronize()
            setup = time.perf_counter() - t0
            idx = [torch.from_numpy(b).cuda() for b in batches]
            eng.step(idx[0], 0.25)
            torch.cuda.current_stream().synchronize()
            eng.comm_stats()
            eng.sync_stats()
            t1 = time.perf_counter()
            for k in range(1, steps + 1):
                eng.step(idx[k], 0.25)
            torch.cuda.current_stream().synchronize()
            t_full = (time.perf_counter() - t1) / steps * 1e3
            loss_full = eng.loss.item()
            coll_full = [c / steps for c in eng.comm_stats()]
            sync_full = [c / steps for c in eng.sync_stats()]
            t2 = time.perf_counter()
            for k in range(steps + 1, 2 * steps + 1):
                eng.step_lazy(idx[k], 0.25)
            torch.cuda.current_stream().synchronize()
            t_lazy = (time.perf_counter() - t2) / steps * 1e3
            coll_lazy = [c / steps for c in eng.comm_stats()]
            sync_lazy = [c / steps for c in eng.sync_stats()]
            fetched, sent, _, u_fetched, u_sent, u_halo = eng.lazy_halo_rows()
            if recompute != 0 and L == 2 and coll_full[0] < 1.5:   # one boundary-row exchange per full step = M_1's: the plan took the exchange-free last hop
                u_fetched = u_sent = u_halo = 0       # the last backward hop runs on A_hat's shard transposed in place: u is not exchanged
            fa, ft = shard.layout.halo_fraction()
            out[rank] = dict(rank=rank, rows=hi - lo, nnz=shard.a.nnz, halo_rows_a=shard.layout.halo_a.n_halo, halo_rows_at=shard.layout.halo_at.n_halo,
                             halo_fraction_a=round(fa, 4), halo_fraction_at=round(ft, 4), send_rows_a=int(shard.layout.halo_a.send_off[-1]),
                             plan_gb=round(eng.device_bytes() / 2 ** 30, 2), setup_s=round(setup, 1), overlapped_hops=shard.layout.overlapped,
                             own_column_entries_a=(shard.split_a[0].nnz if shard.split_a else None),
                             own_column_entries_at=(shard.split_at[0].nnz if shard.split_at else None),
                             exchanged_mb_per_hop_a=round(shard.layout.halo_a.n_halo * d * 4 / 2 ** 20, 1),
                             exchanged_mb_per_hop_at=round(shard.layout.halo_at.n_halo * d * 4 / 2 ** 20, 1),
                             ms_per_step=round(t_full, 2), loss=loss_full, relabelled=shard.relabel is not None,
                             lazy_ms_per_step=round(t_lazy, 2), lazy_loss=eng.loss.item(),
                             lazy_top_m_rows_fetched=(fetched if fetched >= 0 else shard.layout.halo_a.n_halo),
                             lazy_top_m_rows_sent=(sent if sent >= 0 else int(shard.layout.halo_a.send_off[-1])),
                             lazy_top_m_mb_fetched=round((fetched if fetched >= 0 else shard.layout.halo_a.n_halo) * d * 4 / 2 ** 20, 2),
                             u_rows_fetched=(u_fetched if u_fetched >= 0 else u_halo), u_rows_sent=(u_sent if u_sent >= 0 else int(shard.layout.halo_at.send_off[-1])),
                             u_mb_fetched=round((u_fetched if u_fetched >= 0 else u_halo) * d * 4 / 2 ** 20, 2), lazy_halo=fetched >= 0,
                             # what a step moves INTO this rank: the hops' boundary rows (X_1 only without halo_recompute) + the batch rows
                             halo_recompute=recompute != 0, exchange_free_last_hop=bool(u_halo == 0),
                             x1_mb_fetched_per_step=0.0 if recompute != 0 else round(shard.layout.halo_a.n_halo * d * 4 / 2 ** 20, 1),
                             collectives_per_full_step=dict(boundary_row_exchanges=coll_full[0], batch_row_allreduces=coll_full[1], weight_gradient_allreduces=coll_full[2]),
                             host_waits_per_full_step=dict(stream_drains=sync_full[0], request_stream_event_waits=sync_full[1]),
                             host_waits_per_lazy_step=dict(stream_drains=sync_lazy[0], request_stream_event_waits=sync_lazy[1]),
                             collectives_per_lazy_step=dict(boundary_row_exchanges=coll_lazy[0], batch_row_allreduces=coll_lazy[1], weight_gradient_allreduces=coll_lazy[2]))
            o = out[rank]
            o["full_step_mb_received"] = round(o["x1_mb_fetched_per_step"] + o["exchanged_mb_per_hop_a"] + o["u_mb_fetched"], 1)
            o["lazy_step_mb_received"] = round(o["x1_mb_fetched_per_step"] + o["lazy_top_m_mb_fetched"] + o["u_mb_fetched"], 1)
    except Exception as e:  # noqa: BLE001
        import traceback
        errors.append((rank, repr(e), traceback.format_exc()))
        comms[rank].abort()


ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
[t.start() for t in ts]
[t.join() for t in ts]
if errors:
    print(errors[0][2])
    raise SystemExit(1)
peak = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
res = dict(nodes=n, edges=m, world=world, d=d, steps=steps, knobs=dict(split=str(split), lazy_halo=lazy_halo, halo_recompute=recompute, lazy_halo_u=lazy_halo_u), host_peak_rss_gb=round(peak / 2 ** 20, 2), host_rss_before_gb=round(rss0 / 2 ** 20, 2),
           host_rss_per_rank_gb=round((peak - rss0) / 2 ** 20 / world, 2), gpu_peak_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), ranks=out)
assert len({o["loss"] for o in out}) == 1 and len({o["lazy_loss"] for o in out}) == 1, "the replicas disagree on the loss"
print(json.dumps(res, indent=1))
