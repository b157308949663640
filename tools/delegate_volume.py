#!/usr/bin/env python3
"""Round 4 study: what HUB DELEGATION would move per SpMM hop, counted on the product's own partition.

With the hub-first order and nnz-balanced contiguous ranges, rank 0 owns the hubs: their ROWS reference a third of the graph (rank 0's
halo), and everybody's rows reference THEM.  Delegation treats the H hottest nodes (ids < H after relabelling) specially:
  * hub COLUMNS: their operand rows are replicated on every rank by one all-gather of H x d per hop -- entries with col < H leave
    every halo;
  * hub ROWS: an entry (i < H, j) is multiplied where column j lives -- every rank forms the partial sums of the H hub rows over its
    own columns and one all-reduce of H x d adds them up -- entries with row < H leave their owner's halo.
What remains in the halos is the cold-row x cold-column part of the matrix.  This tool builds every rank's rows exactly as
shards.build_shard does (RmatSource, Relabel, nnz-balanced ranges; `world` ranks as threads on one GPU) and counts, per rank, the boundary
rows of A_hat's halo today and with delegation at several H.  No plan, no step: evidence for DESIGN section 8, not a measurement of time.
usage: delegate_volume.py <nodes> <edges> <world> [d]"""
import json
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gcn_drug_repurposing_amd as pkg  # noqa: E402
from gcn_drug_repurposing_amd.dist import Partition, local_comms, nnz_balanced_ranges  # noqa: E402
from gcn_drug_repurposing_amd.shards import ROW_WEIGHT, Relabel, RmatSource  # noqa: E402

pkg.load()
n, m, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
d = int(sys.argv[4]) if len(sys.argv) > 4 else 128
HS = [h for h in (4096, 16384, 65536, 262144) if h < n]
comms = local_comms(world)
out, errors = [None] * world, []


def worker(rank):
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        with torch.cuda.stream(torch.cuda.Stream()):
            src = RmatSource(n, m, seed=4, device="cuda:0")
            work = np.asarray(src.work(comms[rank], dev), dtype=np.int64)
            rl = Relabel(work)
            work = work[rl.perm]
            part = Partition(nnz_balanced_ranges(np.concatenate([[0], np.cumsum(work + max(0, ROW_WEIGHT - 1))]), world))
            lo, hi = part.rows(rank)
            rowptr, col, _ = src.rows(lo, hi, dev, relabel=rl)
            counts = (rowptr[1:] - rowptr[:-1]).long()
            row = torch.repeat_interleave(torch.arange(lo, hi, device=dev), counts)
            col = col.long()
            remote = (col < lo) | (col >= hi)
            res = {"rank": rank, "rows": hi - lo, "entries": int(col.numel()), "halo_rows_now": int(torch.unique(col[remote]).numel())}
            for h in HS:
                keep = remote & (col >= h) & (row >= h)
                res[f"halo_rows_H{h}"] = int(torch.unique(col[keep]).numel())
                res[f"delegated_entries_H{h}"] = int(((row < h) & remote).sum().item())
            src.release()
            out[rank] = res
            torch.cuda.current_stream().synchronize()
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
row_mb = d * 4 / 2 ** 20
summary = {"nodes": n, "edges": m, "world": world, "d": d,
           "note": "boundary rows of A_hat's halo per rank and hop: today, and with the H hottest nodes delegated (their operand rows all-gathered, "
                   "their rows' products formed at the column owners and all-reduced: 2 x H x d x 4 bytes per hop and rank on top of the halo)",
           "today": {"halo_mb_per_hop_by_rank": [round(o["halo_rows_now"] * row_mb, 1) for o in out],
                     "max_mb": round(max(o["halo_rows_now"] for o in out) * row_mb, 1), "sum_mb": round(sum(o["halo_rows_now"] for o in out) * row_mb, 1)}}
for h in HS:
    halos = [o[f"halo_rows_H{h}"] for o in out]
    summary[f"H{h}"] = {"halo_mb_per_hop_by_rank": [round(v * row_mb, 1) for v in halos], "max_mb": round(max(halos) * row_mb, 1),
                        "sum_mb": round(sum(halos) * row_mb, 1), "hub_collectives_mb_per_hop_and_rank": round(2 * h * row_mb, 1),
                        "max_mb_incl_hub_collectives": round(max(halos) * row_mb + 2 * h * row_mb, 1)}
summary["ranks"] = out
print(json.dumps(summary, indent=1))
