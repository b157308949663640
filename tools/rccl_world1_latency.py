#!/usr/bin/env python3
"""Round 6: what one collective costs beyond its bytes, measured instead of assumed.

The one-GPU box has one real RCCL rank.  A world-1 communicator (ncclCommInitRank with nranks = 1) runs the collectives of the sharded step
through the same enqueue path as a real job -- the host-side group bookkeeping, the kernel RCCL launches for an all-reduce (a local copy
kernel at world 1), the completion on the caller's stream -- without any peer: a LOWER BOUND of the per-collective cost between devices
(no rendezvous, no link latency).  tools/scaling_forecast.py reads the JSON this prints (profiles/r06_rccl_world1_latency.txt holds the text).

Timed, 1,000 repetitions each, on the caller's stream:
  * gss_allreduce_sum over the four weight gradients' 2 (d^2 + d) floats (132 KB at d = 128) and over the batch rows' B (2 d + 1) floats;
  * an empty grouped gss_exchange_rows (every pair's list empty: what a world-1 halo exchange is);
  * per repetition: host wall time from the call to the completion of the stream (enqueue -> completion) and the device time between two
    events recorded around the call; and the back-to-back rate (1,000 enqueues, one wait).
usage: rccl_world1_latency.py [d] [B] [reps] [--json out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gcn_drug_repurposing_amd as pkg  # noqa: E402
from gcn_drug_repurposing_amd.dist import rccl_comm  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
d = int(args[0]) if len(args) > 0 else 128
B = int(args[1]) if len(args) > 1 else 2048
reps = int(args[2]) if len(args) > 2 else 1000
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
pkg.load()
torch.cuda.set_device(0)
fd = os.dup(1)
os.dup2(2, 1)                       # RCCL prints its banner to the C stdout
comm = rccl_comm(1, 0)
os.dup2(fd, 1)
grads = torch.randn(2 * (d * d + d), device="cuda")
batch = torch.randn(B * (2 * d + 1), device="cuda")
send, recv = torch.zeros(16, d, device="cuda"), torch.zeros(16, d, device="cuda")
zero_off = np.zeros(2, dtype=np.int64)
cases = {
    f"allreduce_weight_gradients_{grads.numel() * 4}_bytes": lambda: comm.all_reduce_sum_(grads),
    f"allreduce_batch_rows_{batch.numel() * 4}_bytes": lambda: comm.all_reduce_sum_(batch),
    "exchange_rows_empty_group": lambda: comm.exchange_rows(d, send, zero_off, recv, zero_off),
}
# reference: the same bracket around a trivial kernel (what a launch + completion costs on this box without RCCL)
tiny = torch.zeros(64, device="cuda")
cases["trivial_kernel_for_reference"] = lambda: tiny.add_(1.0)
res = {}
for name, fn in cases.items():
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    wall, dev = [], []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        fn()
        e1.record()
        torch.cuda.current_stream().synchronize()
        wall.append((time.perf_counter() - t0) * 1e6)
        dev.append(e0.elapsed_time(e1) * 1e3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t0) / reps * 1e6
    wall, dev = np.array(wall), np.array(dev)
    res[name] = {"enqueue_to_completion_us": {"median": float(np.median(wall)), "mean": float(wall.mean()), "p10": float(np.percentile(wall, 10)),
                                              "p90": float(np.percentile(wall, 90))},
                 "device_event_bracket_us": {"median": float(np.median(dev)), "mean": float(dev.mean()), "p90": float(np.percentile(dev, 90))},
                 "back_to_back_us_per_call": b2b, "reps": reps}
    print(f"{name:48s} enqueue -> completion {np.median(wall):7.1f} us (median; p10 {np.percentile(wall, 10):.1f}, p90 {np.percentile(wall, 90):.1f})   "
          f"device bracket {np.median(dev):7.1f} us   back to back {b2b:7.1f} us per call", flush=True)
ref = res["trivial_kernel_for_reference"]["enqueue_to_completion_us"]["median"]
floor = {k: v["back_to_back_us_per_call"] for k, v in res.items() if k != "trivial_kernel_for_reference"}
summary = {"what": "world-1 RCCL communicator on one MI355X: a LOWER BOUND of the per-collective cost between devices (no peer, no rendezvous, no link)",
           "d": d, "B": B, "rccl_ranks": comm.count(), "cases": res,
           "per_collective_floor_us": {"allreduce": max(v for k, v in floor.items() if k.startswith("allreduce")),
                                       "exchange": floor["exchange_rows_empty_group"]},
           "note": "per_collective_floor_us = what one collective adds to a stream of enqueued work at world 1 (back-to-back rate; the event bracket around a single "
                   "call is dominated by the bracket's own cost: see the trivial kernel).  enqueue_to_completion includes the host's launch + wait (trivial kernel: "
                   "%.1f us).  A collective between devices adds a rendezvous and a link round trip on top, which a one-GPU box cannot measure." % ref}
print(json.dumps(summary))
if out_json:
    json.dump(summary, open(out_json, "w"), indent=1)
