#!/usr/bin/env python3
"""Experiment (SpMM in the HBM regime): does hub-first relabelling + non-temporal gathers of the cold rows raise the hit
rate of the hubs' rows?   usage: spmm_hot_cold.py <nodes> <edges> [d] [slices]
Prints the time of y = A x for: the generator's node order; nodes relabelled by descending in-degree; the same with rows
beyond the H hottest fetched non-temporally, for several H."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gcn_drug_repurposing_amd as pkg  # noqa: E402
from gcn_drug_repurposing_amd import _lib  # noqa: E402
from gcn_drug_repurposing_amd.dist import local_comms  # noqa: E402
from gcn_drug_repurposing_amd.graph import DeviceCSR  # noqa: E402
from gcn_drug_repurposing_amd.shards import RmatSource  # noqa: E402

lib = pkg.load()
n, m = int(sys.argv[1]), int(sys.argv[2])
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
slices = int(sys.argv[4]) if len(sys.argv) > 4 else 0
src = RmatSource(n, m, seed=4)
src.prepare(local_comms(1)[0])
rowptr, col, _ = src.rows(0, n, "cuda")
src.release()
nnz = col.numel()
val = torch.full((nnz,), 1.0 / 21.0, dtype=torch.float32, device="cuda")
x = torch.randn(n, d, device="cuda")
y = torch.empty(n, d, device="cuda")
st = _lib.current_stream()


def bench(csr, xin, label, reps=5):
    for _ in range(2):
        _lib.check(lib.gss_spmm(csr.handle, d, xin.data_ptr(), y.data_ptr(), None, None, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(lib.gss_spmm(csr.handle, d, xin.data_ptr(), y.data_ptr(), None, None, st))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{label:66s} {us:10.1f} us   gather {nnz * d * 4 / us / 1e6:6.2f} TB/s", flush=True)
    return y.clone()


if slices:
    lib.gss_debug_set_option(b"spmm_slices", slices)
base = DeviceCSR(rowptr.cpu().numpy(), col, val, n, n, "cuda")
y0 = bench(base, x, "generator order")
# hub-first relabelling: new id = rank by descending in-degree (stable)
indeg = torch.bincount(col.long(), minlength=n)
perm = torch.sort(indeg, descending=True, stable=True).indices          # perm[new] = old
inv = torch.empty_like(perm)
inv[perm] = torch.arange(n, device="cuda")
rows_old = torch.repeat_interleave(torch.arange(n, device="cuda"), (rowptr[1:] - rowptr[:-1]).long())
key = inv[rows_old] * n + inv[col.long()]
key = torch.sort(key).values
new_rows, new_cols = key // n, (key % n).to(torch.int32)
rp = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
rp[1:] = torch.cumsum(torch.bincount(new_rows, minlength=n), 0)
del key, rows_old, new_rows
hub = DeviceCSR(rp.to(torch.int32).cpu().numpy(), new_cols, val, n, n, "cuda")
xp = x[perm].contiguous()
y1 = bench(hub, xp, "hub-first relabelling")
err = (y1 - y0[perm]).abs().max().item()
print(f"   max |difference| after undoing the permutation: {err:.2e}")
cs = torch.cumsum(torch.sort(indeg, descending=True).values.double(), 0) / float(nnz)
for h in (4096, 8192, 16384, 65536, 262144, 500000):
    if h >= n:
        continue
    lib.gss_debug_set_option(b"spmm_hot_rows", h)
    bench(hub, xp, f"hub-first + cold rows non-temporal, H = {h} ({cs[h - 1].item():.2f} of the gathers hot)")
lib.gss_debug_set_option(b"spmm_hot_rows", 0)
# deeper gather queues and feature slicing on the relabelled graph
# (8 gathers in flight -- knob spmm_fly, removed in round 6 -- measured 3-9 % slower here: profiles/r02_spmm_hot_cold_rmat10m.txt)
for ns in (1, 2, 4):
    lib.gss_debug_set_option(b"spmm_slices", ns)
    for h in (0, 65536):
        lib.gss_debug_set_option(b"spmm_hot_rows", h)
        bench(hub, xp, f"hub-first, 4 gathers in flight, {ns} time-separated slice(s), H = {h}")
lib.gss_debug_set_option(b"spmm_hot_rows", 0)
lib.gss_debug_set_option(b"spmm_slices", 0)
