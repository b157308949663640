#!/usr/bin/env python3
"""Round 6 experiment: can ONE pass of y = A_hat x in the HBM regime keep more of the hubs' rows in the L2s?

Part 1 (no new code): the launch policies the kernel already has, at BASELINE config 5's size -- feature slices of 256 B (2) or 128 B
(4), time-separated (grid.y) or pinned to XCDs (slice = workgroup id mod ns), with the H hottest rows declared hot (default policy) and
every other row fetched non-temporally.  An L2 of 4 MB holds 16k rows' 256-B slices or 32k rows' 128-B slices; the index arrays are
read once per slice (1.68 GB each).
Part 2: the cold tail in a LOCALITY order instead of the degree order.  Nodes [0, H) keep their hub-first ids; the others are renumbered by
(length class, smallest cold column of their row): rows that share their most popular cold neighbour become neighbours in the schedule, so
that neighbour's row is fetched once per cluster instead of once per reference.  The schedule sorts rows by block size and length / q
(knob spmm_seg_len_q), then by id.

usage: spmm_rmat_sweep.py <nodes> <edges> [d] [part1|part2|all]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gcn_drug_repurposing_amd as pkg  # noqa: E402
from gcn_drug_repurposing_amd import _lib  # noqa: E402
from gcn_drug_repurposing_amd.dist import local_comms  # noqa: E402
from gcn_drug_repurposing_amd.graph import DeviceCSR  # noqa: E402
from gcn_drug_repurposing_amd.shards import HOT_ROWS, RmatSource, build_shard  # noqa: E402

lib = pkg.load()
n, m = int(sys.argv[1]), int(sys.argv[2])
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
part = sys.argv[4] if len(sys.argv) > 4 else "all"
st = _lib.current_stream()


def opt(name, v):
    _lib.check(lib.gss_debug_set_option(name.encode(), int(v)), name)


g = build_shard(RmatSource(n, m, seed=4), local_comms(1)[0], need_transpose=False)
a = g.a
nnz = a.nnz
x = torch.randn(n, d, device="cuda")
y = torch.empty(n, d, device="cuda")
alg = 8 * nnz + 4 * (n + 1) + 8 * n * d
print(f"RMAT {n} nodes / {m} edges: nnz(A_hat) = {nnz}, d = {d}, algorithmic bytes per product {alg / 1e9:.2f} GB", flush=True)


def timed(csr, xin, reps=5):
    for _ in range(2):
        _lib.check(lib.gss_spmm(csr.handle, d, xin.data_ptr(), y.data_ptr(), None, None, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(lib.gss_spmm(csr.handle, d, xin.data_ptr(), y.data_ptr(), None, None, st))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def line(label, us):
    print(f"{label:100s} {us:9.1f} us   alg {alg / us / 1e6:6.3f} TB/s = {alg / us / 1e6 / 8.0:.3f} of 8 TB/s", flush=True)


def policies(csr, xin, tag, combos):
    best = None
    for ns, pin, h in combos:
        opt("spmm_slices", ns)
        opt("spmm_pin", pin)
        opt("spmm_hot_rows", h)
        us = timed(csr, xin)
        line(f"{tag}{ns} slices of {d * 4 // ns:3d} B {'pinned to XCDs ' if pin else 'time-separated'}, H = {h:6d}", us)
        if best is None or us < best[0]:
            best = (us, ns, pin, h)
    opt("spmm_slices", 0)
    opt("spmm_pin", 0)
    opt("spmm_hot_rows", -1)
    return best


us0 = timed(a, x)
line(f"the product's default (2 time-separated slices of 256 B, H = {HOT_ROWS})", us0)

if part in ("part1", "all"):
    combos = [(ns, pin, h) for ns in (2, 4) for pin in (0, 1) for h in (8192, 16384, 32768, 65536, 131072)]
    combos += [(1, 0, 65536), (8, 1, 65536), (8, 1, 32768)]
    b = policies(a, x, "", combos)
    print(f"-> best launch policy: {b[1]} slices, pin = {b[2]}, H = {b[3]}: {b[0]:.1f} us ({(b[0] / us0 - 1) * 100:+.1f} % vs the default)", flush=True)

if part in ("part2", "all"):
    H = HOT_ROWS
    rp = a.rowptr.long()
    counts = rp[1:] - rp[:-1]
    col64 = a.col[:nnz].long()
    row_of = torch.repeat_interleave(torch.arange(n, device="cuda"), counts)
    big = torch.full((n,), n, dtype=torch.int64, device="cuda")
    key = big.scatter_reduce(0, row_of, torch.where(col64 >= H, col64, torch.full_like(col64, n)), reduce="amin", include_self=True)
    del row_of
    val = a.val[:nnz]
    for q in (1, 8, 4096):
        # the tail in (length class descending, key ascending, id) order; the hubs keep their ids
        cls = -(counts[H:] // q)
        order = torch.argsort(cls * (n + 1) + key[H:], stable=True)
        perm = torch.cat([torch.arange(H, device="cuda"), H + order])      # perm[new] = old
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(n, device="cuda")
        ncounts = counts[perm]
        nrp = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        nrp[1:] = torch.cumsum(ncounts, 0)
        nrow_of = torch.repeat_interleave(torch.arange(n, device="cuda"), ncounts)
        src = rp[perm[nrow_of]] + (torch.arange(nnz, device="cuda") - nrp[nrow_of])
        del nrow_of
        ncol = inv[col64[src]].to(torch.int32)
        nval = val[src].contiguous()
        del src
        opt("spmm_seg_len_q", q)
        loc = DeviceCSR(nrp.to(torch.int32).cpu().numpy(), ncol, nval, n, n, "cuda")
        xp = x[perm].contiguous()
        if q == 1:
            yref = torch.empty_like(y)
            _lib.check(lib.gss_spmm(a.handle, d, x.data_ptr(), yref.data_ptr(), None, None, st))
            torch.cuda.synchronize()
        policies(loc, xp, f"tail by (length / {q}, smallest cold column): ", [(2, 0, 65536), (4, 0, 32768), (4, 1, 32768), (4, 1, 65536)])
        if q == 1:
            err = (y - yref[perm]).abs().max().item() / yref.abs().max().item()
            print(f"   max rel difference to the degree order after undoing the permutation: {err:.1e}", flush=True)
        # the same schedule granularity WITHOUT the locality key (what the coarser length classes alone cost or gain)
        plain = DeviceCSR(a.h_indptr, a.col, a.val, n, n, "cuda")
        policies(plain, x, f"degree order, length classes of {q}: ", [(2, 0, 65536)])
        del loc, plain, ncol, nval, xp, perm, inv, order
        torch.cuda.empty_cache()
    opt("spmm_seg_len_q", 1)
