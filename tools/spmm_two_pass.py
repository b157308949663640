#!/usr/bin/env python3
"""Round 4 experiment: the SpMM of the HBM regime as TWO passes over a column split of A_hat.

At RMAT 10M / 200M one launch of y = A x moves 8.2 x its algorithmic bytes through the fabric: a feature row is referenced 21
times on average by rows scattered over the graph and 4 MB of L2 per XCD hold 8,192 whole rows.  The nodes are numbered hub-first,
so the H most referenced rows are operand rows [0, H); here the entries are split by column,
    hot  = entries with col <  H  (about half of all gathers at H = 65,536)
    cold = entries with col >= H,
and the product runs as gss_spmm(hot) + gss_spmm_add(cold) with a launch policy per pass: the hot pass with XCD-PINNED NARROW
feature slices -- slice s of every row is only ever gathered by XCD s mod 8, so an XCD's L2 holds H x slice bytes of the hub table
(H = 65,536 x 64 B = 4 MB: resident) --, the cold pass streaming with non-temporal gathers.

usage: spmm_two_pass.py <nodes> <edges> [d] sweep            time the single pass and every (H, slices) pair, check the results
       spmm_two_pass.py <nodes> <edges> [d] base <reps>      `reps` single-pass products   (for rocprofv3 --pmc)
       spmm_two_pass.py <nodes> <edges> [d] fwd1 <reps>      `reps` single-pass products with the Hadamard epilogue (AX = A x, M = AX (.) x)
       spmm_two_pass.py <nodes> <edges> [d] two <H> <slices> <reps>   `reps` two-pass products  (for rocprofv3 --pmc)
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
mode = sys.argv[4] if len(sys.argv) > 4 else "sweep"
st = _lib.current_stream()


def opt(name, v):
    _lib.check(lib.gss_debug_set_option(name.encode(), int(v)), name)


# the graph exactly as bench.py's rmat workload builds it: hub-first relabelled, normalised A_hat, the H hottest rows declared
g = build_shard(RmatSource(n, m, seed=4), local_comms(1)[0], need_transpose=False)
a = g.a
nnz = a.nnz
x = torch.randn(n, d, device="cuda")
y = torch.empty(n, d, device="cuda")
alg = 8 * nnz + 4 * (n + 1) + 8 * n * d


def split(h):
    """(hot CSR, cold CSR, share of the entries that are hot): same rows, a row's entries in their original order"""
    counts = (a.rowptr[1:] - a.rowptr[:-1]).long()
    row_of = torch.repeat_interleave(torch.arange(n, device="cuda"), counts)
    hot = a.col[:nnz].long() < h
    out = []
    for mask in (hot, ~hot):
        rp = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        rp[1:] = torch.cumsum(torch.bincount(row_of[mask], minlength=n), 0)
        out.append(DeviceCSR(rp.to(torch.int32).cpu().numpy(), a.col[:nnz][mask].contiguous(), a.val[:nnz][mask].contiguous(), n, n, "cuda"))
    share = float(hot.sum().item()) / nnz
    del row_of, hot
    return out[0], out[1], share


def single():
    _lib.check(lib.gss_spmm(a.handle, d, x.data_ptr(), y.data_ptr(), None, None, st))


def two(hot, cold, ns):
    # hot pass: every gather hits the hub table; narrow slices pinned to XCDs, plain loads
    opt("spmm_slices", ns)
    opt("spmm_pin", 1)
    opt("spmm_hot_rows", 0)
    _lib.check(lib.gss_spmm(hot.handle, d, x.data_ptr(), y.data_ptr(), None, None, st))
    # cold pass: every gather is a once-read row: the automatic slicing, all rows non-temporal
    opt("spmm_slices", 0)
    opt("spmm_pin", 0)
    opt("spmm_hot_rows", 1)
    _lib.check(lib.gss_spmm_add(cold.handle, d, x.data_ptr(), y.data_ptr(), y.data_ptr(), None, None, st))
    opt("spmm_hot_rows", -1)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if mode == "base":
    for _ in range(int(sys.argv[5])):
        single()
    torch.cuda.synchronize()
    print(f"base: n={n} nnz={nnz} d={d} alg={alg}")
elif mode == "fwd1":
    m_out = torch.empty(n, d, device="cuda")
    for _ in range(int(sys.argv[5])):
        _lib.check(lib.gss_spmm(a.handle, d, x.data_ptr(), y.data_ptr(), x.data_ptr(), m_out.data_ptr(), st))
    torch.cuda.synchronize()
    print(f"fwd1: n={n} nnz={nnz} d={d} alg={alg + 4 * n * d}")
elif mode == "two":
    h, ns, reps = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    hot, cold, share = split(h)
    for _ in range(reps):
        two(hot, cold, ns)
    torch.cuda.synchronize()
    print(f"two: n={n} nnz={nnz} d={d} alg={alg} H={h} slices={ns} hot_share={share:.3f}")
else:
    us = timed(single)
    print(f"single pass (hub-first, H = {HOT_ROWS} declared hot, the automatic slicing)   {us:10.1f} us   alg {alg / us / 1e6:6.3f} TB/s = "
          f"{alg / us / 1e6 / 8.0:.3f} of 8 TB/s", flush=True)
    ref = y.clone()
    for h in (16384, 32768, 65536, 131072):
        if h >= n:
            continue
        hot, cold, share = split(h)
        for ns in (2, 4, 8):
            if (d // 4) % ns or (d // 4) // ns < 4:
                continue
            us2 = timed(lambda: two(hot, cold, ns))
            err = (y - ref).abs().max().item() / ref.abs().max().item()
            # each pass alone
            opt("spmm_slices", ns); opt("spmm_pin", 1); opt("spmm_hot_rows", 0)
            t_hot = timed(lambda: _lib.check(lib.gss_spmm(hot.handle, d, x.data_ptr(), y.data_ptr(), None, None, st)))
            opt("spmm_slices", 0); opt("spmm_pin", 0); opt("spmm_hot_rows", 1)
            t_cold = timed(lambda: _lib.check(lib.gss_spmm_add(cold.handle, d, x.data_ptr(), y.data_ptr(), y.data_ptr(), None, None, st)))
            opt("spmm_hot_rows", -1)
            print(f"two passes, H = {h:6d} ({share:.2f} of the entries hot), hot pass {ns} pinned slices of {d * 4 // ns:3d} B: {us2:10.1f} us "
                  f"(hot {t_hot:8.1f} + cold {t_cold:8.1f})   alg {alg / us2 / 1e6:6.3f} TB/s = {alg / us2 / 1e6 / 8.0:.3f}   max rel diff {err:.1e}",
                  flush=True)
        del hot, cold
        torch.cuda.empty_cache()
