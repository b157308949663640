#!/usr/bin/env python3
"""A/B of the static wave-priority knobs of the fp32-MFMA kernels (GPU box only): gemm_prio (the first generation of projection
workgroups at raised priority) and wgrad_prio (one half of a weight-gradient workgroup's waves), per-op entry points, min of 3 rounds
of 20 launches.  usage: mfma_prio_ab.py [n] [d ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 29960


def timeit(call):
    best = 1e9
    for _ in range(3):
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    return best


for d in [int(v) for v in (sys.argv[2:] or ["128", "256"])]:
    ax, am, pp, dp = (torch.randn(n, d, device="cuda") for _ in range(4))
    w1, w2 = (torch.randn(d, d, device="cuda") * 0.05 for _ in range(2))
    b1, b2 = (torch.randn(d, device="cuda") for _ in range(2))
    p = torch.empty(n, d, device="cuda"); xn = torch.empty(n, d, device="cuda")
    gw1, gw2 = torch.empty(d, d, device="cuda"), torch.empty(d, d, device="cuda"); gb = torch.empty(d, device="cuda")
    ws = torch.empty(lib.gss_wgrad_workspace_bytes(n, d), dtype=torch.uint8, device="cuda")
    st = _lib.current_stream()
    fl = 2.0 * n * 2 * d * d
    ref = None
    for prio in (0, -1, 256, 100000, 0, -1):
        lib.gss_debug_set_option(b"gemm_prio", prio)
        us = timeit(lambda: lib.gss_dense_fwd(n, d, ax.data_ptr(), am.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                              pp.data_ptr(), 0.3, p.data_ptr(), xn.data_ptr(), st))
        torch.cuda.synchronize()
        if ref is None: ref = p.clone()
        same = bool(torch.equal(ref, p))
        print(f"projection n={n} d={d} gemm_prio={prio:6d}: {us:7.1f} us  {fl/us/1e6:6.1f} TFLOP/s  {fl/us/1e6/157.3*100:4.1f} %  bits_same={same}", flush=True)
    lib.gss_debug_set_option(b"gemm_prio", 0)
    ref = None
    for prio in (0, 1, 2):
        lib.gss_debug_set_option(b"wgrad_prio", prio)
        us = timeit(lambda: lib.gss_dense_bwd_weight(n, d, dp.data_ptr(), ax.data_ptr(), am.data_ptr(), None, gw1.data_ptr(), gw2.data_ptr(),
                                                     gb.data_ptr(), 0, ws.data_ptr(), st))
        torch.cuda.synchronize()
        if ref is None: ref = gw1.clone()
        print(f"wgrad+reduce n={n} d={d} wgrad_prio={prio}: {us:7.1f} us  {fl/us/1e6:6.1f} TFLOP/s  {fl/us/1e6/157.3*100:4.1f} %  bits_same={bool(torch.equal(ref, gw1))}", flush=True)
    lib.gss_debug_set_option(b"wgrad_prio", 0)
