#!/usr/bin/env python3
"""micro-benchmark of gss_dense_fwd variants (GPU box only): usage gemm_bench.py [n] [d ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 29960
for d in [int(v) for v in (sys.argv[2:] or ["128", "256", "64"])]:
    ax, am, pp = (torch.randn(n, d, device="cuda") for _ in range(3))
    w1, w2 = (torch.randn(d, d, device="cuda") * 0.05 for _ in range(2))
    b1, b2 = (torch.randn(d, device="cuda") for _ in range(2))
    p = torch.empty(n, d, device="cuda"); xn = torch.empty(n, d, device="cuda")
    res = {}
    for rnd in range(3):
        for variant in (1, 2, 3, 4):
            lib.gss_debug_set_option(b"gemm_variant", variant)
            st = _lib.current_stream()
            def call():
                lib.gss_dense_fwd(n, d, ax.data_ptr(), am.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                  pp.data_ptr(), 0.3, p.data_ptr(), xn.data_ptr(), st)
            for _ in range(3): call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): call()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(variant, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * n * 2 * d * d
    for v, ts in res.items():
        us = min(ts)
        print(f"gemm v{v} n={n} d={d}: {us:7.1f} us (min of {len(ts)} rounds, all {[round(t,1) for t in ts]})  {fl/us/1e6:6.1f} TFLOP/s  {fl/us/1e6/157.3*100:4.1f}% of fp32 MFMA peak")
