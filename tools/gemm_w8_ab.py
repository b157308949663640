#!/usr/bin/env python3
"""Round 4: 128-node projection tiles as eight waves of 16 nodes (gemm_variant 5: 16 waves per CU at <= 128 registers) against four waves
of 32 nodes (variant 3, what large graphs use) and 64-node tiles (variant 4).  usage: gemm_w8_ab.py [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
d = 128
for n in [int(v) for v in (sys.argv[1:] or ["29960", "1000000", "4000000"])]:
    ax, am, pp = (torch.randn(n, d, device="cuda") for _ in range(3))
    w1, w2 = (torch.randn(d, d, device="cuda") * 0.05 for _ in range(2))
    b1, b2 = (torch.randn(d, device="cuda") for _ in range(2))
    p = torch.empty(n, d, device="cuda"); xn = torch.empty(n, d, device="cuda")
    ref = None
    res = {}
    cfgs = (("64-node tiles, 4 waves", 4), ("128-node tiles, 4 waves x 32 nodes", 3), ("128-node tiles, 8 waves x 16 nodes", 5))
    for rnd in range(4):
        for name, variant in cfgs:
            assert lib.gss_debug_set_option(b"gemm_variant", variant) == 0
            st = _lib.current_stream()
            def call():
                _lib.check(lib.gss_dense_fwd(n, d, ax.data_ptr(), am.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                             pp.data_ptr(), 0.3, p.data_ptr(), xn.data_ptr(), st))
            for _ in range(3): call()
            reps = 20 if n < 200000 else 5
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): call()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / reps * 1e3)
            if ref is None:
                ref = (p.clone(), xn.clone())
            else:
                assert torch.equal(p, ref[0]) and torch.equal(xn, ref[1]), "the variants disagree"
    fl = 2.0 * n * 2 * d * d
    for name, ts in res.items():
        us = min(ts)
        print(f"n={n} d={d} {name:36s}: {us:9.1f} us (min of {len(ts)}; all {[round(t, 1) for t in ts]})  {fl / us / 1e6:6.1f} TFLOP/s = {fl / us / 1e6 / 157.3:.3f} of the fp32 MFMA peak", flush=True)
    del ax, am, pp, p, xn, ref
    torch.cuda.empty_cache()
