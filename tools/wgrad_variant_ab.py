#!/usr/bin/env python3
"""weight gradient: operands by direct loads (wgrad_variant 1) against the LDS-DMA ring (2), interleaved, bits compared.
GPU box only.  usage: wgrad_variant_ab.py [n] [d ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 29960
for d in [int(v) for v in (sys.argv[2:] or ["128", "256", "64"])]:
    ax, am, dp = (torch.randn(n, d, device="cuda") for _ in range(3))
    gw1, gw2 = torch.empty(d, d, device="cuda"), torch.empty(d, d, device="cuda"); gb = torch.empty(d, device="cuda")
    ws = torch.empty(lib.gss_wgrad_workspace_bytes(n, d), dtype=torch.uint8, device="cuda")
    st = _lib.current_stream()
    fl = 2.0 * n * 2 * d * d
    call = lambda: _lib.check(lib.gss_dense_bwd_weight(n, d, dp.data_ptr(), ax.data_ptr(), am.data_ptr(), None, gw1.data_ptr(), gw2.data_ptr(),
                                                        gb.data_ptr(), 0, ws.data_ptr(), st))
    for _ in range(100): call()
    ref = None
    for v in (1, 2, 1, 2, 1, 2):
        assert lib.gss_debug_set_option(b"wgrad_variant", v) == 0
        best = 1e9
        for _ in range(3):
            for _ in range(3): call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): call()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
        if ref is None: ref = (gw1.clone(), gw2.clone(), gb.clone())
        same = torch.equal(ref[0], gw1) and torch.equal(ref[1], gw2) and torch.equal(ref[2], gb)
        print(f"wgrad+reduce n={n} d={d} wgrad_variant={v}: {best:7.2f} us  {fl/best/1e6:6.1f} TFLOP/s  {fl/best/1e6/157.3*100:4.1f} %  bits_same={same}", flush=True)
    lib.gss_debug_set_option(b"wgrad_variant", 1)
