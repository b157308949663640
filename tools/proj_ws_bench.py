#!/usr/bin/env python3
"""Round 5: the d = 128 forward projection as the weight-stationary persistent kernel (gemm_ws = 1) against the staged tiles (gemm_ws = 0),
per-op entry point, interleaved rounds, bits compared; then a sweep of the persistent grid and the staggered start.
GPU box only.  usage: proj_ws_bench.py [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
d = 128

def setk(**kw):
    for k, v in kw.items():
        assert lib.gss_debug_set_option(k.encode(), v) == 0, (k, v)

for n in [int(v) for v in (sys.argv[1:] or ["29960", "250000", "1000000"])]:
    ax, am, pp = (torch.randn(n, d, device="cuda") for _ in range(3))
    w1, w2 = (torch.randn(d, d, device="cuda") * 0.05 for _ in range(2))
    b1, b2 = (torch.randn(d, device="cuda") for _ in range(2))
    p = torch.empty(n, d, device="cuda"); xn = torch.empty(n, d, device="cuda")
    st = _lib.current_stream()
    def call(prev=True):
        _lib.check(lib.gss_dense_fwd(n, d, ax.data_ptr(), am.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                     pp.data_ptr() if prev else None, 0.3, p.data_ptr(), xn.data_ptr(), st))
    def timed(reps, prev=True):
        for _ in range(3): call(prev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): call(prev)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    reps = 50 if n < 200000 else 8
    # (round 6: the kernel's workgroup count, stagger and weight-fetch mode are constants now -- 512 / 4 / per-row loads after an L2 warm-up,
    #  the optima of the round-5 sweep in profiles/r05_proj_ws_bench.txt)
    cfgs = [("staged tiles (gemm_ws=0)", dict(gemm_ws=0)), ("weight-stationary (gemm_ws=1)", dict(gemm_ws=1))]
    res, ref = {}, None
    for rnd in range(4):
        for name, kw in cfgs:
            setk(**kw)
            res.setdefault(name, []).append(timed(reps))
            if ref is None:
                ref = (p.clone(), xn.clone())
            else:
                assert torch.equal(p, ref[0]) and torch.equal(xn, ref[1]), f"{name}: bits differ"
    setk(gemm_ws=-1)
    fl = 2.0 * n * 2 * d * d
    for name, ts in res.items():
        us = min(ts)
        print(f"n={n} d={d} {name:40s}: {us:9.2f} us (min of {len(ts)}; all {[round(t, 2) for t in ts]})  {fl / us / 1e6:6.1f} TFLOP/s = {fl / us / 1e6 / 157.3:.3f} of the fp32 MFMA peak", flush=True)
    del ax, am, pp, p, xn, ref
    torch.cuda.empty_cache()
