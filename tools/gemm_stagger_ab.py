#!/usr/bin/env python3
"""de-phasing sweep of the projection (knob gemm_stagger: second-generation workgroups start k x 512 cycles late); interleaved with the
baseline so that the drift of the run shows.  GPU box only.  usage: gemm_stagger_ab.py [n] [d]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 29960
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ax, am, pp = (torch.randn(n, d, device="cuda") for _ in range(3))
w1, w2 = (torch.randn(d, d, device="cuda") * 0.05 for _ in range(2))
b1, b2 = (torch.randn(d, device="cuda") for _ in range(2))
p = torch.empty(n, d, device="cuda"); xn = torch.empty(n, d, device="cuda")
st = _lib.current_stream()
fl = 2.0 * n * 2 * d * d


def timeit():
    best = 1e9
    call = lambda: lib.gss_dense_fwd(n, d, ax.data_ptr(), am.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                     pp.data_ptr(), 0.3, p.data_ptr(), xn.data_ptr(), st)
    for _ in range(3):
        for _ in range(5): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
    return best


for _ in range(200):   # settle the clocks
    lib.gss_dense_fwd(n, d, ax.data_ptr(), am.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), pp.data_ptr(), 0.3, p.data_ptr(), xn.data_ptr(), st)
torch.cuda.synchronize()
for k in (0, 1, 0, 2, 0, 3, 0, 4, 0, 6, 0, 8, 0, 12, 0, 16, 0, 24):
    lib.gss_debug_set_option(b"gemm_stagger", k)
    us = timeit()
    print(f"projection n={n} d={d} gemm_stagger={k:3d} ({k * 512 / 2400:.2f} us): {us:7.2f} us  {fl/us/1e6/157.3*100:4.1f} % of fp32 MFMA peak", flush=True)
lib.gss_debug_set_option(b"gemm_stagger", 0)
