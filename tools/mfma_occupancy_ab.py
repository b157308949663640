#!/usr/bin/env python3
"""Does the hardware dispatcher double workgroups up on some CUs while others idle?  The three fp32-MFMA kernels of a step with their
natural LDS footprint against footprints that admit exactly one (>= 81 KB) or two (54-80 KB) workgroups per CU (knobs gemm_lds_kb,
wgrad_lds_kb, loss_lds_kb).  GPU box only; per-op entry points, min of 3 rounds of 20 launches.  usage: mfma_occupancy_ab.py [n] [d]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 29960
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = 2048


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


ax, am, pp, dp = (torch.randn(n, d, device="cuda") for _ in range(4))
w1, w2 = (torch.randn(d, d, device="cuda") * 0.05 for _ in range(2))
b1, b2 = (torch.randn(d, device="cuda") for _ in range(2))
p = torch.empty(n, d, device="cuda"); xn = torch.empty(n, d, device="cuda")
gw1, gw2 = torch.empty(d, d, device="cuda"), torch.empty(d, d, device="cuda"); gb = torch.empty(d, device="cuda")
ws = torch.empty(lib.gss_wgrad_workspace_bytes(n, d), dtype=torch.uint8, device="cuda")
e = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=1)
idx = torch.randperm(n, device="cuda")[:B].to(torch.int32)
loss = torch.zeros(1, device="cuda"); de = torch.empty(B, d, device="cuda")
lws = torch.empty(lib.gss_loss_workspace_bytes(B, d), dtype=torch.uint8, device="cuda")
st = _lib.current_stream()
fl = 2.0 * n * 2 * d * d
for kb in (0, 56, 80, 0, 56, 80):
    assert lib.gss_debug_set_option(b"gemm_lds_kb", kb) == 0
    us = timeit(lambda: lib.gss_dense_fwd(n, d, ax.data_ptr(), am.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                          pp.data_ptr(), 0.3, p.data_ptr(), xn.data_ptr(), st))
    print(f"projection n={n} d={d} gemm_lds_kb={kb:3d}: {us:7.1f} us  {fl/us/1e6:6.1f} TFLOP/s  {fl/us/1e6/157.3*100:4.1f} %", flush=True)
lib.gss_debug_set_option(b"gemm_lds_kb", 0)
for kb in (0, 81, 120, 0, 81):
    assert lib.gss_debug_set_option(b"wgrad_lds_kb", kb) == 0
    us = timeit(lambda: lib.gss_dense_bwd_weight(n, d, dp.data_ptr(), ax.data_ptr(), am.data_ptr(), None, gw1.data_ptr(), gw2.data_ptr(),
                                                 gb.data_ptr(), 0, ws.data_ptr(), st))
    print(f"wgrad+reduce n={n} d={d} wgrad_lds_kb={kb:3d}: {us:7.1f} us  {fl/us/1e6:6.1f} TFLOP/s  {fl/us/1e6/157.3*100:4.1f} %", flush=True)
lib.gss_debug_set_option(b"wgrad_lds_kb", 0)
fl2 = 4.0 * B * B * d
for kb in (0, 81, 40, 0, 81):
    assert lib.gss_debug_set_option(b"loss_lds_kb", kb) == 0
    us = timeit(lambda: lib.gss_loss_fwd_bwd(n, d, e.data_ptr(), idx.data_ptr(), B, 0.25, 1.0, loss.data_ptr(), de.data_ptr(), lws.data_ptr(), st))
    print(f"loss gather+sweep+finish B={B} d={d} loss_lds_kb={kb:3d}: {us:7.1f} us  {fl2/us/1e6:6.1f} TFLOP/s", flush=True)
lib.gss_debug_set_option(b"loss_lds_kb", 0)
