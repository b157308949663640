#!/usr/bin/env python3
"""gss_loss_fwd_bwd at B = 2048, d = argv[1] (default 128): sweep of the grid-size knob loss_wgs"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
n, d, b = 29960, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 2048
e = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=1)
idx = torch.randperm(n, device="cuda")[:b].to(torch.int32)
loss = torch.zeros(1, device="cuda"); de = torch.empty(b, d, device="cuda")
ws = torch.empty(lib.gss_loss_workspace_bytes(b, d), dtype=torch.uint8, device="cuda")
xx = torch.randn(4096, 4096, device="cuda")
for _ in range(300):
    xx @ xx
for wgs in (128, 192, 256, 320, 512):
    lib.gss_debug_set_option(b"loss_wgs", wgs)
    ws = torch.empty(lib.gss_loss_workspace_bytes(b, d), dtype=torch.uint8, device="cuda")
    best = 1e9
    for _ in range(3):
        for _ in range(10):
            lib.gss_loss_fwd_bwd(n, d, e.data_ptr(), idx.data_ptr(), b, 0.25, 1.0, loss.data_ptr(), de.data_ptr(), ws.data_ptr(), _lib.current_stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            lib.gss_loss_fwd_bwd(n, d, e.data_ptr(), idx.data_ptr(), b, 0.25, 1.0, loss.data_ptr(), de.data_ptr(), ws.data_ptr(), _lib.current_stream())
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    print(f"loss_wgs={wgs}: gather + sweep + finish {best:.1f} us")
lib.gss_debug_set_option(b"loss_wgs", 256)
ws = torch.empty(lib.gss_loss_workspace_bytes(b, d), dtype=torch.uint8, device="cuda")
for _ in range(5):
    lib.gss_loss_fwd_bwd(n, d, e.data_ptr(), idx.data_ptr(), b, 0.25, 1.0, loss.data_ptr(), de.data_ptr(), ws.data_ptr(), _lib.current_stream())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lib.gss_loss_fwd_bwd(n, d, e.data_ptr(), idx.data_ptr(), b, 0.25, 1.0, loss.data_ptr(), de.data_ptr(), ws.data_ptr(), _lib.current_stream())
e1.record(); torch.cuda.synchronize()
print("loss_fwd_bwd (3 launches) us:", e0.elapsed_time(e1) / 20 * 1e3)
