#!/usr/bin/env python3
"""run N launches of the weight gradient and of the loss sweep for rocprofv3 (usage: wgrad_loss_prof.py [n] [d] [reps])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 29960
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
B = 2048
ax, am, dp = (torch.randn(n, d, device="cuda") for _ in range(3))
gw1, gw2 = torch.empty(d, d, device="cuda"), torch.empty(d, d, device="cuda"); gb = torch.empty(d, device="cuda")
ws = torch.empty(lib.gss_wgrad_workspace_bytes(n, d), dtype=torch.uint8, device="cuda")
e = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=1)
idx = torch.randperm(n, device="cuda")[:B].to(torch.int32)
loss = torch.zeros(1, device="cuda"); de = torch.empty(B, d, device="cuda")
lws = torch.empty(lib.gss_loss_workspace_bytes(B, d), dtype=torch.uint8, device="cuda")
st = _lib.current_stream()
for _ in range(reps):
    _lib.check(lib.gss_dense_bwd_weight(n, d, dp.data_ptr(), ax.data_ptr(), am.data_ptr(), None, gw1.data_ptr(), gw2.data_ptr(), gb.data_ptr(), 0, ws.data_ptr(), st))
    _lib.check(lib.gss_loss_fwd_bwd(n, d, e.data_ptr(), idx.data_ptr(), B, 0.25, 1.0, loss.data_ptr(), de.data_ptr(), lws.data_ptr(), st))
torch.cuda.synchronize()
