#!/usr/bin/env python3
"""run N launches of gss_dense_fwd for rocprofv3 (usage: gemm_prof.py [n] [d] [reps])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 29960
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ax, am, pp = (torch.randn(n, d, device="cuda") for _ in range(3))
w1, w2 = (torch.randn(d, d, device="cuda") * 0.05 for _ in range(2))
b1, b2 = (torch.randn(d, device="cuda") for _ in range(2))
p = torch.empty(n, d, device="cuda"); xn = torch.empty(n, d, device="cuda")
for _ in range(reps):
    lib.gss_dense_fwd(n, d, ax.data_ptr(), am.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                      pp.data_ptr(), 0.3, p.data_ptr(), xn.data_ptr(), _lib.current_stream())
torch.cuda.synchronize()
