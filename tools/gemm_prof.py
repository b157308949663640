#!/usr/bin/env python3
"""run N launches of gss_dense_fwd for rocprofv3: gemm_prof.py <variant> <noepi> [d]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib
lib = pkg.load()
variant, noepi = int(sys.argv[1]), int(sys.argv[2])
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
n = 29960
lib.gss_debug_set_option(b"gemm_variant", variant); lib.gss_debug_set_option(b"gemm_noepi", noepi)
ax, am, pp = (torch.randn(n, d, device="cuda") for _ in range(3))
w1, w2 = (torch.randn(d, d, device="cuda") * 0.05 for _ in range(2))
b1, b2 = (torch.randn(d, device="cuda") for _ in range(2))
p = torch.empty(n, d, device="cuda"); xn = torch.empty(n, d, device="cuda")
for _ in range(5):
    lib.gss_dense_fwd(n, d, ax.data_ptr(), am.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), pp.data_ptr(), 0.3,
                      p.data_ptr(), xn.data_ptr(), _lib.current_stream())
torch.cuda.synchronize()
