#!/usr/bin/env python3
"""run N launches of one SpMM variant for rocprofv3 (usage: spmm_prof.py <variant> <d> [reps])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib, synth
from gcn_drug_repurposing_amd.graph import GssGraph
lib = pkg.load()
variant, d = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
adj, _, _ = synth.whole_graph_standin(1)
g = GssGraph(adj)
lib.gss_debug_set_option(b"spmm_variant", variant)
x = torch.randn(g.n, d, device="cuda"); y = torch.empty(g.n, d, device="cuda")
for _ in range(reps):
    lib.gss_spmm(g.a.handle, d, x.data_ptr(), y.data_ptr(), None, None, _lib.current_stream())
torch.cuda.synchronize()
