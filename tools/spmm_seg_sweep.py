#!/usr/bin/env python3
"""sweep the SpMM segment length (debug knob spmm_seg_edges) on the whole_graph stand-in"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib, synth
from gcn_drug_repurposing_amd.graph import GssGraph
lib = pkg.load()
adj, _, _ = synth.whole_graph_standin(1)
for d in (128, 256):
    for seg in (8, 16, 24, 32, 48, 64, 128):
        lib.gss_debug_set_option(b"spmm_seg_edges", seg)
        g = GssGraph(adj, need_transpose=False)
        x = torch.randn(g.n, d, device="cuda"); y = torch.empty(g.n, d, device="cuda")
        st = _lib.current_stream()
        best = 1e9
        for rnd in range(3):
            for _ in range(5): lib.gss_spmm(g.a.handle, d, x.data_ptr(), y.data_ptr(), None, None, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): lib.gss_spmm(g.a.handle, d, x.data_ptr(), y.data_ptr(), None, None, st)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
        print(f"d={d} seg_edges={seg:4d}: {best:7.1f} us")
