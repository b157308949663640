#!/usr/bin/env python3
"""micro-benchmark of gss_spmm on the bench graphs: time per launch vs feature width (GPU box only)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import _lib, synth
from gcn_drug_repurposing_amd.graph import GssGraph

lib = pkg.load()
which = sys.argv[1] if len(sys.argv) > 1 else "whole"
if which == "whole":
    adj, _, _ = synth.whole_graph_standin(1)
elif which == "knn":
    from gcn_drug_repurposing_amd.graph import knn_descriptor_adj_device
    adj = knn_descriptor_adj_device(synth.gaussian_features(29960, 128, 2).astype(np.float64), 5)
else:
    n, m = int(sys.argv[2]), int(sys.argv[3])
    adj = synth.rmat_adj(n, m)
g = GssGraph(adj)
n, nnz = g.n, g.nnz
print(f"graph {which}: N={n} nnz={nnz} long_rows={(np.diff(g.a.h_indptr) > 512).sum()}")
for d in [int(v) for v in (sys.argv[4:] or ["16", "32", "64", "128", "256"])]:
  for variant in (1, 2):
    lib.gss_debug_set_option(b"spmm_variant", variant)   # 1 = row per wave, 2 = nnz-balanced segments (automatic slicing)
    x = torch.randn(n, d, device="cuda")
    y = torch.empty(n, d, device="cuda")
    st = _lib.current_stream()
    for _ in range(5):
        lib.gss_spmm(g.a.handle, d, x.data_ptr(), y.data_ptr(), None, None, st)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    ev0.record()
    for _ in range(reps):
        lib.gss_spmm(g.a.handle, d, x.data_ptr(), y.data_ptr(), None, None, st)
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) / reps * 1e3
    alg = 8 * nnz + 4 * (n + 1) + 8 * n * d
    print(f"v{variant} d={d:4d}  {us:8.1f} us/launch  edges/s={nnz / us * 1e6:.3e}  alg GB/s={alg / us / 1e3:8.1f}  gather TB/s={nnz * d * 4 / us / 1e6:6.2f}")
