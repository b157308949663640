#!/usr/bin/env python3
"""time the kNN graph build at whole_graph size: device (gss_knn_topk) vs host (chunked numpy)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gcn_drug_repurposing_amd import graph, synth
n, d, k = int(sys.argv[1]) if len(sys.argv) > 1 else 29960, 128, 5
X = synth.gaussian_features(n, d, 2).astype(np.float64)
graph.knn_descriptor_adj_device(X[:512], k)
torch.cuda.synchronize(); t0 = time.perf_counter()
a = graph.knn_descriptor_adj_device(X, k)
t1 = time.perf_counter()
print(f"device kNN N={n} d={d} k={k}: {t1 - t0:.3f} s total (incl. H2D/D2H + host CSR assembly), nnz={a.nnz}")
import ctypes as C
from gcn_drug_repurposing_amd import _lib
lib = _lib.load(); xd = torch.from_numpy(X).cuda(); tv = torch.empty(n, k, dtype=torch.float64, device="cuda"); ti = torch.empty(n, k, dtype=torch.int32, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); lib.gss_knn_topk(n, d, xd.data_ptr(), k, tv.data_ptr(), ti.data_ptr(), _lib.current_stream()); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1); print(f"  kernel: {ms:.2f} ms = {2.0 * n * n * d / ms / 1e9:.1f} TFLOP/s fp64")
if "--host" in sys.argv:
    t0 = time.perf_counter(); b = graph.knn_descriptor_adj(X, k); t1 = time.perf_counter()
    print(f"host   kNN: {t1 - t0:.2f} s, identical structure: {np.array_equal(a.indices, b.indices) and np.array_equal(a.indptr, b.indptr)}")
