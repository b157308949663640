#!/usr/bin/env python3
"""diffusion profiles of every drug and indication of the whole-graph stand-in (N = 29,960, 2,502 start nodes):
time of the device batch and of its fp64 SpMM (GPU box only; the CPU comparison is bench.py --workload diffusion).
usage: diffusion_bench.py [scale]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from gcn_drug_repurposing_amd import synth
from gcn_drug_repurposing_amd.diffusion import PprEngine, PprProblem

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1
adj, ntype, _ = synth.whole_graph_standin(seed=1, scale=scale)
m0 = sp.csr_matrix(adj, dtype=np.float64)
di = np.flatnonzero(ntype <= 1)
prot = {int(s): m0.indices[m0.indptr[s]:m0.indptr[s + 1]].tolist() for s in di}
hp = (0.8595436247434408, 1000, 1e-6)
t0 = time.time(); prob = PprProblem(m0, di, prot); t_prep = time.time() - t0
eng = PprEngine(prob)
print(f"N={prob.n} nnz={prob.mt.nnz} start nodes={prob.k} (kpad {prob.kpad}) overrides={len(prob.ovr_col)} empty rows={len(prob.z_rows)} "
      f"host prep {t_prep:.2f} s, device slab {eng.device_bytes() / 1e6:.0f} MB + x {prob.n * prob.kpad * 8 / 1e6:.0f} MB")
x, it = eng.run(hp[0], hp[2], hp[1])
torch.cuda.synchronize(); t0 = time.time()
x, it = eng.run(hp[0], hp[2], hp[1])
torch.cuda.synchronize(); t_run = time.time() - t0
n_it = int(it.max())
print(f"device: {t_run * 1e3:.1f} ms for all {prob.k} profiles, {n_it} iterations (min {it.min()}), {t_run / n_it * 1e3:.3f} ms/iteration")
# the product alone
y = torch.empty_like(x)
for _ in range(3): eng.spmm(x, y)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): eng.spmm(x, y)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
alg = prob.mt.nnz * 12 + 4 * (prob.n + 1) + 2 * prob.n * prob.kpad * 8
print(f"fp64 SpMM: {us:.0f} us/launch, algorithmic {alg / 1e6:.0f} MB -> {alg / us / 1e3:.0f} GB/s = {alg / us / 1e3 / 8000:.3f} of 8 TB/s; "
      f"gathered {prob.mt.nnz * prob.kpad * 8 / us / 1e6:.1f} TB/s")
xs = x[:, :prob.k].t().contiguous().cpu().numpy()
print(f"column sums in [{xs.sum(1).min():.12f}, {xs.sum(1).max():.12f}]")
