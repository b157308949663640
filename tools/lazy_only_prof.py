#!/usr/bin/env python3
"""config 2, LAZY steps only (gss_plan_step_lazy with layer 1's SpMM results kept: the step train.py runs) -- for a rocprofv3
--kernel-trace --stats run whose per-kernel averages are not mixed with the full step's (tools/profile_r04.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd.dist import local_comms
from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_engine, shard_rows
from gcn_drug_repurposing_amd.synth import whole_graph_standin
pkg.load()
d, L, B = 128, 2, 2048
keep = len(sys.argv) < 2 or sys.argv[1] != "nokeep"
adj = whole_graph_standin()[0]
n = adj.shape[0]
X = np.random.RandomState(2).randn(n, d).astype(np.float32)
w = np.random.RandomState(7).randn(d, d) * 1e-5; np.fill_diagonal(w, 1.0)
p = {"W1": w.astype(np.float32), "b1": np.zeros(d, np.float32), "W2": w.astype(np.float32).copy(), "b2": np.zeros(d, np.float32)}
comm = local_comms(1)[0]
shard = build_shard(ScipySource(adj), comm, need_transpose=True)
eng = shard_engine(shard, shard_rows(shard, X), p, comm, num_layers=L, layer_decay=0.3, alpha=1.0, lr=3e-4, max_batch=B, cache_layer1=keep)
rng = np.random.RandomState(1)
batches = [torch.from_numpy(rng.permutation(n)[:B].astype(np.int32)).cuda() for _ in range(15)]
for k in range(1500):
    eng.step_lazy(batches[k % 15], 0.25)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for k in range(1500):
    eng.step_lazy(batches[k % 15], 0.25)
torch.cuda.synchronize()
print(f"lazy steps (layer 1 kept: {keep}): {(time.perf_counter() - t0) / 1500 * 1e3:.4f} ms/step, loss {eng.loss.item():.6f}")
