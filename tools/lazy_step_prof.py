#!/usr/bin/env python3
"""per-kernel-class times of gss_plan_step vs gss_plan_step_lazy on the config-2 stand-in (HIP events inside the plan)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd.dist import local_comms
from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_engine, shard_rows
from gcn_drug_repurposing_amd.synth import whole_graph_standin
pkg.load()
d, L, B = 128, 2, 2048
adj = whole_graph_standin()[0]
n = adj.shape[0]
X = np.random.RandomState(2).randn(n, d).astype(np.float32)
w = np.random.RandomState(7).randn(d, d) * 1e-5; np.fill_diagonal(w, 1.0)
p = {"W1": w.astype(np.float32), "b1": np.zeros(d, np.float32), "W2": w.astype(np.float32).copy(), "b2": np.zeros(d, np.float32)}
comm = local_comms(1)[0]
shard = build_shard(ScipySource(adj), comm, need_transpose=True)
eng = shard_engine(shard, shard_rows(shard, X), p, comm, num_layers=L, layer_decay=0.3, alpha=1.0, lr=3e-4, max_batch=B)
idx = torch.from_numpy(np.random.RandomState(1).permutation(n)[:B].astype(np.int32)).cuda()
for name, fn in (("full", eng.step), ("lazy", eng.step_lazy)):
    for _ in range(600):
        fn(idx, 0.25)
    eng.profile(True)
    for _ in range(50):
        fn(idx, 0.25)
    pr = eng.profile_read()
    eng.profile(False)
    print(name, " ".join(f"{k}={v[0] / max(v[1], 1) * 1e3:.1f}x{v[1] // 50}" for k, v in pr.items() if v[1]))
