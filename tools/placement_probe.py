#!/usr/bin/env python3
"""Round 4: how much does a plan's PLACEMENT cost?  K plans with identical settings in one process, timed in rotating blocks; printed with
the addresses of a few of their buffers.  (Two plans differ by up to 3.4 us per step, constant inside a process: tools/ab_live.py.)
usage: placement_probe.py [plans] [rounds] [steps per block]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd.dist import local_comms
from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_engine, shard_rows
from gcn_drug_repurposing_amd.synth import whole_graph_standin
lib = pkg.load()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
d, L, B = 128, 2, 2048
adj = whole_graph_standin(seed=1)[0]
n = adj.shape[0]
X = np.random.RandomState(2).randn(n, d).astype(np.float32)
w = np.random.RandomState(7).randn(d, d) * 1e-5; np.fill_diagonal(w, 1.0)
p = {"W1": w.astype(np.float32), "b1": np.zeros(d, np.float32), "W2": w.astype(np.float32).copy(), "b2": np.zeros(d, np.float32)}
comm = local_comms(1)[0]
shard = build_shard(ScipySource(adj), comm, need_transpose=True)
engs = [shard_engine(shard, shard_rows(shard, X), p, comm, num_layers=L, layer_decay=0.3, alpha=1.0, lr=3e-4, max_batch=B) for _ in range(K)]
rng = np.random.RandomState(1)
batches = [torch.from_numpy(rng.permutation(n)[:B].astype(np.int32)).cuda() for _ in range(15)]
def run(e, k):
    for i in range(k):
        e.step(batches[i % 15], 0.25)
for e in engs:
    run(e, 600)
torch.cuda.synchronize()
t = [[] for _ in engs]
for r in range(rounds):
    order = list(range(K)) if r % 2 == 0 else list(range(K - 1, -1, -1))
    for k in order:
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(engs[k], steps); torch.cuda.synchronize()
        t[k].append((time.perf_counter() - t0) / steps * 1e3)
base = np.mean([np.mean(v) for v in t])
for k, e in enumerate(engs):
    a = np.array(t[k])
    print(f"plan {k}: mean {a.mean():.4f} ms ({(a.mean() - base) * 1e3:+.2f} us vs the mean of all), spread {a.min():.4f}..{a.max():.4f}; emb @ {e.emb.data_ptr():#x} x @ {e.x.data_ptr():#x}")
