#!/usr/bin/env python3
"""Round 4 experiment: does a hipGraph shorten the step?  The step is 13 dependent launches of 5-38 us; the host is far ahead of the GPU
(so a graph cannot save launch time), but a kernel boundary costs 4.7-5 us on this machine and a graph's kernel nodes are submitted as
one batch of AQL packets.  15 steps (one per batch of an epoch) are enqueued (a) directly, as gss_plan_step does, and (b) as one captured
graph of 15 x 13 kernel nodes replayed -- alternating blocks in one process.  A TIMING experiment only: a captured step freezes the
by-value arguments (Adam's bias corrections, beta), so the replayed trajectory is not the trainer's.
usage: graph_ab.py [full | lazy] [blocks] [replays per block]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd.dist import local_comms
from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_engine, shard_rows
from gcn_drug_repurposing_amd.synth import whole_graph_standin
lib = pkg.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
d, L, B = 128, 2, 2048
adj = whole_graph_standin(seed=1)[0]
n = adj.shape[0]
X = np.random.RandomState(2).randn(n, d).astype(np.float32)
w = np.random.RandomState(7).randn(d, d) * 1e-5; np.fill_diagonal(w, 1.0)
p = {"W1": w.astype(np.float32), "b1": np.zeros(d, np.float32), "W2": w.astype(np.float32).copy(), "b2": np.zeros(d, np.float32)}
comm = local_comms(1)[0]
shard = build_shard(ScipySource(adj), comm, need_transpose=True)
eng = shard_engine(shard, shard_rows(shard, X), p, comm, num_layers=L, layer_decay=0.3, alpha=1.0, lr=3e-4, max_batch=B)
rng = np.random.RandomState(1)
batches = [torch.from_numpy(rng.permutation(n)[:B].astype(np.int32)).cuda() for _ in range(15)]
side = torch.cuda.Stream()
def epoch():
    f = eng.step if mode == "full" else eng.step_lazy
    for b in batches:
        f(b, 0.25)
with torch.cuda.stream(side):
    for _ in range(40):
        epoch()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        epoch()
    side.synchronize()
    for _ in range(20):
        g.replay()
    side.synchronize()
    t = [[], []]
    for blk in range(blocks):
        for k in ((0, 1) if blk % 2 == 0 else (1, 0)):
            side.synchronize(); t0 = time.perf_counter()
            for _ in range(reps):
                epoch() if k == 0 else g.replay()
            side.synchronize()
            t[k].append((time.perf_counter() - t0) / (reps * 15) * 1e3)
for k, name in enumerate(("direct launches", "captured graph of 15 steps")):
    a = np.array(t[k])
    print(f"{name} ({mode} step): mean {a.mean():.4f} ms/step, median {np.median(a):.4f}, min {a.min():.4f}, max {a.max():.4f} over {blocks} blocks of {reps * 15} steps; loss {eng.loss.item():.8f}")
print(f"difference of the means: {(np.mean(t[0]) - np.mean(t[1])) * 1e3:+.2f} us per step (direct minus graph)")
