#!/usr/bin/env python3
"""where does a sharded step spend host time?  world = 1 over RCCL on one GPU (GPU box only)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd import synth
from gcn_drug_repurposing_amd.dist import ShardedEngine
pkg.load()
adj, _, _ = synth.whole_graph_standin(1)
n, d, B = adj.shape[0], 128, 2048
x = synth.gaussian_features(n, d, 2)
np.random.seed(7)
w = np.random.randn(d, d) * 1e-5; np.fill_diagonal(w, 1.0)
params = {"W1": w.astype(np.float32), "b1": np.zeros(d, np.float32), "W2": w.astype(np.float32).copy(), "b2": np.zeros(d, np.float32)}
eng = ShardedEngine(adj, x, params, num_layers=2, layer_decay=0.3, alpha=1.0, lr=3e-4, max_batch=B)
rng = np.random.RandomState(0)
idx = torch.from_numpy(rng.permutation(n).astype(np.int32)).cuda()
def run(k):
    for s in range(k):
        eng.step(idx, 0.25, count=B, offset=(s % 14) * B)
run(10); torch.cuda.synchronize()
t0 = time.perf_counter(); run(100); t_host = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"100 steps: host issue {t_host * 10:.3f} ms/step, with GPU drain {t_all * 10:.3f} ms/step")
pr = cProfile.Profile(); pr.enable(); run(100); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
dist.destroy_process_group()
