#!/usr/bin/env python3
"""RMAT plan, phase by phase with a device synchronisation and a time stamp after each (where does a 10M-node step spend / lose
its time): python tools/rmat_phase_probe.py [nodes edges] -- appends to gpurun_out/probe.log"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd.dist import local_comms
from gcn_drug_repurposing_amd.shards import RmatSource, build_shard, gaussian_rows, shard_engine
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10_000_000, 200_000_000)
os.makedirs("gpurun_out", exist_ok=True)
log = open("gpurun_out/probe.log", "a")
t0 = time.perf_counter()
def say(s):
    torch.cuda.synchronize()
    log.write(f"{time.perf_counter() - t0:8.2f} s  {s}\n"); log.flush(); print(s, flush=True)
pkg.load()
comm = local_comms(1)[0]
shard = build_shard(RmatSource(n, m, seed=4, device="cuda:0"), comm, need_transpose=True, device="cuda:0")
say("shard built")
d, L, B = 128, 2, 2048
rng = np.random.RandomState(7)
w = rng.randn(d, d) * 1e-5; np.fill_diagonal(w, 1.0)
p = {"W1": w.astype(np.float32), "b1": np.zeros(d, np.float32), "W2": w.astype(np.float32).copy(), "b2": np.zeros(d, np.float32)}
eng = shard_engine(shard, gaussian_rows(0, n, d, 5), p, comm, num_layers=L, layer_decay=0.3, alpha=1.0, lr=3e-4, max_batch=B)
say("engine created")
idx = torch.from_numpy(np.random.RandomState(1).permutation(n)[:B].astype(np.int32)).cuda()
eng.forward(); say("forward done")
eng.loss_backward(idx, 0.25); say(f"loss_backward done, loss {eng.loss.item():.6f}")
eng.adam(); say("adam done")
for k in range(3):
    eng.step(idx, 0.25); say(f"step {k} done, loss {eng.loss.item():.6f}")
eng.profile(True)
for k in range(3):
    eng.step(idx, 0.25)
pr = eng.profile_read()
say("profile: " + ", ".join(f"{k} {v[0] / max(v[1], 1) * 1e3:.0f} us x{v[1]}" for k, v in pr.items() if v[1]))
for k in range(2):
    eng.step_lazy(idx, 0.25)
eng.profile(True)
for k in range(3):
    eng.step_lazy(idx, 0.25)
pr = eng.profile_read()
say("lazy profile: " + ", ".join(f"{k} {v[0] / max(v[1], 1) * 1e3:.0f} us x{v[1]}" for k, v in pr.items() if v[1]))

