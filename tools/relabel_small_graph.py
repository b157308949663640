#!/usr/bin/env python3
"""does hub-first relabelling help a cache-resident graph?  config 2 stand-in through shards.build_shard with and without it:
SpMM time and whole-step time"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gcn_drug_repurposing_amd as pkg  # noqa: E402
from gcn_drug_repurposing_amd import _lib, synth  # noqa: E402
from gcn_drug_repurposing_amd.dist import local_comms  # noqa: E402
from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_engine, shard_rows  # noqa: E402

lib = pkg.load()
adj, _, _ = synth.whole_graph_standin(1)
n, d, L, B = adj.shape[0], 128, 2, 2048
X = synth.gaussian_features(n, d, 2)
np.random.seed(7)
w = np.random.randn(d, d) * 1e-5
np.fill_diagonal(w, 1.0)
params = {"W1": w.astype(np.float32), "b1": np.zeros(d, np.float32), "W2": w.astype(np.float32).copy(), "b2": np.zeros(d, np.float32)}
rng = np.random.RandomState(1)
batches = [torch.from_numpy(rng.permutation(n)[:B].astype(np.int32)).cuda() for _ in range(20)]
xx = torch.randn(4096, 4096, device="cuda")
for _ in range(200):
    xx @ xx
for rl in (False, True):
    comm = local_comms(1)[0]
    shard = build_shard(ScipySource(adj), comm, relabel=rl)
    eng = shard_engine(shard, shard_rows(shard, X), params, comm, num_layers=L, lr=3e-4, max_batch=B)
    x = torch.randn(n, d, device="cuda")
    y = torch.empty(n, d, device="cuda")
    st = _lib.current_stream()
    for _ in range(20):
        lib.gss_spmm(shard.a.handle, d, x.data_ptr(), y.data_ptr(), None, None, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        lib.gss_spmm(shard.a.handle, d, x.data_ptr(), y.data_ptr(), None, None, st)
    e1.record()
    torch.cuda.synchronize()
    t_spmm = e0.elapsed_time(e1) / 200 * 1e3
    for b in batches:
        eng.step(b, 0.25)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        for b in batches:
            eng.step(b, 0.25)
    torch.cuda.synchronize()
    print(f"relabel={rl}: plain SpMM {t_spmm:.1f} us, step {(time.perf_counter() - t0) / 1000 * 1e3:.4f} ms, loss {eng.loss.item():.8f}")
