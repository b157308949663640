#!/usr/bin/env python3
"""A/B of one KERNEL-SELECTION knob on ONE live plan (gss_plan_debug_set_option): the same buffers at the same addresses run alternating
blocks of steps under value a and value b.  tools/ab_inproc.py -- two plans in one process -- carries a placement bias: two plans with
IDENTICAL settings differ by up to +-3.4 us per step from one process to the next (constant inside a process, hence invisible in the
block-to-block spread), because their buffers sit at different addresses.  This tool has none (check: ab_live.py <knob> <v> <v>).
usage: ab_live.py <knob> <value a> <value b> [full | lazy | lazy_kept] [blocks] [steps per block] [config: 2 (default) | 3 (whole_graph_pathway, d = 256, L = 3)]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gcn_drug_repurposing_amd as pkg
from gcn_drug_repurposing_amd.dist import local_comms
from gcn_drug_repurposing_amd.shards import ScipySource, build_shard, shard_engine, shard_rows
from gcn_drug_repurposing_amd.synth import whole_graph_standin
lib = pkg.load()
knob, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "full"
blocks = int(sys.argv[5]) if len(sys.argv) > 5 else 16
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 300
config = int(sys.argv[7]) if len(sys.argv) > 7 else 2
d, L, B = (128, 2, 2048) if config == 2 else (256, 3, 2048)
adj = whole_graph_standin(seed=1, pathway_edges=config == 3)[0]
n = adj.shape[0]
X = np.random.RandomState(2 if config == 2 else 3).randn(n, d).astype(np.float32)
w = np.random.RandomState(7).randn(d, d) * 1e-5; np.fill_diagonal(w, 1.0)
p = {"W1": w.astype(np.float32), "b1": np.zeros(d, np.float32), "W2": w.astype(np.float32).copy(), "b2": np.zeros(d, np.float32)}
comm = local_comms(1)[0]
shard = build_shard(ScipySource(adj), comm, need_transpose=True)
eng = shard_engine(shard, shard_rows(shard, X), p, comm, num_layers=L, layer_decay=0.3, alpha=1.0, lr=3e-4, max_batch=B, cache_layer1=(mode == "lazy_kept"))
rng = np.random.RandomState(1)
batches = [torch.from_numpy(rng.permutation(n)[:B].astype(np.int32)).cuda() for _ in range(15)]
def run(k):
    f = eng.step if mode == "full" else eng.step_lazy
    for i in range(k):
        f(batches[i % 15], 0.25)
def use(v):
    assert lib.gss_plan_debug_set_option(eng.handle, knob.encode(), v) == 0, lib.gss_last_error().decode()
for kv in os.environ.get("AB_PRESET", "").split():      # e.g. AB_PRESET="spmm_pin=1": other live knobs held at a value for the whole run
    k_, v_ = kv.split("=")
    assert lib.gss_plan_debug_set_option(eng.handle, k_.encode(), int(v_)) == 0, lib.gss_last_error().decode()
for v in (va, vb):
    use(v); run(600 if config == 2 else 150)
torch.cuda.synchronize()
t = [[], []]
for blk in range(blocks):
    for k in ((0, 1) if blk % 2 == 0 else (1, 0)):
        use((va, vb)[k])
        run(20)                                         # the switch itself: first launches of the other kernels
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
        t[k].append((time.perf_counter() - t0) / steps * 1e3)
for k, v in enumerate((va, vb)):
    a = np.array(t[k])
    print(f"{knob}={v} ({mode} step, one live plan): mean {a.mean():.4f} ms, median {np.median(a):.4f}, min {a.min():.4f}, max {a.max():.4f} over {blocks} blocks of {steps} steps")
print(f"difference of the means: {(np.mean(t[0]) - np.mean(t[1])) * 1e3:+.2f} us per step ({knob}={va} minus {knob}={vb}); loss {eng.loss.item():.8f}")
