#!/usr/bin/env python3
"""wall time of the whole train.py run at whole_graph size (the reference's train.sh flags), by phase (GPU box only)"""
import os, sys, time, tempfile, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gcn_drug_repurposing_amd import embio, synth
tmp = tempfile.mkdtemp()
n, d = 29960, 128
x = synth.gaussian_features(n, d, 2).astype(np.float64)
t0 = time.time(); embio.write_embs(os.path.join(tmp, "in.embs.txt"), [str(i) for i in range(n)], x); print(f"write synthetic input {time.time() - t0:.1f} s")
t0 = time.time(); names, xr = embio.read_embs(os.path.join(tmp, "in.embs.txt")); print(f"read_embs {time.time() - t0:.2f} s")
e = np.random.RandomState(0).randn(n, d)
t0 = time.time(); embio.write_graph_embs(os.path.join(tmp, "o.txt"), e); print(f"write_graph_embs (np.savetxt) {time.time() - t0:.2f} s")
t0 = time.time(); np.loadtxt(os.path.join(tmp, "o.txt")); print(f"np.loadtxt of the output {time.time() - t0:.2f} s")
t0 = time.time()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, os.path.join(root, "train.py"), "--emb-file", os.path.join(tmp, "in.embs.txt"), "--num-layers", "2", "--hidden-units", "128",
                    "--k", "5", "--kq", "5", "--epoch", "20", "--lr", "0.0003", "--graph-mode", "descriptor", "--beta-percentile", "98",
                    "--batch-size", "2048", "--out", os.path.join(tmp, "graph_embs.txt")], capture_output=True, text=True, cwd=tmp)
print(f"train.py end to end {time.time() - t0:.1f} s (rc {r.returncode})")
print("\n".join(l for l in r.stdout.splitlines() if "Created G" in l or "selected beta" in l or "time" in l.lower())[-600:])
print(r.stderr[-400:] if r.returncode else "")
