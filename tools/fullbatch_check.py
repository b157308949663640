import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gcn_drug_repurposing_amd import synth, embio, trainer
adj, nt, names = synth.whole_graph_standin(seed=3, scale=4)
n = adj.shape[0]; d = 128
X = synth.gaussian_features(n, d, 1) / 11.0
os.makedirs("/tmp/fb", exist_ok=True)
embio.write_embs("/tmp/fb/in.embs.txt", names, X)
coo = adj.tocoo()
with open("/tmp/fb/g.edgelist", "w") as f:
    for u, v, w in zip(coo.row, coo.col, coo.data):
        f.write(f"{names[u]} {names[v]} {float(w)!r}\n")
t0 = time.time()
eng = trainer.main(["--emb-file", "/tmp/fb/in.embs.txt", "--adj-file", "/tmp/fb/g.edgelist", "--num-layers", "3", "--hidden-units", "128",
                    "--epochs", "3", "--lr", "0.0003", "--beta-percentile", "98", "--batch-size", "0", "--seed", "1", "--out", "/tmp/fb/o.txt", "--log-loss"])
e = np.loadtxt("/tmp/fb/o.txt")
print("full-batch L=3 ok", e.shape, np.isfinite(e).all(), abs(np.sqrt((e**2).sum(1)) - 1).max(), "time", time.time() - t0)
