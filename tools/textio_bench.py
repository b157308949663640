#!/usr/bin/env python3
"""host text I/O of train.py at config-2 size (N = 29,960, d = 128): the native reader / writer of libgssgcn.so against the numpy /
Python ones they replace (same bytes, same doubles)"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from gcn_drug_repurposing_amd import embio  # noqa: E402

n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (29960, 128)
rng = np.random.RandomState(0)
x = rng.randn(n, d) / 7
emb = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
names = [str(1000 + i) if i % 3 else f"GO:{i:07d}" for i in range(n)]
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "e.embs.txt")
    embio.write_embs(p, names, x)
    for thr in (1, 8, 0):
        t = time.time()
        nm, X = embio.read_embs(p, threads=thr)
        print(f"read_embs  native threads={thr or 'all'}: {time.time() - t:.3f} s  equal={nm == names and np.array_equal(X, x)}")
    t = time.time()
    rows = []
    with open(p) as f:
        f.readline()
        for line in f:
            parts = line.split()
            rows.append(np.array(parts[1:], dtype=np.float64))
    np.vstack(rows)
    print(f"read_embs  python line parser: {time.time() - t:.3f} s")
    a, b = os.path.join(td, "a.txt"), os.path.join(td, "b.txt")
    t = time.time()
    np.savetxt(a, emb)
    print(f"write      np.savetxt: {time.time() - t:.3f} s")
    for thr in (1, 8, 0):
        t = time.time()
        embio.write_graph_embs(b, emb, threads=thr)
        print(f"write      native threads={thr or 'all'}: {time.time() - t:.3f} s  identical={open(a, 'rb').read() == open(b, 'rb').read()}")
# the whole_graph edgelist ('u v w', 958,068 lines) against the .embs.txt names
from gcn_drug_repurposing_amd import synth  # noqa: E402
adj, _, gnames = synth.whole_graph_standin(1)
coo = adj.tocoo()
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "g.edgelist")
    with open(p, "w") as f:
        for u, v, w in zip(coo.row, coo.col, coo.data):
            f.write(f"{gnames[u]} {gnames[v]} {float(w)!r}\n")
    t = time.time()
    src, dst, w, _ = embio.read_edgelist(p, gnames)
    print(f"edgelist   native: {time.time() - t:.3f} s  equal={np.array_equal(src, coo.row) and np.array_equal(dst, coo.col) and np.array_equal(w, coo.data)}")
    t = time.time()
    index = {n: i for i, n in enumerate(gnames)}
    s2, d2, w2 = [], [], []
    with open(p) as f:
        for line in f:
            q = line.split()
            s2.append(index[q[0]]); d2.append(index[q[1]]); w2.append(float(q[2]))
    print(f"edgelist   python loop: {time.time() - t:.3f} s")
print("host cpus", os.cpu_count())
