"""File formats of the reference hot path.

  * input  '.embs.txt': first line '<N> <d>', then '<node> v1 ... vd'
    (written by multiscale/openne/node2vec.py:40-47, read at train.py:79-80; names dropped, order kept)
  * output 'graph_embs.txt': np.savetxt default '%.18e', space separated, no header, row order == input
    order (train.py:193); consumers load it with np.loadtxt (predict_drug.py:52).
  * adjacency 'u v w' weighted edgelist written by nx.write_weighted_edgelist (predict_drug.py:224-226)
    and '.sif' ('src 1 dst', 2016data/toy.sif).
"""
from __future__ import annotations

import numpy as np


def read_embs(path):
    """-> (names: list[str], X: float64 [N, d])"""
    names, rows = [], []
    with open(path) as f:
        header = f.readline().split()
        for line in f:
            parts = line.split()
            if not parts:
                continue
            names.append(parts[0])
            rows.append(np.array(parts[1:], dtype=np.float64))
    x = np.vstack(rows) if rows else np.zeros((0, 0))
    if len(header) == 2 and header[0].isdigit() and int(header[0]) != len(names):
        raise ValueError(f"{path}: header says {header[0]} nodes, file has {len(names)}")
    return names, x


def write_embs(path, names, x):
    """the node2vec.py:40-47 writer, for tests and synthetic inputs"""
    with open(path, "w") as f:
        f.write(f"{len(names)} {x.shape[1]}\n")
        for n, v in zip(names, x):
            f.write(f"{n} {' '.join(str(t) for t in v)}\n")


def write_graph_embs(path, emb):
    np.savetxt(path, np.asarray(emb))


def read_edgelist(path, names=None):
    """'u v [w]' or sif 'u <rel> v' lines -> (src, dst, w, names).  Node ids are strings; with `names`
    given (the .embs.txt row order) ids are mapped to those rows, otherwise in order of appearance.
    A .sif file is undirected: both directions are emitted."""
    sif = path.endswith(".sif") or path.endswith(".sif.lcc")
    index = {n: i for i, n in enumerate(names)} if names is not None else {}
    fixed = names is not None
    names = list(names) if names is not None else []
    src, dst, w = [], [], []

    def node(tok):
        i = index.get(tok)
        if i is None:
            if fixed:
                raise KeyError(f"{path}: node '{tok}' is not in the embedding file")
            i = index[tok] = len(names)
            names.append(tok)
        return i

    with open(path) as f:
        for line in f:
            p = line.split()
            if not p or p[0].startswith("#"):
                continue
            if sif:
                u, v, wt = node(p[0]), node(p[2]), 1.0
                src += [u, v]
                dst += [v, u]
                w += [wt, wt]
            else:
                src.append(node(p[0]))
                dst.append(node(p[1]))
                w.append(float(p[2]) if len(p) > 2 else 1.0)
    return np.asarray(src, np.int64), np.asarray(dst, np.int64), np.asarray(w, np.float64), names
