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


def read_embs(path, threads=0):
    """-> (names: list[str], X: float64 [N, d]).  The native reader of libgssgcn.so parses the lines on all host cores with
    strtod (correctly rounded, i.e. the values float() gives); a file it does not accept (ragged rows, odd tokens) goes to the
    line-by-line parser below, which raises the errors."""
    try:
        from . import _lib
        import ctypes as C
        lib = _lib.load()
        h = C.c_void_p()
        if lib.gss_embs_open(C.byref(h), str(path).encode(), int(threads)) == 0:
            try:
                n, d, nb = lib.gss_embs_rows(h), lib.gss_embs_cols(h), lib.gss_embs_names_bytes(h)
                x = np.empty((n, d), dtype=np.float64)
                buf = C.create_string_buffer(max(int(nb), 1))
                hdr = C.c_int64(-1)
                _lib.check(lib.gss_embs_copy(h, x.ctypes.data, buf, nb, C.byref(hdr)), "gss_embs_copy")
                names = buf.raw[:nb].decode().split("\n")[:n] if n else []
                if hdr.value >= 0 and hdr.value != n:
                    raise ValueError(f"{path}: header says {hdr.value} nodes, file has {n}")
                return names, x
            finally:
                lib.gss_embs_close(h)
    except ValueError:
        raise
    except Exception:  # noqa: BLE001  (library not built, ...): the plain parser decides
        pass
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


def write_graph_embs(path, emb, threads=0):
    """np.savetxt(path, emb) (train.py:193).  float32 input (what the trainer produces) goes through the native writer of
    libgssgcn.so -- the same bytes, formatted exactly on all host cores; anything else through numpy itself."""
    emb = np.asarray(emb)
    if emb.dtype == np.float32 and emb.ndim == 2 and emb.shape[1] >= 1:
        from . import _lib
        emb = np.ascontiguousarray(emb)
        _lib.check(_lib.load().gss_write_embs_text(str(path).encode(), emb.ctypes.data, emb.shape[0], emb.shape[1], int(threads)),
                   "gss_write_embs_text")
        return
    np.savetxt(path, emb)


def read_edgelist(path, names=None):
    """'u v [w]' or sif 'u <rel> v' lines -> (src, dst, w, names).  Node ids are strings; with `names`
    given (the .embs.txt row order) ids are mapped to those rows, otherwise in order of appearance.
    A .sif file is undirected: both directions are emitted."""
    sif = path.endswith(".sif") or path.endswith(".sif.lcc")
    if not sif and names is not None and len(names) > 0 and not any("\n" in n for n in names):
        # the trainer's case (ids mapped to the .embs.txt rows): native multi-threaded parser; a file it rejects (an unknown id,
        # a malformed weight) goes through the Python loop below, which raises the error with its context
        try:
            from . import _lib
            import ctypes as C
            lib = _lib.load()
            blob = "\n".join(names).encode()
            h = C.c_void_p()
            if lib.gss_edgelist_open(C.byref(h), str(path).encode(), blob, len(blob), len(names), 0) == 0:
                try:
                    if lib.gss_edgelist_bad_line(h) < 0:
                        m = lib.gss_edgelist_edges(h)
                        src32, dst32, w = np.empty(m, np.int32), np.empty(m, np.int32), np.empty(m, np.float64)
                        _lib.check(lib.gss_edgelist_copy(h, src32.ctypes.data, dst32.ctypes.data, w.ctypes.data), "gss_edgelist_copy")
                        return src32.astype(np.int64), dst32.astype(np.int64), w, list(names)
                finally:
                    lib.gss_edgelist_close(h)
        except Exception:  # noqa: BLE001  (library not built ...): the plain parser decides
            pass
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
