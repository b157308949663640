"""Seeded synthetic stand-ins for graphs the reference does not ship (.MISSING_LARGE_BLOBS: the
whole_graph edgelist, the PPI table and every .embs.txt are absent) and the RMAT stress graph.

whole_graph_standin follows the multiscale interactome's published shape:
1,661 drugs, 840 indications, 17,660 proteins, 9,798 functional pathways + 'NodeCovid' = 29,960 nodes;
undirected layer sizes drug-protein 8,568, indication-protein 25,212, covid-protein 306, protein-protein
387,626, protein-pathway 34,777, pathway-pathway 22,545 = 479,034 edges = 958,068 directed entries.
Edge weights follow MSI.weight_graph (multiscale/msi/msi.py:255-262): w(u->v) = W[type(v)] / #{successors
of u with that type}, pathway->pathway split into up/down, type weights from predict_drug.py:173-180.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

N_DRUG, N_IND, N_PROT, N_PATH = 1661, 840, 17660, 9798
TYPE_W = {"drug": 3.2071696595616364, "indication": 3.541889556309463, "protein": 4.396695660380823,
          "functional_pathway": 6.583155399238509, "up_functional_pathway": 2.09685000906964,
          "down_functional_pathway": 4.4863053901688685}
LAYERS = {"drug_protein": 8568, "indication_protein": 25212, "covid_protein": 306, "protein_protein": 387626,
          "protein_pathway": 34777, "pathway_pathway": 22545}


def _powerlaw_p(n, gamma, rng):
    w = (np.arange(n) + 10.0) ** (-gamma)
    rng.shuffle(w)
    return w / w.sum()


def _sample_pairs(m, a_ids, a_p, b_ids, b_p, rng, same_set=False):
    """m unique unordered (same_set) or ordered-role (bipartite) pairs"""
    seen = np.zeros(0, dtype=np.int64)
    nb = int(max(a_ids.max(), b_ids.max())) + 1
    while len(seen) < m:
        k = int((m - len(seen)) * 1.3) + 64
        u = rng.choice(a_ids, k, p=a_p)
        v = rng.choice(b_ids, k, p=b_p)
        if same_set:
            keep = u != v
            u, v = u[keep], v[keep]
            lo, hi = np.minimum(u, v), np.maximum(u, v)
            key = lo.astype(np.int64) * nb + hi
        else:
            key = u.astype(np.int64) * nb + v
        seen = np.unique(np.concatenate([seen, key]))
        if len(seen) > m:
            seen = rng.permutation(seen)[:m]
    return seen // nb, seen % nb


def whole_graph_standin(seed=1, pathway_edges=False, scale=1):
    """-> (adj CSR fp64 [N, N] directed+weighted, node type array, names); scale > 1 divides every node and
    edge count (small graphs of the same shape for tests)"""
    rng = np.random.RandomState(seed)
    N_DRUG, N_IND, N_PROT, N_PATH = (max(4, v // scale) for v in (1661, 840, 17660, 9798))
    LAYERS = {k: max(8, v // scale) for k, v in globals()["LAYERS"].items()}
    o_drug, o_ind, o_prot, o_path = 0, N_DRUG, N_DRUG + N_IND, N_DRUG + N_IND + N_PROT
    covid = o_path + N_PATH
    n = covid + 1
    drug = np.arange(o_drug, o_ind); ind = np.arange(o_ind, o_prot)
    prot = np.arange(o_prot, o_path); path = np.arange(o_path, covid)
    p_prot = _powerlaw_p(N_PROT, 0.75, rng)
    und = []  # (u, v, kind)
    u, v = _sample_pairs(LAYERS["drug_protein"], drug, _powerlaw_p(N_DRUG, 0.6, rng), prot, p_prot, rng)
    und.append((u, v))
    u, v = _sample_pairs(LAYERS["indication_protein"], ind, _powerlaw_p(N_IND, 0.8, rng), prot, p_prot, rng)
    und.append((u, v))
    cp = rng.choice(prot, LAYERS["covid_protein"], replace=False, p=p_prot)
    und.append((np.full(len(cp), covid), cp))
    u, v = _sample_pairs(LAYERS["protein_protein"], prot, p_prot, prot, p_prot, rng, same_set=True)
    und.append((u, v))
    u, v = _sample_pairs(LAYERS["protein_pathway"], prot, p_prot, path, _powerlaw_p(N_PATH, 0.7, rng), rng)
    und.append((u, v))
    # pathway hierarchy: child -> parent with parent id < child id (a DAG like GO)
    child, parent = _sample_pairs(LAYERS["pathway_pathway"], path, _powerlaw_p(N_PATH, 0.3, rng), path,
                                  _powerlaw_p(N_PATH, 0.9, rng), rng, same_set=True)
    child, parent = np.maximum(child, parent), np.minimum(child, parent)
    if pathway_edges:
        # config_gcn_pathway.json: 324 NodeCovid <-> pathway edges, w = 3/353 (predict_drug.py:182-196)
        n_extra = min(324, len(path))
        extra = rng.choice(path, n_extra, replace=False)
    ntype = np.empty(n, dtype=np.int8)   # 0 drug 1 indication 2 protein 3 pathway
    ntype[drug] = 0; ntype[ind] = 1; ntype[prot] = 2; ntype[path] = 3; ntype[covid] = 1
    src = np.concatenate([np.concatenate([a, b]) for a, b in und] + [child, parent])
    dst = np.concatenate([np.concatenate([b, a]) for a, b in und] + [parent, child])
    # class of the successor: 0..3 = node type, 4 = up pathway (child -> parent), 5 = down pathway
    cls = ntype[dst].astype(np.int64)
    n_und = sum(2 * len(a) for a, _ in und)
    cls[n_und:n_und + len(child)] = 4
    cls[n_und + len(child):] = 5
    tw = np.array([TYPE_W["drug"], TYPE_W["indication"], TYPE_W["protein"], TYPE_W["functional_pathway"],
                   TYPE_W["up_functional_pathway"], TYPE_W["down_functional_pathway"]])
    cnt = np.zeros((n, 6), dtype=np.int64)
    np.add.at(cnt, (src, cls), 1)
    w = tw[cls] / cnt[src, cls]
    if pathway_edges:
        src = np.concatenate([src, np.full(n_extra, covid), extra])
        dst = np.concatenate([dst, extra, np.full(n_extra, covid)])
        w = np.concatenate([w, np.full(2 * n_extra, 3.0 / 353.0)])
    adj = sp.csr_matrix((w, (src, dst)), shape=(n, n))
    adj.sort_indices()
    names = ([f"DB{i:05d}" for i in range(N_DRUG)] + [f"C{i:07d}" for i in range(N_IND)] + [str(1000 + i) for i in range(N_PROT)]
             + [f"GO:{i:07d}" for i in range(N_PATH)] + ["NodeCovid"])
    return adj, ntype, names


def gaussian_features(n, d, seed):
    return np.random.RandomState(seed).randn(n, d).astype(np.float32)


def rmat_adj(scale_nodes, n_edges, seed=4, abcd=(0.57, 0.19, 0.19, 0.05), chunk=1 << 24):
    """RMAT directed graph on n = scale_nodes ids (sampled on the next power of two, out-of-range endpoints
    rejected), `n_edges` unique entries, unit weights, no self loops."""
    rng = np.random.RandomState(seed)
    bits = int(np.ceil(np.log2(scale_nodes)))
    a, b, c, _ = abcd
    keys = np.zeros(0, dtype=np.int64)
    while len(keys) < n_edges:
        k = min(chunk, int((n_edges - len(keys)) * 1.4) + 1024)
        u = np.zeros(k, dtype=np.int64); v = np.zeros(k, dtype=np.int64)
        for _ in range(bits):
            r = rng.rand(k)
            right = (r >= a) & (r < a + b) | (r >= a + b + c)
            down = r >= a + b
            u = (u << 1) | down
            v = (v << 1) | right
        keep = (u < scale_nodes) & (v < scale_nodes) & (u != v)
        keys = np.unique(np.concatenate([keys, u[keep] * scale_nodes + v[keep]]))
    if len(keys) > n_edges:
        keys = np.sort(rng.permutation(keys)[:n_edges])
    src, dst = keys // scale_nodes, keys % scale_nodes
    adj = sp.csr_matrix((np.ones(len(src)), (src, dst)), shape=(scale_nodes, scale_nodes))
    adj.sort_indices()
    return adj
