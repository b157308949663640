"""Stand-ins for graphs the reference does not ship (.MISSING_LARGE_BLOBS: the whole_graph edgelist, the PPI
table and every .embs.txt are absent) and the RMAT stress graph.

whole_graph_standin: the four MSI layers the reference DOES ship are used verbatim --
drug-protein 8,568, indication-protein 25,212 (+ 306 NodeCovid-protein), protein-pathway 34,777 and
pathway-pathway 22,545 edges, packed as index pairs in data/msi_real_layers.npz by tools/make_msi_layers_fixture.py
-- and only the absent protein-protein layer is synthetic: 387,626 seeded power-law pairs over 17,660 proteins
(the 10,345 proteins of the real tables + 7,315 PPI-only ones).  1,661 drugs + 840 indications + NodeCovid +
17,660 proteins + 9,798 pathways = 29,960 nodes, 479,034 undirected edges = 958,068 directed entries.  Node
order is networkx's insertion order over MSI.load_graph's table sequence (multiscale/msi/msi.py:109-146), i.e.
the row order of every .embs.txt derived from the graph; edge weights follow MSI.weight_graph (msi.py:255-262):
w(u->v) = W[class(v)] / #{successors of u in that class}, pathway->pathway split into up/down by the direction
of the hierarchy table, type weights from predict_drug.py:173-180.  (predict_drug.py:169 passes the covid table
IN PLACE of the indication table; SURVEY/BASELINE define the workload with both, as evaluate_auc.py:119's MSI()
+ NodeCovid, and that is what is built here.)  scale > 1 gives small fully synthetic graphs of the same shape.
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


_REAL_LAYERS = None


def real_layers():
    """data/msi_real_layers.npz -> dict with decoded name lists (cached)"""
    global _REAL_LAYERS
    if _REAL_LAYERS is None:
        import os
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "msi_real_layers.npz"))
        d = {k: z[k] for k in z.files}
        for t in ("drug", "indication", "protein", "pathway"):
            d["names_" + t] = bytes(d["names_" + t]).decode().split("\n")
        _REAL_LAYERS = d
    return _REAL_LAYERS


def synthetic_ppi(seed=1, n_real=10345):
    """the absent protein_to_protein.tsv: LAYERS['protein_protein'] unique unordered pairs over N_PROT proteins
    (local ids: [0, n_real) are the proteins of the real tables), power-law endpoint probabilities, seeded; rows
    in random order and orientation -> (u, v) int64"""
    rng = np.random.RandomState(seed)
    prot = np.arange(N_PROT)
    p = _powerlaw_p(N_PROT, 0.75, rng)
    u, v = _sample_pairs(LAYERS["protein_protein"], prot, p, prot, p, rng, same_set=True)
    order = rng.permutation(len(u))
    flip = rng.rand(len(u)) < 0.5
    u, v = u[order], v[order]
    return np.where(flip, v, u), np.where(flip, u, v)


def whole_graph_standin(seed=1, pathway_edges=False, scale=1):
    """-> (adj CSR fp64 [N, N] directed+weighted, node type array, names).  scale == 1: the four real MSI layers +
    a synthetic PPI layer (module docstring); scale > 1: a fully synthetic graph with every node and edge count
    divided by `scale` (small graphs of the same shape for tests)."""
    if scale != 1:
        return _synthetic_standin(seed, pathway_edges, scale)
    R = real_layers()
    n_t = [len(R["names_drug"]), len(R["names_indication"]), N_PROT, len(R["names_pathway"])]   # 1661, 841, 17660, 9798
    off = np.concatenate([[0], np.cumsum(n_t)])
    n = int(off[-1])
    pu, pv = synthetic_ppi(seed, len(R["names_protein"]))
    tables = [  # MSI.load_graph order; (u block ids, v block ids)
        (off[0] + R["drug_to_protein_u"], off[2] + R["drug_to_protein_v"]),
        (off[1] + R["indication_to_protein_u"], off[2] + R["indication_to_protein_v"]),
        (off[1] + R["covid_to_protein_u"], off[2] + R["covid_to_protein_v"]),
        (off[2] + pu, off[2] + pv),
        (off[2] + R["protein_to_functional_pathway_u"], off[3] + R["protein_to_functional_pathway_v"]),
        (off[3] + R["functional_pathway_to_functional_pathway_u"], off[3] + R["functional_pathway_to_functional_pathway_v"]),
    ]
    U = np.concatenate([t[0] for t in tables]).astype(np.int64)
    V = np.concatenate([t[1] for t in tables]).astype(np.int64)
    # node order = first appearance in the stream u0, v0, u1, v1, ... (nx.Graph.add_edge inserts u, then v)
    stream = np.stack([U, V], 1).reshape(-1)
    seen, first = np.unique(stream, return_index=True)
    assert len(seen) == n, (len(seen), n)
    new_id = np.empty(n, np.int64)
    new_id[seen[np.argsort(first)]] = np.arange(n)
    block_type = np.repeat(np.arange(4, dtype=np.int8), n_t)
    ntype = np.empty(n, np.int8)
    ntype[new_id] = block_type
    # directed entries: both directions of every row, duplicates merged (an undirected nx edge exists once)
    src, dst = new_id[np.concatenate([U, V])], new_id[np.concatenate([V, U])]
    key = np.unique(src * n + dst)
    src, dst = key // n, key % n
    cls = ntype[dst].astype(np.int64)            # class of the successor: 0..3 node type, 4 up pathway, 5 down pathway
    pp = (ntype[src] == 3) & (ntype[dst] == 3)
    hu, hv = tables[5]
    up_keys = np.unique(new_id[hu] * n + new_id[hv])      # node_1 -> node_2 rows of the hierarchy table are "up" (msi.py:230-253)
    is_up = np.isin(key, up_keys)
    cls[pp & is_up] = 4
    cls[pp & ~is_up] = 5
    tw = np.array([TYPE_W["drug"], TYPE_W["indication"], TYPE_W["protein"], TYPE_W["functional_pathway"],
                   TYPE_W["up_functional_pathway"], TYPE_W["down_functional_pathway"]])
    cnt = np.zeros((n, 6), dtype=np.int64)
    np.add.at(cnt, (src, cls), 1)
    w = tw[cls] / cnt[src, cls]
    if pathway_edges:
        # config_gcn_pathway.json: NodeCovid <-> pathway edges, w = 3 / 353 (predict_drug.py:182-196), added after weighting
        covid = new_id[off[1] + int(R["covid_to_protein_u"][0])]
        extra = new_id[off[3] + R["covid_pathway_idx"].astype(np.int64)]
        src = np.concatenate([src, np.full(len(extra), covid), extra])
        dst = np.concatenate([dst, extra, np.full(len(extra), covid)])
        w = np.concatenate([w, np.full(2 * len(extra), 3.0 / float(R["covid_pathway_total"]))])
    adj = sp.csr_matrix((w, (src, dst)), shape=(n, n))
    adj.sort_indices()
    block_names = (R["names_drug"] + R["names_indication"] + R["names_protein"]
                   + [f"S{900000000 + i}" for i in range(N_PROT - len(R["names_protein"]))] + R["names_pathway"])
    names = [None] * n
    for b, i in enumerate(new_id):
        names[i] = block_names[b]
    return adj, ntype, names


def standin_drug_indications():
    """evaluate_auc.py:156-170's labels for the stand-in: {indication id: set of drug ids} from the 5,926 (drug, indication) pairs of
    the reference's data/drug_indication_df.tsv (all of them name nodes of the graph: 840 indications, 1,661 drugs)"""
    R = real_layers()
    out = {}
    for a, b in zip(R["drug_indication_drug"], R["drug_indication_indication"]):
        out.setdefault(R["names_indication"][b], set()).add(R["names_drug"][a])
    return out


def standin_tables(seed=1):
    """the stand-in as the five node_1/node_2 tables MSI.load_graph reads (for a test that rebuilds it with
    msi.MsiGraph): {component: [(node_1, node_2), ...]}; the covid rows ride at the end of the indication table"""
    R = real_layers()
    prot = R["names_protein"] + [f"S{900000000 + i}" for i in range(N_PROT - len(R["names_protein"]))]
    pu, pv = synthetic_ppi(seed, len(R["names_protein"]))
    nm = {"drug": R["names_drug"], "indication": R["names_indication"], "protein": prot, "pathway": R["names_pathway"]}

    def rows(stem, t1, t2):
        return [(nm[t1][a], nm[t2][b]) for a, b in zip(R[stem + "_u"], R[stem + "_v"])]
    return {
        "drug_to_protein": rows("drug_to_protein", "drug", "protein"),
        "indication_to_protein": rows("indication_to_protein", "indication", "protein") + rows("covid_to_protein", "indication", "protein"),
        "protein_to_protein": [(prot[a], prot[b]) for a, b in zip(pu, pv)],
        "protein_to_functional_pathway": rows("protein_to_functional_pathway", "protein", "pathway"),
        "functional_pathway_to_functional_pathway": rows("functional_pathway_to_functional_pathway", "pathway", "pathway"),
    }


def _synthetic_standin(seed=1, pathway_edges=False, scale=1):
    """fully synthetic graph of the whole_graph's shape with every node and edge count divided by `scale`"""
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
