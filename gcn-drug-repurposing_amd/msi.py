"""The multiscale-interactome graph straight from its TSV edge tables to CSR.

Replaces, for the purpose of feeding the trainer, multiscale/msi/msi.py:109-146 (MSI.load_graph: five node_1/node_2
edge tables -> undirected networkx graph -> .to_directed()), :230-262 (class-specific adjacency + weight_graph:
w(u -> v) = W[class(v)] / #{successors of u in that class}, pathway -> pathway split into up / down by the direction
of the GO hierarchy table), predict_drug.py:182-196 (NodeCovid <-> pathway edges) and :224-226
(nx.write_weighted_edgelist).  No networkx/pandas: insertion-ordered dicts reproduce networkx's node and adjacency
order, so node order (== row order of every .embs.txt derived from the graph) and the edgelist text are identical.
"""
from __future__ import annotations

import csv

import numpy as np
import scipy.sparse as sp

DRUG, INDICATION, PROTEIN, FUNCTIONAL_PATHWAY = "drug", "indication", "protein", "functional_pathway"
UP, DOWN = "up_functional_pathway", "down_functional_pathway"
COMPONENTS = (  # (name, from type, to type) in MSI.load_graph order
    ("drug_to_protein", DRUG, PROTEIN),
    ("indication_to_protein", INDICATION, PROTEIN),
    ("protein_to_protein", PROTEIN, PROTEIN),
    ("protein_to_functional_pathway", PROTEIN, FUNCTIONAL_PATHWAY),
    ("functional_pathway_to_functional_pathway", FUNCTIONAL_PATHWAY, FUNCTIONAL_PATHWAY),
)
COVID_WEIGHTS = {  # predict_drug.py:173-180
    DOWN: 4.4863053901688685, INDICATION: 3.541889556309463, FUNCTIONAL_PATHWAY: 6.583155399238509,
    UP: 2.09685000906964, PROTEIN: 4.396695660380823, DRUG: 3.2071696595616364,
}


def read_node_table(path):
    """node_1 / node_2 columns of one MSI edge table (multiscale/msi/node_to_node.py:85-95)"""
    with open(path, newline="") as f:
        rows = csv.reader(f, delimiter="\t")
        header = next(rows)
        i1, i2 = header.index("node_1"), header.index("node_2")
        return [(r[i1], r[i2]) for r in rows if len(r) > max(i1, i2)]


class MsiGraph:
    def __init__(self):
        self.adj = {}        # node -> {successor: weight or None}, insertion ordered like networkx
        self.type = {}
        self.up = {}         # pathway -> set of parents (node_1 -> node_2 rows of the GO table)
        self.down = {}
        self.drug_or_indication2proteins = {}   # MSI.load_drug_or_indication2proteins (msi.py:192-205)

    # -- MSI.load_graph ------------------------------------------------------------------------------------
    def _add_edge(self, u, v):
        for a, b in ((u, v), (v, u)):
            if a not in self.adj:
                self.adj[a] = {}
            if b not in self.adj:
                self.adj[b] = {}
        if v not in self.adj[u]:
            self.adj[u][v] = None
        if u not in self.adj[v]:
            self.adj[v][u] = None

    def load(self, files):
        """files: {component name: path}; missing components are skipped (MSI(nodes=..., edges=...) subsets)"""
        for name, t_from, t_to in COMPONENTS:
            if name not in files or files[name] is None:
                continue
            for u, v in read_node_table(files[name]):
                self._add_edge(u, v)
                self.type[u] = t_from
                self.type[v] = t_to
                if name in ("drug_to_protein", "indication_to_protein"):
                    self.drug_or_indication2proteins.setdefault(u, set()).add(v)
                if name == "functional_pathway_to_functional_pathway":
                    self.up.setdefault(u, set()).add(v)
                    self.down.setdefault(v, set()).add(u)
        return self

    # -- MSI.weight_graph ------------------------------------------------------------------------------------
    def _class_of(self, node, succ):
        t = self.type[succ]
        if self.type[node] == FUNCTIONAL_PATHWAY and t == FUNCTIONAL_PATHWAY:
            if succ in self.up.get(node, ()):
                return UP
            if succ in self.down.get(node, ()):
                return DOWN
            raise AssertionError(f"pathway edge {node} -> {succ} is in neither direction of the hierarchy table")
        return t

    def weight_graph(self, weights=COVID_WEIGHTS):
        for node, succs in self.adj.items():
            cls = {s: self._class_of(node, s) for s in succs}
            count = {}
            for c in cls.values():
                count[c] = count.get(c, 0) + 1
            for s in succs:
                succs[s] = weights[cls[s]] / float(count[cls[s]])
        return self

    # -- predict_drug.py:182-196 ------------------------------------------------------------------------------------
    def add_covid_pathway_edges(self, pathway_ids, covid="NodeCovid"):
        ids = list(set(pathway_ids))
        w = 3.0 / len(ids)
        for p in ids:
            if p in self.adj:
                if covid not in self.adj:
                    self.adj[covid] = {}
                    self.type.setdefault(covid, INDICATION)
                self.adj[covid][p] = w
                self.adj[p][covid] = w
        return self

    # -- outputs ------------------------------------------------------------------------------------
    @property
    def names(self):
        return list(self.adj)

    @property
    def drugs_in_graph(self):        # msi.py:186-187 (order there is a set's; here graph order)
        return [n for n in self.adj if self.type[n] == DRUG]

    @property
    def indications_in_graph(self):  # msi.py:189-190
        return [n for n in self.adj if self.type[n] == INDICATION and n in self.drug_or_indication2proteins]

    def write_weighted_edgelist(self, path):
        """nx.write_weighted_edgelist text: 'u v w' per directed edge, in node / adjacency order"""
        with open(path, "w") as f:
            for u, succs in self.adj.items():
                for v, w in succs.items():
                    f.write(f"{u} {v} {1.0 if w is None else w}\n")

    def to_csr(self):
        """-> (adj CSR fp64 [N, N] with A[u, v] = w(u -> v), names, types)"""
        names = self.names
        idx = {n: i for i, n in enumerate(names)}
        src, dst, w = [], [], []
        for u, succs in self.adj.items():
            iu = idx[u]
            for v, wt in succs.items():
                src.append(iu)
                dst.append(idx[v])
                w.append(1.0 if wt is None else wt)
        n = len(names)
        adj = sp.csr_matrix((np.asarray(w, np.float64), (np.asarray(src), np.asarray(dst))), shape=(n, n))
        adj.sort_indices()
        return adj, names, [self.type[x] for x in names]
