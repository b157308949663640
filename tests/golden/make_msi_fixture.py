#!/usr/bin/env python3
"""Golden fixture for the MSI loader: run the REFERENCE's MSI class (multiscale/msi/msi.py) on a small synthetic
set of edge tables (written here, committed under tests/golden/msi_small/) and record its node order and its
weighted edgelist text.  Run in the build container only:  python tests/golden/make_msi_fixture.py"""
import os
import sys

import numpy as np

REF = os.environ.get("GSS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "msi_small")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "multiscale"))
sys.path.insert(0, REF)

HEADER = "node_1\tnode_2\tnode_1_type\tnode_2_type\tnode_1_name\tnode_2_name\n"


def write_table(name, rows, t1, t2):
    with open(os.path.join(OUT, name + ".tsv"), "w") as f:
        f.write(HEADER)
        for a, b in rows:
            f.write(f"{a}\t{b}\t{t1}\t{t2}\tn_{a}\tn_{b}\n")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.RandomState(0)
    drugs = [f"DB{i:05d}" for i in range(12)]
    inds = [f"C{i:07d}" for i in range(8)] + ["NodeCovid"]
    prots = [str(100 + i) for i in range(60)]
    paths = [f"GO:{i:07d}" for i in range(30)]

    def pairs(a, b, m, same=False):
        out = set()
        while len(out) < m:
            u, v = a[rng.randint(len(a))], b[rng.randint(len(b))]
            if u != v and (not same or (v, u) not in out):
                out.add((u, v))
        return sorted(out, key=lambda _: rng.rand())
    write_table("drug_to_protein", pairs(drugs, prots, 40), "drug", "protein")
    write_table("indication_to_protein", pairs(inds, prots, 45), "indication", "protein")
    write_table("protein_to_protein", pairs(prots, prots, 150, same=True), "protein", "protein")
    write_table("protein_to_functional_pathway", pairs(prots, paths, 70), "protein", "functional_pathway")
    hier = [(paths[i], paths[rng.randint(0, i)]) for i in range(1, len(paths))] + [(paths[9], paths[2]), (paths[20], paths[3])]
    write_table("functional_pathway_to_functional_pathway", sorted(set(hier)), "functional_pathway", "functional_pathway")

    import networkx as nx
    from msi.msi import MSI  # the reference
    p = lambda n: os.path.join(OUT, n + ".tsv")  # noqa: E731
    msi = MSI(drug2protein_file_path=p("drug_to_protein"), indication2protein_file_path=p("indication_to_protein"),
              protein2protein_file_path=p("protein_to_protein"), protein2functional_pathway_file_path=p("protein_to_functional_pathway"),
              functional_pathway2functional_pathway_file_path=p("functional_pathway_to_functional_pathway"))
    msi.load()
    weights = {'down_functional_pathway': 4.4863053901688685, 'indication': 3.541889556309463,
               'functional_pathway': 6.583155399238509, 'up_functional_pathway': 2.09685000906964,
               'protein': 4.396695660380823, 'drug': 3.2071696595616364}      # predict_drug.py:173-180
    msi.weight_graph(weights)
    nx.write_weighted_edgelist(msi.graph, os.path.join(OUT, "expected.weighted.edgelist"))
    # predict_drug.py:182-196 with a pathway id list
    ids = [paths[1], paths[5], paths[7], "GO:9999999", paths[5]]
    uniq = list(set(ids))
    for pw in uniq:
        if pw in msi.graph.nodes:
            msi.graph.add_edge('NodeCovid', pw, weight=3.0 / len(uniq))
            msi.graph.add_edge(pw, 'NodeCovid', weight=3.0 / len(uniq))
    nx.write_weighted_edgelist(msi.graph, os.path.join(OUT, "expected_pathway.weighted.edgelist"))
    with open(os.path.join(OUT, "expected_nodes.txt"), "w") as f:
        for n in msi.graph.nodes():
            f.write(f"{n}\t{msi.graph.nodes[n]['type']}\n")
    with open(os.path.join(OUT, "pathway_ids.txt"), "w") as f:
        f.write("\n".join(ids) + "\n")
    print("nodes", msi.graph.number_of_nodes(), "directed edges", msi.graph.number_of_edges())


if __name__ == "__main__":
    main()
