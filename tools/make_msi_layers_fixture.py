#!/usr/bin/env python3
"""Packs the four non-PPI edge tables of the multiscale interactome that the reference ships
(/root/reference/data/{drug_to_protein, indication_to_protein, protein_to_functional_pathway,
functional_pathway_to_functional_pathway}.tsv, + covid_to_protein.tsv, the pathway ids of
config_gcn_pathway.json's perturbation file and the drug-indication pairs of data/drug_indication_df.tsv, the labels of
evaluate_auc.py) into a data-only fixture:

    gcn-drug-repurposing_amd/data/msi_real_layers.npz

Contents: per node type the node ids in order of first appearance (newline-joined bytes), per table the
(node_1, node_2) columns as int32 indices into those lists, in file row order.  No reference source text, only
the tables' two id columns.  Runs in the build container only (the reference tree does not exist on the GPU
box); SURVEY.md section 8(d) prescribes these four layers verbatim for the whole_graph stand-in, with only the
absent PPI layer (.MISSING_LARGE_BLOBS) synthesised -- see synth.whole_graph_standin.

    python tools/make_msi_layers_fixture.py [/root/reference]
"""
import csv
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(ROOT, "gcn-drug-repurposing_amd", "data", "msi_real_layers.npz")

TABLES = (  # (file stem, type of node_1, type of node_2)
    ("drug_to_protein", "drug", "protein"),
    ("indication_to_protein", "indication", "protein"),
    ("covid_to_protein", "indication", "protein"),
    ("protein_to_functional_pathway", "protein", "pathway"),
    ("functional_pathway_to_functional_pathway", "pathway", "pathway"),
)


def read_pairs(path):
    with open(path, newline="") as f:
        rows = csv.reader(f, delimiter="\t")
        header = next(rows)
        i1, i2 = header.index("node_1"), header.index("node_2")
        return [(r[i1], r[i2]) for r in rows if len(r) > max(i1, i2)]


def main():
    ids = {t: {} for t in ("drug", "indication", "protein", "pathway")}
    out = {}
    for stem, t1, t2 in TABLES:
        pairs = read_pairs(os.path.join(REF, "data", stem + ".tsv"))
        u = np.empty(len(pairs), np.int32)
        v = np.empty(len(pairs), np.int32)
        for k, (a, b) in enumerate(pairs):
            u[k] = ids[t1].setdefault(a, len(ids[t1]))
            v[k] = ids[t2].setdefault(b, len(ids[t2]))
        out[stem + "_u"], out[stem + "_v"] = u, v
    # predict_drug.py:182-196 with config_gcn_pathway.json: NodeCovid <-> pathway edges for the ids of the
    # perturbation table's Pathway_ID column that are nodes of the graph, weight 3 / (number of distinct ids)
    with open(os.path.join(REF, "data", "04_Immune_Genes_Enrichment_GO_MF_BP_Intersection.tsv"), newline="") as f:
        rows = csv.reader(f, delimiter="\t")
        header = next(rows)
        col = header.index("Pathway_ID")
        pw = sorted({r[col] for r in rows if len(r) > col})
    out["covid_pathway_total"] = np.int64(len(pw))
    out["covid_pathway_idx"] = np.asarray(sorted(ids["pathway"][p] for p in pw if p in ids["pathway"]), np.int32)
    # evaluate_auc.py:156-170's labels: data/drug_indication_df.tsv (drug, indication) pairs whose two nodes are in the graph
    with open(os.path.join(REF, "data", "drug_indication_df.tsv"), newline="") as f:
        rows = csv.reader(f, delimiter="\t")
        header = next(rows)
        di, ii = header.index("drug"), header.index("indication")
        pairs = sorted({(ids["drug"][r[di]], ids["indication"][r[ii]]) for r in rows
                        if len(r) > max(di, ii) and r[di] in ids["drug"] and r[ii] in ids["indication"]})
    out["drug_indication_drug"] = np.asarray([a for a, _ in pairs], np.int32)
    out["drug_indication_indication"] = np.asarray([b for _, b in pairs], np.int32)
    for t, table in ids.items():
        out["names_" + t] = np.frombuffer("\n".join(table).encode(), dtype=np.uint8)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
    print(f"wrote {OUT}: {os.path.getsize(OUT)} bytes; nodes " + ", ".join(f"{t} {len(v)}" for t, v in ids.items()))


if __name__ == "__main__":
    main()
