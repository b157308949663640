"""The read contract of graph_embs.txt: what the reference's downstream scripts do with the embeddings.

  * predict_drug.graph_embedding (predict_drug.py:49-73): np.loadtxt the GCN file, sklearn normalize(axis=1),
    row i belongs to node_names[i] of the node2vec .embs.txt, score(drug) = e_drug . e_query, rank descending.
  * evaluate_auc.main (evaluate_auc.py:140-170): per indication, ROC-AUC of a score vector over all drugs against
    the drugs indicated for it (data/drug_indication_df.tsv), then median / mean over indications.  The reference's
    GCN branch never fills the score vector it dereferences (dp_saved is only set for method == 'diffusion',
    evaluate_auc.py:131-136,164); the evident intent -- embedding inner products as scores -- is what this does.

Host-side numpy, like the reference's consumers.  This is the harness with which "downstream AUC within 1e-4 of the
CPU reference" is measured: same harness, embeddings from the HIP trainer vs from the CPU path.
"""
from __future__ import annotations

import numpy as np


def load_graph_embs(path):
    """np.loadtxt + row L2 normalisation (predict_drug.py:52-53; sklearn.preprocessing.normalize semantics)"""
    return normalize_rows(np.loadtxt(path))


def normalize_rows(e):
    e = np.asarray(e, dtype=np.float64)
    n = np.sqrt((e * e).sum(1, keepdims=True))
    n[n == 0] = 1.0
    return e / n


def rank_by_query(emb, names, query, candidates):
    """-> (candidate names ranked by e_cand . e_query descending, their scores)   predict_drug.py:55-73"""
    idx = {n: i for i, n in enumerate(names)}
    e = normalize_rows(emb)
    q = e[idx[query]]
    cand = [c for c in candidates if c in idx]
    s = e[[idx[c] for c in cand]] @ q
    order = np.argsort(s)[::-1]
    return [cand[i] for i in order], s[order]


def roc_auc(labels, scores):
    """sklearn.metrics.roc_auc_score for binary labels (rank statistic, ties get average ranks)"""
    labels = np.asarray(labels).astype(bool)
    scores = np.asarray(scores, dtype=np.float64)
    n_pos = int(labels.sum())
    n_neg = len(labels) - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(scores, kind="mergesort")
    s = scores[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return (ranks[labels].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg)


def read_drug_indication_tsv(path):
    """data/drug_indication_df.tsv (columns drug, drug_name, indication, indication_name) -> {indication: set(drugs)}"""
    out = {}
    with open(path) as f:
        header = f.readline().rstrip("\n").split("\t")
        di, ii = header.index("drug"), header.index("indication")
        for line in f:
            p = line.rstrip("\n").split("\t")
            if len(p) > max(di, ii):
                out.setdefault(p[ii], set()).add(p[di])
    return out


def indication_aucs(emb, names, drugs, indications, positives):
    """evaluate_auc.py:156-170 with embedding scores: one ROC-AUC per indication that has >= 1 positive drug in
    the graph (the reference crashes on unknown drugs / single-class vectors; those indications are skipped)."""
    idx = {n: i for i, n in enumerate(names)}
    e = normalize_rows(emb)
    drugs = [d for d in drugs if d in idx]
    dpos = {d: k for k, d in enumerate(drugs)}
    ed = e[[idx[d] for d in drugs]]
    aucs, used = [], []
    for ind in indications:
        if ind not in idx:
            continue
        ref = np.zeros(len(drugs), dtype=int)
        for d in positives.get(ind, ()):
            if d in dpos:
                ref[dpos[d]] = 1
        if ref.sum() == 0 or ref.sum() == len(ref):
            continue
        aucs.append(roc_auc(ref, ed @ e[idx[ind]]))
        used.append(ind)
    return np.asarray(aucs), used


def diffusion_indication_aucs(profiles, names, drugs, indications, positives):
    """evaluate_auc.py:156-161 as written for method == 'diffusion': the score of drug d for an indication is the
    indication's visit probability AT the drug node (profile[indication][idx[d]]), one ROC-AUC per indication.
    profiles: {node: p_visit [N]} (DiffusionProfiles.drug_or_indication2diffusion_profile)."""
    idx = {n: i for i, n in enumerate(names)}
    drugs = [d for d in drugs if d in idx]
    dpos = {d: k for k, d in enumerate(drugs)}
    didx = [idx[d] for d in drugs]
    aucs, used = [], []
    for ind in indications:
        if ind not in profiles:
            continue
        ref = np.zeros(len(drugs), dtype=int)
        for d in positives.get(ind, ()):
            if d in dpos:
                ref[dpos[d]] = 1
        if ref.sum() == 0 or ref.sum() == len(ref):
            continue
        aucs.append(roc_auc(ref, np.asarray(profiles[ind])[didx]))
        used.append(ind)
    return np.asarray(aucs), used


def rank_by_diffusion(profile, names, candidates):
    """predict_drug.py:107-121: candidates (the drug nodes, in node order) ranked by the query node's visit probability at
    them, best first (np.argsort(...)[::-1], ties as numpy breaks them) -> (ranked candidates, their probabilities)"""
    idx = {n: i for i, n in enumerate(names)}
    cand = [c for c in candidates if c in idx]
    prox = np.asarray([profile[idx[c]] for c in cand])
    order = np.argsort(prox)[::-1]
    return [cand[i] for i in order], prox[order]
