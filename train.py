#!/usr/bin/env python3
"""Drop-in for the reference's train.py (same flags, reads --emb-file, writes ./graph_embs.txt)."""
from gcn_drug_repurposing_amd.trainer import main

if __name__ == '__main__':
    main()
