#!/bin/bash
# round 4: does a captured hipGraph shorten the step? (tools/graph_ab.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4o; mkdir -p $O; cd $R
timeout -k 10 300 python3 tools/graph_ab.py full 12 20 > $O/graph_full.txt 2>&1; echo "rc=$?"; tail -4 $O/graph_full.txt
timeout -k 10 300 python3 tools/graph_ab.py lazy 12 20 > $O/graph_lazy.txt 2>&1; echo "rc=$?"; tail -4 $O/graph_lazy.txt
