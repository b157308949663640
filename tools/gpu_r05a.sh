#!/bin/bash
# Round 5, first call: the GPU suite, the default bench line, and the evidence behind the weight-stationary projection (timings + per-wave stamps).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05a; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 200 python3 $R/tools/proj_ws_bench.py > $O/proj_ws_bench.txt 2>&1
timeout -k 10 200 python3 $R/tools/proj_ws_stamps.py > $O/proj_ws_stamps.txt 2>&1
tail -5 $O/proj_ws_stamps.txt
