#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4u; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o /tmp/store_pattern > $O/build.txt 2>&1 || { cat $O/build.txt; exit 1; }
timeout -k 10 120 /tmp/store_pattern 29960 > $O/store_pattern.txt 2>&1; echo "rc=$?"
timeout -k 10 120 /tmp/store_pattern 1000000 >> $O/store_pattern.txt 2>&1; echo "rc=$?"
cat $O/store_pattern.txt
