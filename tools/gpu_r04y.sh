#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4y; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O3 tools/micro/dma_shape.hip -o /tmp/dma_shape > $O/build.txt 2>&1 || { cat $O/build.txt; exit 1; }
timeout -k 10 120 /tmp/dma_shape 29960 > $O/dma_shape.txt 2>&1; echo "rc=$?"
timeout -k 10 120 /tmp/dma_shape 1000000 >> $O/dma_shape.txt 2>&1; echo "rc=$?"
cat $O/dma_shape.txt
