#!/bin/bash
# Round 6, extra kernel statistics: the trainer's step at config 2 (lazy, layer 1 kept) and the full + lazy steps of RMAT 10M / 200M on one GPU
# (where the row-filtered products run in their listed-workgroup form) under rocprofv3 --kernel-trace --stats -> gpurun_out/r06/*_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r06; mkdir -p $O
prof() {
  local name=$1; shift
  ( cd /tmp; export TMPDIR=/tmp; rm -rf $O/prof_$name
    timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 "$@" > $O/${name}_under_rocprof.txt 2> $O/${name}_rocprof.err; echo "rocprof $name rc=$?" )
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv && rm -rf $O/prof_$name
}
prof lazy_step $R/tools/lazy_only_prof.py
prof rmat $R/bench.py --workload rmat:10000000:200000000 --steps 5 --warmup 2 --no-cpu-baseline
