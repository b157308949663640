#!/bin/bash
# Round 4: the evidence behind bench.py's line, collected on the GPU box (run from the repo root):
#   1. the default bench line                                                      -> gpurun_out/r04/bench.json
#   2. rocprofv3 --kernel-trace --stats of the FULL steps only (--no-lazy-top)     -> gpurun_out/r04/kernel_stats.csv
#      and of the lazy steps only (tools/lazy_only_prof.py)                         -> gpurun_out/r04/lazy_kernel_stats.csv
#      (round 3's csv mixed the two: PLAIN's launch time could not be read off it)
#   3. memory-side counters of the two forward SpMMs at config 2, one counter group per pass
#      -> gpurun_out/pmc/r04_wg_{fwd1,plain}_*  -> tools/pmc_pack_r02.py r04 -> profiles/r04_spmm_pmc.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r04; mkdir -p $O
python3 $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-lazy-top > $O/bench_under_rocprof.json 2> $O/rocprof.err; echo "rocprof full rc=$?"
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/prof_lazy; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lazy -- python3 $R/tools/lazy_only_prof.py > $O/lazy_only.txt 2> $O/rocprof_lazy.err; echo "rocprof lazy rc=$?"
cp $(find $O/prof_lazy -name "*kernel_stats.csv" | head -1) $O/lazy_kernel_stats.csv
cd $R
for spec in "wg_fwd1 whole_graph fwd1 5" "wg_plain whole_graph plain 5"; do
  set -- $spec
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    tag="r04_$1_$(echo $grp | cut -d' ' -f1)"
    bash $R/tools/pmc_run.sh "$tag" "$grp" tools/spmm_prof.py 2 128 $4 $2 $3 | grep -v "^$"
  done
done
python3 $R/tools/pmc_pack_r02.py r04
cp $R/profiles/r04_spmm_pmc.json $O/
