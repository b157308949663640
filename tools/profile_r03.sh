#!/bin/bash
# Round 3: the evidence behind bench.py's line, collected in one go on the GPU box (run from the repo root):
#   1. the default bench line                                    -> gpurun_out/r03/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command      -> gpurun_out/r03/kernel_stats.csv   (per-kernel average durations)
#   3. memory-side counters of the two forward SpMMs, one counter group per pass (FETCH_SIZE / WRITE_SIZE / TCC hit-miss-req)
#      -> gpurun_out/pmc/r03_wg_{fwd1,plain}_*  -> tools/pmc_pack_r02.py r03 -> profiles/r03_spmm_pmc.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r03; mkdir -p $O
python3 $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err; echo "rocprof rc=$?"
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cd $R
for spec in "wg_fwd1 whole_graph fwd1 5" "wg_plain whole_graph plain 5"; do
  set -- $spec
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    tag="r03_$1_$(echo $grp | cut -d' ' -f1)"
    bash $R/tools/pmc_run.sh "$tag" "$grp" tools/spmm_prof.py 2 128 $4 $2 $3 | grep -v "^$"
  done
done
python3 $R/tools/pmc_pack_r02.py r03
cp $R/profiles/r03_spmm_pmc.json $O/
