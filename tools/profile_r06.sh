#!/bin/bash
# Round 6: the evidence behind bench.py's lines, collected on the GPU box (run from the repo root).  Arguments: which parts to run (default: all)
#   c2 / c3 / c5        bench line (+ rocprofv3 --kernel-trace --stats of the same command) of config 2 / config 3 / RMAT 10M-200M on one GPU
#   c2pmc / c3pmc / c5pmc   memory-side counters of every SpMM mode inside full steps (+ config 5's dense products per product)
#   -> gpurun_out/r06/*;  counters -> gpurun_out/pmc/r06_* -> tools/pmc_pack_r06.py -> profiles/r06_spmm_pmc.json (records the spmm.hip hash)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r06; mkdir -p $O
parts=${@:-c2pmc c2 c3pmc c3 c5pmc c5}     # (counters first: the bench lines read the packed file)
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
prof() {   # <name> <program and args...>: rocprofv3 --kernel-trace --stats, the kernel_stats csv -> $O/<name>_kernel_stats.csv
  local name=$1; shift
  ( cd /tmp; export TMPDIR=/tmp; rm -rf $O/prof_$name
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 "$@" > $O/${name}_under_rocprof.txt 2> $O/${name}_rocprof.err; echo "rocprof $name rc=$?" )
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv && rm -rf $O/prof_$name
}
pmc_step() {    # <tag> <workload>: counters of full steps
  for grp in "${GROUPS_[@]}"; do
    PMC_TIMEOUT=400 bash $R/tools/pmc_run.sh "r06_$1_$(echo $grp | cut -d' ' -f1)" "$grp" bench.py --workload $2 --steps 5 --warmup 2 --spinup-time 0 --min-time 0 --no-cpu-baseline --no-lazy-top | grep -v "^$" | tail -4
    find $R/gpurun_out/pmc/r06_$1_$(echo $grp | cut -d' ' -f1) -name "*agent_info.csv" -delete
  done
}
for part in $parts; do
  case $part in
    c2) python3 $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
        prof bench $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-lazy-top ;;
    c2pmc) pmc_step c2 whole_graph; python3 $R/tools/pmc_pack_r06.py | tail -5 ;;
    c3) python3 $R/bench.py --workload whole_graph_pathway --no-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err; echo "bench config3 rc=$?"
        prof config3 $R/bench.py --workload whole_graph_pathway --steps 20 --warmup 5 --no-cpu-baseline --no-lazy-top ;;
    c3pmc) pmc_step c3 whole_graph_pathway; python3 $R/tools/pmc_pack_r06.py | tail -7 ;;
    c5) python3 $R/bench.py --workload rmat:10000000:200000000 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_rmat_10M_200M.json 2> $O/bench_rmat.err; echo "bench rmat rc=$?" ;;
    c5pmc) pmc_step c5 rmat:10000000:200000000
           for grp in "${GROUPS_[@]}"; do
             t=$(echo $grp | cut -d' ' -f1)
             PMC_TIMEOUT=400 bash $R/tools/pmc_run.sh "r06_rmat_base_$t" "$grp" tools/spmm_two_pass.py 10000000 200000000 128 base 3 | grep -v "^$" | tail -3
             PMC_TIMEOUT=400 bash $R/tools/pmc_run.sh "r06_rmat_fwd1_$t" "$grp" tools/spmm_two_pass.py 10000000 200000000 128 fwd1 3 | grep -v "^$" | tail -3
           done
           python3 $R/tools/pmc_pack_r06.py | tail -6 ;;
  esac
done
cp $R/profiles/r06_spmm_pmc.json $O/ 2>/dev/null
