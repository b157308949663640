#!/bin/bash
# Round 5: the evidence behind bench.py's lines, collected on the GPU box (run from the repo root).  Arguments: which parts to run
# (default: all) out of  c2 (config 2: bench line, rocprofv3 kernel stats of full / lazy steps)  c2pmc (memory-side counters of its two
# forward SpMMs)  c3 (config 3 = whole_graph_pathway, d = 256, L = 3: bench line + kernel stats of full steps)  c3pmc (its counters)
#   -> gpurun_out/r05/*;  the counters -> gpurun_out/pmc/r05_* -> tools/pmc_pack_r02.py r05 -> profiles/r05_spmm_pmc.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r05; mkdir -p $O
parts=${@:-c2 c2pmc c3 c3pmc}
prof() {   # <name> <program and args...>: rocprofv3 --kernel-trace --stats, the kernel_stats csv -> $O/<name>_kernel_stats.csv
  local name=$1; shift
  ( cd /tmp; export TMPDIR=/tmp; rm -rf $O/prof_$name
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 "$@" > $O/${name}_under_rocprof.txt 2> $O/${name}_rocprof.err; echo "rocprof $name rc=$?" )
  cp $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv && rm -rf $O/prof_$name
}
pmc() {    # <case tag> <workload> <mode> <d>
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    bash $R/tools/pmc_run.sh "r05_$1_$(echo $grp | cut -d' ' -f1)" "$grp" tools/spmm_prof.py 2 $4 5 $2 $3 | grep -v "^$"
  done
}
for part in $parts; do
  case $part in
    c2) python3 $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
        prof bench $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-lazy-top
        prof lazy $R/tools/lazy_only_prof.py ;;
    c2pmc) pmc wg_fwd1 whole_graph fwd1 128; pmc wg_plain whole_graph plain 128 ;;
    c3) python3 $R/bench.py --workload whole_graph_pathway --no-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err; echo "bench config3 rc=$?"
        prof config3 $R/bench.py --workload whole_graph_pathway --steps 20 --warmup 5 --no-cpu-baseline --no-lazy-top ;;
    c3pmc) pmc wgp_fwd1 whole_graph_pathway fwd1 256; pmc wgp_plain whole_graph_pathway plain 256 ;;
  esac
done
python3 $R/tools/pmc_pack_r02.py r05 && cp $R/profiles/r05_spmm_pmc.json $O/
