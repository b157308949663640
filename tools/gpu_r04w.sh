#!/bin/bash
# round 4: whole-line epilogues as the default: the GPU suite, then config 3 (d = 256, L = 3: forward, normalise-free, and N-row input gradients) with and without
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4w; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -5 $O/pytest.txt; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
for v in 1 0 1 0; do
python3 bench.py --workload whole_graph_pathway --no-cpu-baseline --set gemm_lines=$v > $O/bench_c3_lines$v.json 2> $O/bench_c3_lines$v.err; echo "bench rc=$?"
python3 - <<PY
import json
z=json.loads(open("$O/bench_c3_lines$v.json").read().strip().splitlines()[-1])
print("config 3 gemm_lines=$v", "ms/step", round(z["ms_per_step"],4), "long", round(z["long_run"]["ms_per_step"],4), "lazy", round(z["lazy_top"]["ms_per_step"],4), {k: (round(x,1) if x else x) for k,x in z["kernel_us"].items()})
PY
done
