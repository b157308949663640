#!/bin/bash
# round 4: eight-wave 128-node projection tiles as the default at d = 128; config 3 (d = 256) with it forced (gemm_variant 5) against its default
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4af; mkdir -p $O; cd $R
for v in 5 2 5 2; do
python3 bench.py --workload whole_graph_pathway --no-cpu-baseline --set gemm_variant=$v > $O/bench_c3_v$v.json 2> $O/bench_c3_v$v.err; echo "bench rc=$?"
python3 - <<PY
import json
z=json.loads(open("$O/bench_c3_v$v.json").read().strip().splitlines()[-1])
print("config 3 gemm_variant=$v", "ms/step", round(z["ms_per_step"],4), "long", round(z["long_run"]["ms_per_step"],4), "lazy", round(z["lazy_top"]["ms_per_step"],4), {k: (round(x,1) if x else x) for k,x in z["kernel_us"].items()})
PY
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -4 $O/pytest.txt; echo "pytest rc=$rc"
