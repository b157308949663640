#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4l; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -8 $O/pytest.txt; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
for v in 1 0 1 0; do
python3 bench.py --no-cpu-baseline --set prep_side=$v > $O/bench_side$v.json 2> $O/bench_side$v.err; echo "bench side$v rc=$?"
python3 - <<PY
import json
z=json.loads(open("$O/bench_side$v.json").read().strip().splitlines()[-1])
print("prep_side=$v", "ms/step", round(z["ms_per_step"],4), "long", round(z["long_run"]["ms_per_step"],4), "lazy", round(z["lazy_top"]["ms_per_step"],4), round(z["lazy_top"]["ms_per_step_with_layer1_kept"],4), "launches", z.get("launches_per_step"), "loss", z["config"]["final_loss"])
PY
done
