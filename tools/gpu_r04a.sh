#!/bin/bash
# round 4, first GPU call: the loss kernel with the gather prologue / finish tail -- parity, then speed
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4a; mkdir -p $O; cd $R
rm -f $O/parity_measured.jsonl
export GSS_RECORD_PARITY=$O/parity_measured.jsonl
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -5 $O/pytest.txt; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
unset GSS_RECORD_PARITY
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 bench.py --no-cpu-baseline --set loss_idx=0 --set loss_tail=0 > $O/bench_unfused.json 2> $O/bench_unfused.err; echo "bench unfused rc=$?"
python3 bench.py --no-cpu-baseline --set loss_idx=0 > $O/bench_tail_only.json 2>> $O/bench_unfused.err; echo "bench tail-only rc=$?"
python3 tools/lazy_step_prof.py > $O/lazy_step_classes.txt 2>&1
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4a/"
for f in ("bench.json","bench_unfused.json","bench_tail_only.json"):
    try:
        z=json.loads(open(O+f).read().strip().splitlines()[-1])
        print(f, "ms/step", round(z["ms_per_step"],4), "long", round(z["long_run"]["ms_per_step"],4), "lazy", z.get("lazy_top",{}).get("ms_per_step"), z.get("lazy_top",{}).get("ms_per_step_with_layer1_kept"), "launches", z.get("launches_per_step"), "loss", z["config"]["final_loss"])
        print("   kernel_us", {k:(round(v,1) if v else v) for k,v in z["kernel_us"].items()})
    except Exception as e: print(f, "ERR", e)
PY
