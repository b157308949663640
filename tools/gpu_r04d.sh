#!/bin/bash
# round 4 GPU call: whole -m gpu suite, finish + input gradient in one launch (A/B), the two-pass SpMM sweep at RMAT scale
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4d; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -8 $O/pytest.txt; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
for v in 1 0; do
python3 bench.py --no-cpu-baseline --set loss_dgrad=$v > $O/bench_dgrad$v.json 2> $O/bench_dgrad$v.err; echo "bench dgrad$v rc=$?"
done
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4d/"
for f in ("bench_dgrad1.json","bench_dgrad0.json"):
    try:
        z=json.loads(open(O+f).read().strip().splitlines()[-1])
        print(f, "ms/step", round(z["ms_per_step"],4), "long", (z.get("long_run") or {}).get("ms_per_step"), "lazy", z.get("lazy_top",{}).get("ms_per_step"), z.get("lazy_top",{}).get("ms_per_step_with_layer1_kept"), "launches", z.get("launches_per_step"), "loss", z["config"]["final_loss"])
        print("   kernel_us_raw", {k:round(v,1) for k,v in z["kernel_us_raw_event_bracket"].items()})
    except Exception as e: print(f, "ERR", e)
PY
timeout -k 10 300 python3 tools/spmm_two_pass.py 1000000 20000000 128 sweep > $O/two_pass_rmat1m.txt 2>&1; echo "two-pass 1M rc=$?"; cat $O/two_pass_rmat1m.txt
timeout -k 10 600 python3 tools/spmm_two_pass.py 10000000 200000000 128 sweep > $O/two_pass_rmat10m.txt 2>&1; echo "two-pass 10M rc=$?"; cat $O/two_pass_rmat10m.txt
