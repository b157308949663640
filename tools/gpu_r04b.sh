#!/bin/bash
# round 4, second GPU call: loss tail as one batch of loads; the sharded step with 4 collectives
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4b; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -15 $O/pytest.txt; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 bench.py --no-cpu-baseline --set loss_tail=0 > $O/bench_tail0.json 2> $O/bench_tail0.err; echo "bench tail0 rc=$?"
python3 bench.py --no-cpu-baseline --set loss_tail=2 > $O/bench_tail2.json 2>> $O/bench_tail0.err; echo "bench tail2 rc=$?"
python3 bench.py --no-cpu-baseline --set loss_idx=1 > $O/bench_idx1.json 2>> $O/bench_tail0.err; echo "bench idx1 rc=$?"
GSS_COMM_BACKEND=host timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --min-time 0 --spinup-time 0 > $O/bench_rehearsal2.json 2> $O/bench_rehearsal2.err; echo "rehearsal rc=$?"
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4b/"
for f in ("bench.json","bench_tail0.json","bench_tail2.json","bench_idx1.json","bench_rehearsal2.json"):
    try:
        z=json.loads(open(O+f).read().strip().splitlines()[-1])
        print(f, "ms/step", round(z["ms_per_step"],4), "long", (z.get("long_run") or {}).get("ms_per_step"), "lazy", z.get("lazy_top",{}).get("ms_per_step"), z.get("lazy_top",{}).get("ms_per_step_with_layer1_kept"), "launches", z.get("launches_per_step"), "loss", z["config"]["final_loss"])
        print("   kernel_us_raw", {k:round(v,1) for k,v in z["kernel_us_raw_event_bracket"].items()})
        if "collectives_per_step" in z: print("   collectives", z["collectives_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
