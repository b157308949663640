#!/bin/bash
# round 4, final evidence pass 3: BASELINE config 5 on one GPU (bench line, PMC figures wired from profiles/r04_spmm_pmc_rmat10m.json) and the two-process rehearsal
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4ac; mkdir -p $O; cd $R
timeout -k 10 500 python3 bench.py --workload rmat:10000000:200000000 --steps 5 --warmup 2 --min-time 0 --spinup-time 0 > $O/bench_rmat10m.json 2> $O/bench_rmat10m.err; echo "bench rmat rc=$?"
GSS_COMM_BACKEND=host timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --min-time 0 --spinup-time 0 > $O/bench_rehearsal2.json 2> $O/bench_rehearsal2.err; echo "rehearsal rc=$?"
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4ac/"
try:
    z=json.loads(open(O+"bench_rmat10m.json").read().strip().splitlines()[-1])
    print("rmat10m ms/step", z["ms_per_step"], "lazy", z["lazy_top"]["ms_per_step"], "roofline", z["roofline"]["frac"], z["roofline"]["traffic"], "plain", z["roofline_plain"]["frac"])
    print("   kernel_us", z["kernel_us"]); print("   mfma", {k:v["frac"] for k,v in z.get("mfma",{}).items()})
except Exception as e: print("ERR", e)
z=json.loads(open(O+"bench_rehearsal2.json").read().strip().splitlines()[-1]); print("rehearsal", z["collectives_per_step"]["total"], z["config"]["final_loss"], z["ms_per_step"])
PY
