#!/bin/bash
# round 4: the sharded step with 3 collectives -- rehearsal line (two processes, host-staged) and config 5 as 8 shards on one GPU
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4k; mkdir -p $O; cd $R
GSS_COMM_BACKEND=host timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --min-time 0 --spinup-time 0 > $O/bench_rehearsal2.json 2> $O/bench_rehearsal2.err; echo "rehearsal rc=$?"
timeout -k 10 600 python3 tools/shard_emulation.py 10000000 200000000 8 3 128 auto -1 -1 > $O/shard_emulation_rmat10m_world8_recompute.json 2> $O/emu1.err; echo "emulation rc=$?"
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4k/"
z=json.loads(open(O+"bench_rehearsal2.json").read().strip().splitlines()[-1]); print("rehearsal", z["collectives_per_step"], z["config"]["final_loss"])
z=json.load(open(O+"shard_emulation_rmat10m_world8_recompute.json"))
for o in z["ranks"]:
    print("  rank", o["rank"], "rows", o["rows"], "plan_gb", o["plan_gb"], "full MB", o["full_step_mb_received"], "lazy MB", o["lazy_step_mb_received"], "ms", o["ms_per_step"], o["lazy_ms_per_step"], "loss", o["loss"], o["lazy_loss"], o["collectives_per_full_step"], o["collectives_per_lazy_step"])
PY
