#!/bin/bash
# round 4, final: config 5 as 8 shards on one GPU with the round's kernels (identical losses across ranks and step kinds are asserted by the tool)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4aj; mkdir -p $O; cd $R
timeout -k 10 700 python3 tools/shard_emulation.py 10000000 200000000 8 3 128 auto -1 -1 > $O/shard_emulation_rmat10m_world8_recompute.json 2> $O/emu.err; echo "emulation rc=$?"; tail -3 $O/emu.err
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4aj/"
z=json.load(open(O+"shard_emulation_rmat10m_world8_recompute.json"))
for o in z["ranks"]:
    print("  rank", o["rank"], "rows", o["rows"], "plan_gb", o["plan_gb"], "full MB", o["full_step_mb_received"], "lazy MB", o["lazy_step_mb_received"], "ms", o["ms_per_step"], o["lazy_ms_per_step"], "loss", o["loss"], o["lazy_loss"])
PY
