#!/bin/bash
# round 4: config 5 (RMAT 10M / 200M) as 8 shards on one GPU -- what a step moves with and without halo_recompute, and what hub delegation would move
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4i; mkdir -p $O; cd $R
timeout -k 10 500 python3 tools/delegate_volume.py 10000000 200000000 8 > $O/delegate_volume_rmat10m_world8.json 2> $O/delegate.err; echo "delegate rc=$?"
python3 -c "
import json; z=json.load(open('$O/delegate_volume_rmat10m_world8.json'))
for k in ['today']+[k for k in z if k.startswith('H')]: print(k, z[k])"
timeout -k 10 600 python3 tools/shard_emulation.py 10000000 200000000 8 3 128 auto -1 -1 > $O/shard_emulation_rmat10m_world8_recompute.json 2> $O/emu1.err; echo "emulation recompute rc=$?"
timeout -k 10 600 python3 tools/shard_emulation.py 10000000 200000000 8 3 128 auto -1 0 > $O/shard_emulation_rmat10m_world8_norecompute.json 2> $O/emu0.err; echo "emulation no-recompute rc=$?"
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4i/"
for f in ("shard_emulation_rmat10m_world8_recompute.json","shard_emulation_rmat10m_world8_norecompute.json"):
    try:
        z=json.load(open(O+f))
        print(f, "gpu_peak_gb", z["gpu_peak_gb"])
        for o in z["ranks"]:
            print("  rank", o["rank"], "rows", o["rows"], "plan_gb", o["plan_gb"], "full MB", o["full_step_mb_received"], "lazy MB", o["lazy_step_mb_received"], "x1", o["x1_mb_fetched_per_step"], "ms", o["ms_per_step"], o["lazy_ms_per_step"], "loss", o["loss"], o["lazy_loss"], o["collectives_per_full_step"], o["collectives_per_lazy_step"])
    except Exception as e: print(f, "ERR", e)
PY
