#!/bin/bash
# round 4: memory-side counters of the SpMM at BASELINE config 5's size, single pass (plain and Hadamard-fused) vs the two-pass column
# split (tools/pmc_rmat_r04.sh), then the RMAT bench line with the measured traffic wired in
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4g; mkdir -p $O; cd $R
bash tools/pmc_rmat_r04.sh 10000000 200000000 65536 2 > $O/pmc_rmat.log 2>&1; echo "pmc_rmat rc=$?"; tail -8 $O/pmc_rmat.log
cp profiles/r04_spmm_pmc_rmat10m.json $O/ 2>/dev/null
# the raw counter / trace csvs of nine passes over 210 M-entry launches are large: keep the summaries (log.txt), drop the rest
find $R/gpurun_out/pmc -name "*.csv" -size +200k -delete 2>/dev/null
timeout -k 10 400 python3 bench.py --workload rmat:10000000:200000000 --steps 5 --warmup 2 --min-time 0 --spinup-time 0 > $O/bench_rmat10m.json 2> $O/bench_rmat10m.err; echo "bench rmat rc=$?"
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4g/"
try:
    z=json.loads(open(O+"bench_rmat10m.json").read().strip().splitlines()[-1])
    print("rmat10m ms/step", z["ms_per_step"], "roofline", z["roofline"]["frac"], z["roofline"]["traffic"], "plain", z["roofline_plain"]["frac"], z["roofline_plain"]["traffic"])
    print("   kernel_us", z["kernel_us"]); print("   mfma", {k:v["frac"] for k,v in z.get("mfma",{}).items()})
except Exception as e: print("ERR", e)
PY
