#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r4j; mkdir -p $O; cd $R
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.txt
python3 tools/trainer_walltime.py > $O/trainer_walltime.txt 2>&1; echo "walltime rc=$?"; tail -12 $O/trainer_walltime.txt
python3 bench.py --workload whole_graph_pathway --no-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err; echo "config3 rc=$?"
python3 bench.py --workload whole_graph_knn --no-cpu-baseline > $O/bench_knn.json 2> $O/bench_knn.err; echo "knn rc=$?"
python3 bench.py --workload rmat:1000000:20000000 --steps 10 --warmup 3 > $O/bench_rmat1m.json 2> $O/bench_rmat1m.err; echo "rmat1m rc=$?"
GSS_FORCE_SHARDED=1 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_sharded_world1.json 2> $O/bench_sharded_world1.err; echo "sharded world1 rc=$?"
python3 bench.py --workload diffusion --steps 3 --warmup 1 > $O/bench_diffusion.json 2> $O/bench_diffusion.err; echo "diffusion rc=$?"
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4j/"
for f in ("bench_config3.json","bench_knn.json","bench_rmat1m.json","bench_sharded_world1.json","bench_diffusion.json"):
    try:
        z=json.loads(open(O+f).read().strip().splitlines()[-1])
        print(f, "ms/step", round(z["ms_per_step"],4), "value", z["value"], "lazy", (z.get("lazy_top") or {}).get("ms_per_step"), "roofline", (z.get("roofline") or {}).get("frac"), z.get("collectives_per_step",{}).get("total"))
    except Exception as e: print(f, "ERR", e)
PY
