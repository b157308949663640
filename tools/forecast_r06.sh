#!/bin/bash
# Round 6: the scaling FORECAST (never a measurement) with this round's kernels, the RCCL job's round-6 defaults (the lazy step's subset
# exchange is opt-in over RCCL: --lazy-halo 1 forecasts it) and the per-collective latency measured on a world-1 RCCL communicator
# (profiles/r06_rccl_world1_latency.json) beside the 25 us rounds 4-5 assumed.  usage: forecast_r06.sh [parts: c2 c4 c5 c5lazy c5trainer c5trainerlazy]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
for part in ${@:-c2 c4 c5 c5lazy c5trainer c5trainerlazy}; do
  case $part in
    c2) args="whole_graph 1,2,4,8" ;;
    c4) args="whole_graph_pathway 1,2,4,8 --d 256 --layers 3" ;;
    c5) args="rmat:10000000:200000000 1,2,4,8" ;;
    c5lazy) args="rmat:10000000:200000000 1,2,4,8 --lazy-halo 1" ;;
    c5trainer) args="rmat:10000000:200000000 1,2,4,8 --cache-layer1 1" ;;
    c5trainerlazy) args="rmat:10000000:200000000 1,2,4,8 --cache-layer1 1 --lazy-halo 1" ;;
  esac
  echo "== forecast $part: $args"
  timeout -k 10 900 python3 $R/tools/scaling_forecast.py $args > $O/scaling_forecast_$part.json 2> $O/scaling_forecast_$part.err
  echo "rc=$?"
  python3 - $O/scaling_forecast_$part.json <<'PY'
import json, sys
try:
    z = json.load(open(sys.argv[1]))
except Exception as e:
    print("no result:", e); raise SystemExit(0)
for w, e in z["worlds"].items():
    for kind in ("full", "lazy"):
        k = e[kind]
        print(f"world {w} {kind}: slowest rank {k['kernel_ms_max']:.3f} ms (max/mean {k['imbalance_max_over_mean']}), {k['collectives_enqueued']} collectives, "
              f"forecast {k['forecast_ms_per_step_no_overlap']:.3f} ms at the measured floor / {k['forecast_ms_per_step_at_25us_per_collective']:.3f} at 25 us each"
              + (f", x{k['forecast_speedup_vs_world1']}" if 'forecast_speedup_vs_world1' in k else ""))
PY
done
