#!/usr/bin/env python3
import json, sys
o = json.load(open(sys.argv[1]))
print({k: (round(o[k], 5) if isinstance(o[k], float) else o[k]) for k in ("value", "ms_per_step", "epoch_time_s") if k in o}, o.get("ms_per_step_layer1_cached"))
if "roofline" in o:
    r = o["roofline"]; print("roofline: %.1f GB/s frac %.4f avg %.1f us" % (r["achieved"], r["frac"], r["avg_launch_us"]))
    print("us/launch", {k: round(v, 1) for k, v in o["kernel_us"].items()})
    print("ms/step  ", {k: round(v, 4) for k, v in o["kernel_ms_per_step"].items()}, "sum=%.4f" % sum(o["kernel_ms_per_step"].values()))
    print("dense_fwd mfma", {k: round(v, 3) if isinstance(v, float) else v for k, v in o.get("mfma_dense_fwd", {}).items()})
if "cpu_baseline" in o:
    print("cpu", o["cpu_baseline"]["s_per_step"], o["cpu_baseline"]["cores"])
