#!/usr/bin/env python3
"""gpurun_out/pmc/<tag> -> profiles/<tag>_pmc.json (MFMA pipe utilisation of the matrix kernels of a step, under the profiler);
tag = argv[1] (default r02_mfma), workload description = argv[2]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02_mfma"
WHAT = sys.argv[2] if len(sys.argv) > 2 else "config 2"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(ROOT, f"gpurun_out/pmc/{TAG}/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_nt_lds" in k or "loss_fused" in k or "wgrad_tn" in k:
            acc[k.split("(")[0].replace("void gss::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(ROOT, f"gpurun_out/pmc/{TAG}/**/*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_nt_lds" in k or "loss_fused" in k or "wgrad_tn" in k:
            dur[k.split("(")[0].replace("void gss::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"note": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE (--kernel-trace only, "
               "tools/pmc_mfma_r02.sh) on bench.py --steps 5 --warmup 2: every v_mfma_f32_16x16x4_f32 kernel of a step at " + WHAT + ", mean per launch. "
               "GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA busy cycles sum the 1024 SIMDs (32 cycles per instruction). Profiled launches are slower "
               "than in bench.py (no spin-up, counters on).", "kernels": {}}
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    m["kernel_cycles_under_profiler"] = cyc
    m["mfma_pipe_utilisation_under_profiler"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc if cyc else None
    m["avg_launch_us_profiled"] = sum(dur[k]) / len(dur[k]) if dur[k] else None
    out["kernels"][k] = m
json.dump(out, open(os.path.join(ROOT, "profiles", TAG + "_pmc.json"), "w"), indent=1)
for k, m in out["kernels"].items():
    print(f"{k:42s} MFMA {m.get('SQ_INSTS_MFMA', 0):10.0f}  pipe busy {m['mfma_pipe_utilisation_under_profiler']:.3f}  {m['avg_launch_us_profiled']:.1f} us")
