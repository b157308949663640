#!/usr/bin/env python3
"""profiles/r06_scaling_forecast_*.json (tools/forecast_r06.sh -> tools/scaling_forecast.py) -> the markdown table of DESIGN.md section 8.
usage: forecast_table.py "<label>::<file>" ..."""
import json
import sys

print("| config | step | world | slowest rank's kernels, ms | slowest / mean | collectives | forecast ms per step: collectives at the measured world-1 floor | ... at 25 µs each | vs world 1 (floor / 25 µs) |")
print("|---|---|---|---|---|---|---|---|---|")
for arg in sys.argv[1:]:
    label, path = arg.split("::", 1)
    z = json.load(open(path))
    base = z["worlds"].get("1")
    for kind in ("full", "lazy"):
        for w, e in z["worlds"].items():
            k = e[kind]
            sp = sp25 = ""
            if base:
                sp = f"{base[kind]['forecast_ms_per_step_no_overlap'] / k['forecast_ms_per_step_no_overlap']:.2f} ×"
                sp25 = f"{base[kind]['forecast_ms_per_step_at_25us_per_collective'] / k['forecast_ms_per_step_at_25us_per_collective']:.2f} ×"
            print(f"| {label} | {kind} | {w} | {k['kernel_ms_max']:.3f} | {k['imbalance_max_over_mean']:.2f} | {k['collectives_enqueued']} | "
                  f"**{k['forecast_ms_per_step_no_overlap']:.3f}** | {k['forecast_ms_per_step_at_25us_per_collective']:.3f} | {sp} / {sp25} |")
