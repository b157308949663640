#!/usr/bin/env python3
"""gpurun_out/pmc/<round>_* (tools/pmc_spmm_r02.sh, pmc_spmm_r02b.sh; tools/profile_r03.sh; usage: pmc_pack_r02.py [round tag, default r02]) -> profiles/<round>_spmm_pmc.json: per SpMM launch the memory-side
counters, corrected as MI355X_MICROARCH.md section HBM prescribes (FETCH_SIZE counts the 128-B requests of wide reads at
64 B on gfx950 -> doubled; WRITE_SIZE exact; both in KB), next to the algorithmic bytes of SURVEY.md section 8(d)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC = os.path.join(ROOT, "gpurun_out", "pmc")
PREFIX = sys.argv[1] if len(sys.argv) > 1 else "r02"     # round tag of the passes (tools/pmc_spmm_r02*.sh, tools/profile_r03.sh)
CASES = {  # tag -> (description, n, nnz, d, extra operand rows)
    "wg_fwd1": ("whole_graph stand-in as bench.py runs it (nodes relabelled hub-first), spmm_balanced_kernel<FWD1> (AX = A_hat X, M = AX (.) X)", 29960, 988028, 128, 1),
    "wg_plain": ("whole_graph stand-in as bench.py runs it (nodes relabelled hub-first), spmm_balanced_kernel<PLAIN> (AM = A_hat M)", 29960, 988028, 128, 0),
    "wgp_fwd1": ("BASELINE config 3: whole_graph + pathway edges stand-in, d = 256, as bench.py --workload whole_graph_pathway runs it, spmm_balanced_kernel<FWD1>", 29960, 988676, 256, 1),
    "wgp_plain": ("BASELINE config 3: whole_graph + pathway edges stand-in, d = 256, spmm_balanced_kernel<PLAIN>", 29960, 988676, 256, 0),
    "wg_norelabel_fwd1": ("whole_graph stand-in in the loader's node order, FWD1", 29960, 988028, 128, 1),
    "wg_norelabel_plain": ("whole_graph stand-in in the loader's node order, PLAIN", 29960, 988028, 128, 0),
    "r1m_plain": ("RMAT 1M / 20M (+1M self loops), generator node order, PLAIN", 1000000, 21000000, 128, 0),
    "r1m_relabel": ("RMAT 1M / 20M, nodes relabelled hub-first + cold rows non-temporal, PLAIN", 1000000, 21000000, 128, 0),
    "r10m_plain": ("RMAT 10M / 200M (+10M self loops), generator node order, PLAIN", 10000000, 210000000, 128, 0),
    "r10m_relabel": ("RMAT 10M / 200M, nodes relabelled hub-first + cold rows non-temporal, PLAIN", 10000000, 210000000, 128, 0),
}


def counters(tag):
    acc, dur = collections.defaultdict(list), []
    for d in glob.glob(os.path.join(PMC, f"{PREFIX}_{tag}_*")):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "spmm_balanced" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "spmm_balanced" in r["Kernel_Name"]:
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: sum(v) / len(v) for k, v in acc.items()}, dur


out = {"note": "rocprofv3 --pmc, one counter group per pass with --kernel-trace only (tools/pmc_run.sh), tools/spmm_prof.py: per launch, mean over "
               "the launches of a pass.  FETCH_SIZE / WRITE_SIZE are L2 memory-side (fabric) requests: Infinity-Cache hits are counted, "
               "so 'traffic' is what leaves the L2s, not DRAM bytes.  Durations are under the profiler (a few % slower than unprofiled).",
       "correction": "MI355X_MICROARCH.md section HBM: FETCH_SIZE x 2 (128-B requests tallied at 64 B on gfx950), WRITE_SIZE exact, both in KB",
       "cases": {}, "hbm_traffic": {}}
for tag, (desc, n, nnz, d, extra) in CASES.items():
    c, dur = counters(tag)
    if "FETCH_SIZE" not in c:
        continue
    alg = 8 * nnz + 4 * (n + 1) + 8 * n * d + 4 * n * d * extra
    fetch, write = c["FETCH_SIZE"] * 1024 * 2, c.get("WRITE_SIZE", 0.0) * 1024
    hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    us = sum(dur) / len(dur) if dur else None
    case = {"what": desc, "counters": c, "fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes_per_launch": fetch + write,
            "alg_bytes_per_launch": alg, "traffic_over_alg": (fetch + write) / alg, "gather_bytes": nnz * d * 4,
            "l2_hit_rate": hit / (hit + miss) if hit + miss else None, "avg_launch_us_profiled": us,
            "traffic_TBps": (fetch + write) / us / 1e6 if us else None, "alg_TBps": alg / us / 1e6 if us else None}
    out["cases"][tag] = case
    if tag == "wg_fwd1":
        out["hbm_traffic"]["fwd1"] = {"traffic_bytes_per_launch": fetch + write}
    if tag == "wg_plain":
        out["hbm_traffic"]["plain"] = {"traffic_bytes_per_launch": fetch + write}
    if tag in ("wgp_fwd1", "wgp_plain"):     # config 3 (d = 256): bench.py --workload whole_graph_pathway reads these
        out.setdefault("hbm_traffic_config3", {})[tag[4:]] = {"traffic_bytes_per_launch": fetch + write}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{PREFIX}_spmm_pmc.json"), "w"), indent=1)
for tag, c in out["cases"].items():
    print(f"{tag:14s} traffic {c['traffic_bytes_per_launch'] / 1e9:8.3f} GB = {c['traffic_over_alg']:.2f} x alg, L2 hit {c['l2_hit_rate']:.3f}, "
          f"{c['avg_launch_us_profiled']:.1f} us, traffic {c['traffic_TBps']:.2f} TB/s, alg {c['alg_TBps']:.3f} TB/s")
