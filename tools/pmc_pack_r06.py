#!/usr/bin/env python3
"""gpurun_out/pmc/r06_* (tools/profile_r06.sh) -> profiles/r06_spmm_pmc.json: memory-side counters PER LAUNCH of every SpMM mode of a full
training step at BASELINE configs 2, 3 and 5, corrected as MI355X_MICROARCH.md's HBM section prescribes
(FETCH_SIZE x 2 on gfx950: 128-B requests tallied at 64 B; WRITE_SIZE exact; both in KB), with the sha256 of the spmm.hip that ran.

Passes: r06_<cfg>_<group> = rocprofv3 --pmc <group> over `bench.py --workload <cfg's workload> --steps 5 ...` (one counter group per pass);
launches are told apart by the kernel's first template argument (spmm.hip SpmmMode: 0 PLAIN, 1 FWD1, 2 BWD1, 3 BWD2, 4 BWD1S, 5 BWD2S).
Config 5's dense products have giant rows and run as three launches (a PLAIN chunk pass, the product, a finish pass): their counters come
from r06_rmat_{base,fwd1}_<group> = tools/spmm_two_pass.py base / fwd1 with 3 products, summed over the launches and divided by 3; its two
sparsity-aware backward hops (modes 4, 5: one launch each) come from the step run by name."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PMC = os.path.join(ROOT, "gpurun_out", "pmc")
MODES = {0: "plain", 1: "fwd1", 2: "bwd1", 3: "bwd2", 4: "bwd1s", 5: "bwd2s"}
CFG = {"config2": ("c2", 29960, 988028, 128, 2048), "config3": ("c3", 29960, 988676, 256, 2048), "config5": ("c5", 10000000, 210000000, 128, 2048)}
REPS_RMAT = 3


def mode_of(name):
    m = re.search(r"spmm_balanced(?:_list)?_kernel<(\d+),", name)      # (the list form of a row-filtered product: same mode numbers)
    return int(m.group(1)) if m else None


def newest(files):
    """gpurun merges every job's files into the same local directories (the csv names carry the profiled pid): of several passes only the latest"""
    return [max(files, key=os.path.getmtime)] if files else []


def read(tag):
    """{mode: {counter: [values per launch]}}, {mode: [durations us]} over every pass r06_<tag>_*"""
    acc, dur = collections.defaultdict(lambda: collections.defaultdict(list)), collections.defaultdict(list)
    for dd in sorted(glob.glob(os.path.join(PMC, f"r06_{tag}_*"))):
        for f in newest(glob.glob(dd + "/**/*counter_collection.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                m = mode_of(r["Kernel_Name"])
                if m is not None:
                    acc[m][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if dd.endswith("FETCH_SIZE"):
            for f in newest(glob.glob(dd + "/**/*kernel_trace.csv", recursive=True)):
                for r in csv.DictReader(open(f)):
                    m = mode_of(r["Kernel_Name"])
                    if m is not None:
                        dur[m].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc, dur


def case(c, us, alg, what):
    fetch, write = c["FETCH_SIZE"] * 1024 * 2, c.get("WRITE_SIZE", 0.0) * 1024
    hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    out = {"what": what, "counters_per_launch": c, "fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes_per_launch": fetch + write,
           "l2_hit_rate": hit / (hit + miss) if hit + miss else None, "avg_launch_us_profiled": us,
           "traffic_TBps": (fetch + write) / us / 1e6 if us else None}
    if alg:
        out.update({"alg_bytes_per_launch": alg, "traffic_over_alg": (fetch + write) / alg, "alg_TBps": alg / us / 1e6 if us else None,
                    "alg_frac_of_8TBps": alg / us / 1e6 / 8.0 if us else None})
    return out


def alg_bytes(mode, n, nnz, d, b):
    base = 8 * nnz + 4 * (n + 1)
    return {"plain": base + 8 * n * d, "fwd1": base + 12 * n * d, "bwd1": base + 24 * n * d, "bwd2": base + 16 * n * d,
            "bwd2s": base + 16 * n * d + 4 * b * d, "bwd1s": None}[mode]       # bwd1s: bench.py counts its hits / live rows per batch


import gcn_drug_repurposing_amd as pkg  # noqa: E402
lib = pkg.load()
hashes = {k: (lib.gss_source_hash(k.encode()) or b"").decode() for k in ("spmm.hip", "segments.h", "common.h", "plan.hip", "*")}
out = {"note": "rocprofv3 --pmc, one counter group per pass with --kernel-trace only (tools/pmc_run.sh); per launch = mean over the launches of a mode in "
               "the profiled steps (bench.py --steps 5: full training steps).  FETCH_SIZE / WRITE_SIZE are the L2s' memory-side (fabric) requests: "
               "Infinity-Cache hits are counted, so 'traffic' is what leaves the L2s, not DRAM bytes.  Durations are under the profiler.",
       "correction": "MI355X_MICROARCH.md section HBM: FETCH_SIZE x 2 (128-B requests tallied at 64 B on gfx950), WRITE_SIZE exact, both in KB",
       "source_hash": hashes}
# configurations whose passes are not under gpurun_out/pmc here (collected in another job) are kept from the existing file -- if, and only
# if, it was taken with the same spmm.hip
OUT_PATH = os.path.join(ROOT, "profiles", "r06_spmm_pmc.json")
if os.path.exists(OUT_PATH):
    old = json.load(open(OUT_PATH))
    if old.get("source_hash", {}).get("spmm.hip") == hashes["spmm.hip"]:
        for cfg in CFG:
            if cfg in old:
                out[cfg] = old[cfg]
for cfg, (tag, n, nnz, d, b) in CFG.items():
    acc, dur = read(tag)
    res = {}
    for m, cs in sorted(acc.items()):
        name = MODES[m]
        if "FETCH_SIZE" not in cs:
            continue
        if cfg == "config5" and m in (0, 1, 2, 3):
            continue            # dense products with giant rows: three launches each, taken per product below
        c = {k: sum(v) / len(v) for k, v in cs.items()}
        us = sum(dur[m]) / len(dur[m]) if dur[m] else None
        res[name] = case(c, us, alg_bytes(name, n, nnz, d, b), f"{cfg}: spmm_balanced_kernel<{name.upper()}> inside full training steps, N = {n}, nnz = {nnz}, d = {d}")
        res[name]["launches_profiled"] = len(cs["FETCH_SIZE"])
    if cfg == "config5":
        for tag2, name in (("rmat_base", "plain"), ("rmat_fwd1", "fwd1")):
            acc2, dur2 = read(tag2)
            tot, us = collections.defaultdict(float), 0.0
            for m, cs in acc2.items():
                for k, v in cs.items():
                    tot[k] += sum(v)
                us += sum(dur2[m])
            if "FETCH_SIZE" not in tot:
                continue
            c = {k: v / REPS_RMAT for k, v in tot.items()}
            res[name] = case(c, us / REPS_RMAT, alg_bytes(name, n, nnz, d, b),
                             f"{cfg}: one product ({name}) = chunk pass + product + finish pass of the giant rows, summed (tools/spmm_two_pass.py), N = {n}, nnz = {nnz}, d = {d}")
    if res:
        out[cfg] = {**out.get(cfg, {}), **res}
json.dump(out, open(OUT_PATH, "w"), indent=1)
for cfg in CFG:
    for name, c in out.get(cfg, {}).items():
        ratio = f"{c['traffic_over_alg']:.2f} x alg" if c.get("traffic_over_alg") else "(alg bytes per batch: bench.py)"
        print(f"{cfg} {name:6s} traffic {c['traffic_bytes_per_launch'] / 1e6:10.1f} MB = {ratio}, L2 hit {c['l2_hit_rate']:.3f}, {c['avg_launch_us_profiled']:.1f} us, "
              f"{c['traffic_TBps']:.2f} TB/s")
print("spmm.hip", hashes["spmm.hip"][:16])
