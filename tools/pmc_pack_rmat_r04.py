#!/usr/bin/env python3
"""gpurun_out/pmc/r04_rmat_{base,two}_* (tools/pmc_rmat_r04.sh) -> profiles/r04_spmm_pmc_rmat10m.json: memory-side counters PER PRODUCT
y = A_hat x (the single pass: one launch; the two-pass column split: hot launch + cold launch summed), corrected as
MI355X_MICROARCH.md's HBM section prescribes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE exact, both in KB)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC = os.path.join(ROOT, "gpurun_out", "pmc")
n, m, h, ns = (int(v) for v in sys.argv[1:5])
REPS = 3
ROUND = os.environ.get("PMC_ROUND", "r04")     # tools/pmc_rmat_r05.sh sets r05
nnz, d = m + n, 128
alg = 8 * nnz + 4 * (n + 1) + 8 * n * d
out = {"note": "rocprofv3 --pmc, one counter group per pass (tools/pmc_run.sh), tools/spmm_two_pass.py base / two: counters summed over the "
               "spmm_balanced_kernel launches of a run and divided by its %d products.  FETCH_SIZE / WRITE_SIZE are what leaves the L2s "
               "(Infinity-Cache hits included), not DRAM bytes.  Durations are under the profiler." % REPS,
       "correction": "MI355X_MICROARCH.md section HBM: FETCH_SIZE x 2 (128-B requests tallied at 64 B on gfx950), WRITE_SIZE exact, both in KB",
       "workload": f"RMAT {n} nodes / {m} edges (+{n} self loops), d = {d}, hub-first relabelled, as bench.py --workload rmat:{n}:{m} builds it",
       "alg_bytes_per_product": alg, "cases": {}, "hbm_traffic": {}}
for tag, what in (("base", "single pass (the product's kernel: H = 65,536 hubs declared hot, cold rows non-temporal, 2 time-separated slices)"),
                  ("fwd1", "single pass with the Hadamard epilogue (AX = A_hat x, M = AX (.) x): the bench line's `roofline` kernel; algorithmic bytes + 4 N d"),
                  ("two", f"two passes over a column split at H = {h}: hot entries with {ns} XCD-pinned slices of {d * 4 // ns} B, cold entries streaming")):
    acc, dur = collections.defaultdict(float), 0.0
    for dd in glob.glob(os.path.join(PMC, f"{ROUND}_rmat_{tag}_*")):
        for f in glob.glob(dd + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "spmm_balanced" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]] += float(r["Counter_Value"])
        if dd.endswith("FETCH_SIZE"):
            for f in glob.glob(dd + "/**/*kernel_trace.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    if "spmm_balanced" in r["Kernel_Name"]:
                        dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "FETCH_SIZE" not in acc:
        continue
    c = {k: v / REPS for k, v in acc.items()}
    fetch, write = c["FETCH_SIZE"] * 1024 * 2, c.get("WRITE_SIZE", 0.0) * 1024
    hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    us = dur / REPS
    alg_t = alg + (4 * n * d if tag == "fwd1" else 0)
    out["cases"][tag] = {"what": what, "alg_bytes": alg_t, "counters_per_product": c, "fetch_bytes_corrected": fetch, "write_bytes": write,
                         "traffic_bytes_per_product": fetch + write, "traffic_over_alg": (fetch + write) / alg_t,
                         "l2_hit_rate": hit / (hit + miss) if hit + miss else None, "us_per_product_profiled": us,
                         "traffic_TBps": (fetch + write) / us / 1e6 if us else None, "alg_TBps": alg_t / us / 1e6 if us else None,
                         "alg_frac_of_8TBps": alg_t / us / 1e6 / 8.0 if us else None}
if "base" in out["cases"]:
    out["hbm_traffic"]["plain"] = {"traffic_bytes_per_launch": out["cases"]["base"]["traffic_bytes_per_product"]}
if "fwd1" in out["cases"]:
    out["hbm_traffic"]["fwd1"] = {"traffic_bytes_per_launch": out["cases"]["fwd1"]["traffic_bytes_per_product"]}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{ROUND}_spmm_pmc_rmat10m.json" if n == 10000000 else f"{ROUND}_spmm_pmc_rmat_{n}.json"), "w"), indent=1)
for tag, c in out["cases"].items():
    print(f"{tag:5s} traffic {c['traffic_bytes_per_product'] / 1e9:8.3f} GB = {c['traffic_over_alg']:.2f} x alg, L2 hit {c['l2_hit_rate']}, "
          f"{c['us_per_product_profiled']:.1f} us, alg {c['alg_TBps']:.3f} TB/s = {c['alg_frac_of_8TBps']:.3f} of 8")
