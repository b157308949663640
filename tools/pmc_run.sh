#!/bin/bash
# usage: pmc_run.sh <tag> "<counters>" <program args...>   (one rocprofv3 --pmc pass with a hard timeout)
tag=$1; ctrs=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ -z "$R" ] || [ ! -f "$R/bench.py" ]; then echo "pmc_run.sh: cannot find the repo root (GRAFT_REPO_ROOT=$GRAFT_REPO_ROOT)"; exit 2; fi
O=$R/gpurun_out/pmc/$tag; rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
echo "== $tag [$ctrs]"
timeout -k 5 ${PMC_TIMEOUT:-300} rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O -- python3 $R/"$1" "${@:2}" > $O/log.txt 2>&1
echo "rc=$?"
python3 $R/tools/pmc_summary.py $O ${PMC_FILTER:-spmm}
# keep only what the pack scripts read: the rows of the filtered kernels (gpurun merges at most 64 MiB back; a pass over an RMAT build is tens of MB)
python3 - "$O" "${PMC_FILTER:-spmm}" <<'PY'
import csv, glob, os, sys
root, flt = sys.argv[1], sys.argv[2]
for f in glob.glob(root + "/**/*.csv", recursive=True):
    if f.endswith("counter_collection.csv") or f.endswith("kernel_trace.csv"):
        rows = list(csv.DictReader(open(f)))
        keep = [r for r in rows if flt in r.get("Kernel_Name", "")]
        if rows:
            with open(f, "w", newline="") as out:
                w = csv.DictWriter(out, fieldnames=list(rows[0].keys()))
                w.writeheader()
                w.writerows(keep)
    else:
        os.remove(f)
PY
