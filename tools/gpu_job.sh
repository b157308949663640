#!/bin/bash
# One GPU-box job = a list of named steps; each step's output goes to gpurun_out/<tag>/<name>.txt; the job stops at the first failing step
# (no GPU step is started after one that failed or timed out) and exits with its code.
# usage: tools/gpu_job.sh <tag> "<name>|<timeout s>|<command>" ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; shift
O=$R/gpurun_out/$tag; mkdir -p $O
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; lim=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${lim}s): $cmd"
  ( cd $R && timeout -k 10 $lim bash -c "$cmd" ) > $O/$name.txt 2>&1
  rc=$?
  tail -4 $O/$name.txt
  if [ $rc -ne 0 ]; then echo "== $name FAILED rc=$rc"; exit $rc; fi
done
echo "== all steps ok"
