#!/bin/bash
# One GPU-box job = a list of named steps; each step's output goes to gpurun_out/<tag>/<name>.txt.  A step that times out or dies by a
# signal ends the job at once (no GPU step is started after one that was killed); an ordinary failure (rc 1: a red test) is reported and
# the remaining steps still run.  Exit code: the first non-zero one.
# usage: tools/gpu_job.sh <tag> "<name>|<timeout s>|<command>" ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; shift
O=$R/gpurun_out/$tag; mkdir -p $O
first=0
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; lim=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${lim}s): $cmd"
  ( cd $R && timeout -k 10 $lim bash -c "$cmd" ) > $O/$name.txt 2>&1
  rc=$?
  tail -4 $O/$name.txt
  if [ $rc -ne 0 ]; then
    echo "== $name FAILED rc=$rc"
    [ $first -eq 0 ] && first=$rc
    if [ $rc -ge 124 ]; then echo "== killed or timed out: stopping here"; exit $rc; fi
  fi
done
[ $first -eq 0 ] && echo "== all steps ok"
exit $first
