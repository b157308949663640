#!/bin/bash
# gpurun with a retry ONLY for exit code 3 (no box or slot free right now: nothing was charged, nothing ran).  Any other outcome -- a red test, a
# timeout, a refused call -- is returned at once: a GPU step that failed or was killed is never retried blindly.
# usage: tools/gpurun_retry.sh <log file> <gpurun args...>
log=$1; shift
for attempt in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun "$@" > "$log" 2>&1
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
