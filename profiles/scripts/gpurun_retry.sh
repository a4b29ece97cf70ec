#!/bin/bash
# gpurun_retry.sh <timeout> <log> <command...>: gpurun, retried while the pod has no free slot (exit code 3)
t=$1; log=$2; shift 2
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > $log 2>&1; rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" $log; then exit $rc; fi
  sleep 60
done
exit 3
