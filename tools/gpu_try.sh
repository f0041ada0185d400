#!/bin/bash
# gpurun wrapper: retries ONLY while no box is free (exit 3: nothing ran, nothing charged).
# usage: tools/gpu_try.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
