#!/bin/bash
# container side: gpurun with retries while no GPU slot / box is free (exit code 3: nothing charged)
# usage: tools/gpu.sh TIMEOUT_SECONDS 'command'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
