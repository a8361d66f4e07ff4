#!/bin/bash
# gpurun with retries while every GPU slot of the pool is busy (exit code 3: nothing was charged); usage as gpurun's own
for k in $(seq 1 30); do
  /usr/local/graft/bin/gpurun "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 40
done
exit 3
