#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout_s> <logfile> <command...>   -- retries while the pod's GPU slots are busy (nothing is charged then)
t=$1; log=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@" > "$log" 2>&1
  if ! grep -q "status=transient" "$log"; then exit 0; fi
  sleep 60
done
