#!/bin/bash
# usage: scripts/gr.sh <logname> <timeout> '<command>'   -- runs gpurun, retrying while no GPU slot is free (exit code 3: nothing charged)
log=gpurun_out/$1.log; to=$2; shift 2
mkdir -p gpurun_out
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then break; fi
  sleep 45
done
echo "gr.sh done rc=$rc" >> $log
