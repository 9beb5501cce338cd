#!/bin/bash
# stream-priority experiment: audio chain high priority vs default
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for p in "" "-1,0" "0,1" "-1,1"; do
  echo "== priority '$p'" >> gpurun_out/prio.log
  timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-roofline --no-cpu-baseline --no-parity-mode --no-configs ${p:+--stream-priority=$p} 2>>gpurun_out/prio.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> gpurun_out/prio.log
done
cat gpurun_out/prio.log
