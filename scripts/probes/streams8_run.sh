#!/bin/bash
# 8 clips per GPU: three streams vs one stream, alternating on one box
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
: > gpurun_out/streams8.log
for rep in 1 2; do
  for f in "" "--single-stream"; do
    echo "== rep $rep ${f:-three streams}" >> gpurun_out/streams8.log
    timeout -k 10 300 python bench.py --steps 2 --warmup 1 --clips-per-gpu 8 --no-roofline --no-cpu-baseline --no-parity-mode --no-configs --no-batched $f 2>>gpurun_out/streams8.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> gpurun_out/streams8.log
  done
done
cat gpurun_out/streams8.log
