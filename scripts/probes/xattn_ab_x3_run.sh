#!/bin/bash
# A/B of the one-launch cross-attention in bf16x3 mode, same box, alternating
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
true
true
: > gpurun_out/xattn_ab_x3.log
for rep in 1 2; do
  for f in "" "--no-fuse-xattn"; do
    echo "== rep $rep bf16x3 ${f:-one launch}" >> gpurun_out/xattn_ab_x3.log
    timeout -k 10 300 python bench.py --dtype bf16x3 --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-parity-mode --no-configs --no-batched --no-video2roll --no-vocoder $f 2>>gpurun_out/xattn_ab_x3.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> gpurun_out/xattn_ab_x3.log
  done
done
cat gpurun_out/xattn_ab_x3.log
