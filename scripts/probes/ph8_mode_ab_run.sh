#!/bin/bash
# sampler A/B of the 8-phase kernel's K-loop form: v2a_tuning.gemm_8phase 1 (four phases per K tile) vs 3 (two phases), alternating
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
: > gpurun_out/ph8_mode_ab.log
for rep in 1 2; do
  for m in 1 3; do
    echo "== rep $rep gemm_8phase=$m: one clip, then 8 clips" >> gpurun_out/ph8_mode_ab.log
    timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-roofline --no-cpu-baseline --no-parity-mode --no-configs --gemm-8phase $m 2>>gpurun_out/ph8_mode_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('batched', {}).get('mel_frames_per_s'))" >> gpurun_out/ph8_mode_ab.log
  done
done
cat gpurun_out/ph8_mode_ab.log
