#!/bin/bash
# A/B of the one-launch cross-attention (v2a_qproj_xattn) in the sampler, same box, alternating
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_full_shape_gpu.py -x -q -k "one_launch_cross" > gpurun_out/xattn_test.log 2>&1 || { tail -30 gpurun_out/xattn_test.log; exit 1; }
tail -3 gpurun_out/xattn_test.log
: > gpurun_out/xattn_ab.log
for rep in 1 2 3; do
  for f in "" "--no-fuse-xattn"; do
    echo "== rep $rep ${f:-fused}" >> gpurun_out/xattn_ab.log
    timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-roofline --no-cpu-baseline --no-parity-mode --no-configs $f 2>>gpurun_out/xattn_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> gpurun_out/xattn_ab.log
  done
done
cat gpurun_out/xattn_ab.log
