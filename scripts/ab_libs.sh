#!/bin/bash
# Same-box A/B of two builds of libv2a_cfm.so (ab_libs/old.so, ab_libs/new.so, built here and shipped with the snapshot): the library file
# is swapped between bench runs on the GPU box's scratch copy of the repo.  usage: scripts/ab_libs.sh [rounds] [bench flags...]
set -e
cd "$(dirname "$0")/.."
N=${1:-2}; shift || true
LIB=video-to-audio-and-piano-rp_amd/libv2a_cfm.so
mkdir -p gpurun_out
for i in $(seq 1 $N); do for v in old new; do
  cp ab_libs/$v.so $LIB
  timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder "$@" > gpurun_out/abl_x.log 2>&1
  echo "--- lib [$v] $*: $(grep -o '"value": [0-9.]*' gpurun_out/abl_x.log | head -1) $(grep -o '"mel_frames_per_s": [0-9.]*' gpurun_out/abl_x.log | head -1)"
done; done
cp ab_libs/new.so $LIB
