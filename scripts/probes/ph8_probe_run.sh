#!/bin/bash
# builds the 8-phase probe variants (on the GPU box or here) and times them
set -e
cd "$(dirname "$0")/../.."
C=video-to-audio-and-piano-rp_amd/csrc
mkdir -p gpurun_out
for s in 0 1 3 4 5 10 15; do
  python scripts/probes/ph8_probe.py video-to-audio-and-piano-rp_amd/libv2a_probe8_s$s.so >> gpurun_out/ph8_probe.log 2>>gpurun_out/ph8_probe.err
done
cat gpurun_out/ph8_probe.log
