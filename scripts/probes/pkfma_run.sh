#!/bin/bash
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/pkfma scripts/probes/pkfma_probe.hip
timeout -k 10 300 /tmp/pkfma 300 "python scripts/probes/concurrency_probe.py --load 100 matmul > /dev/null 2>&1 &" > gpurun_out/pkfma.log 2>&1
cat gpurun_out/pkfma.log
