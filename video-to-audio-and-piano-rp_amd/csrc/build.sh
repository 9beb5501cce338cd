#!/bin/bash
# Build libv2a_cfm.so (the C-ABI library) for gfx950.  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=../libv2a_cfm.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result $@"
mkdir -p build
pids=()
for f in gemm gemm_8phase rowops attention conv vocoder; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ v2a_common.h -nt build/$f.o ] || [ gemm_common.h -nt build/$f.o ] || [ ../../include/v2a_cfm.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/gemm.o build/gemm_8phase.o build/rowops.o build/attention.o build/conv.o build/vocoder.o -o $OUT
echo "built $(realpath $OUT)"
