#!/bin/bash
# Build libv2a_cfm.so (the C-ABI library) for gfx950.  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=../libv2a_cfm.so
BUILD=build
if [ "$1" = "--probe" ]; then   # instrumented K loops for scripts/probes/kloop_probe.py: separate objects, separate library
  shift; OUT=../libv2a_cfm_probe.so; BUILD=build_probe; set -- -DV2A_GEMM_PROBE "$@"
fi
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result $@"
mkdir -p $BUILD
pids=()
for f in gemm gemm_8phase rowops attention conv vocoder qproj_xattn; do
  if [ ! -f $BUILD/$f.o ] || [ $f.hip -nt $BUILD/$f.o ] || [ v2a_common.h -nt $BUILD/$f.o ] || [ gemm_common.h -nt $BUILD/$f.o ] || [ ../../include/v2a_cfm.h -nt $BUILD/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o $BUILD/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $BUILD/gemm.o $BUILD/gemm_8phase.o $BUILD/rowops.o $BUILD/attention.o $BUILD/conv.o $BUILD/vocoder.o $BUILD/qproj_xattn.o -o $OUT
echo "built $(realpath $OUT)"
