#!/bin/bash
# Build libv2a_cfm.so (the C-ABI library) for gfx950.  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=../libv2a_cfm.so
BUILD=build
if [ "$1" = "--probe" ]; then   # instrumented K loops for scripts/probes/kloop_probe.py: separate objects, separate library
  shift; OUT=../libv2a_cfm_probe.so; BUILD=build_probe; set -- -DV2A_GEMM_PROBE "$@"
fi
# -save-temps=obj: the gfx950 assembly of every object stays in $BUILD/<file>-hip-amdgcn-amd-amdhsa-gfx950.s -- tests/test_isa_guard.py reads
# it (packed-FMA forms that must not come back, register spills); the other intermediates are removed below
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -save-temps=obj $@"
mkdir -p $BUILD
pids=()
for f in gemm gemm_8phase rowops attention conv vocoder qproj_xattn; do
  if [ ! -f $BUILD/$f.o ] || [ $f.hip -nt $BUILD/$f.o ] || [ v2a_common.h -nt $BUILD/$f.o ] || [ gemm_common.h -nt $BUILD/$f.o ] || [ ../../include/v2a_cfm.h -nt $BUILD/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o $BUILD/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
rm -f $BUILD/*.hipi $BUILD/*.bc $BUILD/*.hipfb $BUILD/*.out $BUILD/*.resolution.txt $BUILD/*-host-*.s $BUILD/*-gfx950.o
hipcc --offload-arch=gfx950 -shared -fPIC $BUILD/gemm.o $BUILD/gemm_8phase.o $BUILD/rowops.o $BUILD/attention.o $BUILD/conv.o $BUILD/vocoder.o $BUILD/qproj_xattn.o -o $OUT
echo "built $(realpath $OUT)"
