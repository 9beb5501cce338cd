#!/bin/bash
# Runs the given steps on the GPU box one after another; a step that times out (rc 124/137) ends the
# run (no further GPU step after a hang).  Each step logs to gpurun_out/<name>.log.
# usage: scripts/gpu_ci.sh step1 step2 ...   with steps from: kernels sampler full smoke bench bench_fp32 prof pmc
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name: $*"
  timeout -k 10 $to "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "=== $name rc=$rc"; tail -n ${TAILN:-15} gpurun_out/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 99; fi
  return $rc
}
for s in "$@"; do
  case $s in
    kernels) run kernels 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x --tb=short ;;
    kernels_all) run kernels 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=line ;;
    sampler) run sampler 600 python -m pytest tests/test_sampler_gpu.py -q -m gpu -x --tb=short -s ;;
    sampler_all) run sampler 600 python -m pytest tests/test_sampler_gpu.py -q -m gpu --tb=short -s ;;
    full) run full 900 python -m pytest tests/test_full_shape_gpu.py -q -m gpu -x --tb=short -s ;;
    alltests) run alltests 1100 python -m pytest tests -q -m gpu -x --tb=short ;;
    smoke) run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) run bench 900 python bench.py --steps 3 --warmup 1 ;;
    bench_nocpu) run bench_nocpu 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ;;
    bench_shapes) run bench_shapes 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --shapes ;;
    bench8) run bench8 600 python bench.py --steps 2 --warmup 1 --clips-per-gpu 8 --no-cpu-baseline ;;
    tiles) for t in auto 0 1 2 3; do
             if [ $t = auto ]; then unset V2A_GEMM_TILE; else export V2A_GEMM_TILE=$t; fi
             TAILN=0 run tiles_$t 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --shapes
             echo "--- tile cfg $t"; grep -E "timed|gemm<" gpurun_out/tiles_$t.log | sed -E 's/\[bench [0-9.]+s\] //' | sort | head -40
           done; unset V2A_GEMM_TILE ;;
    dbg) for t in 0 1 2 4 6 7; do
             export V2A_GEMM_DBG=$t
             TAILN=0 run dbg_$t 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --shapes
             echo "--- dbg $t"; grep -E "timed|gemm<" gpurun_out/dbg_$t.log | sed -E 's/\[bench [0-9.]+s\] //' | sort | head -40
           done; unset V2A_GEMM_DBG ;;
    pmc) B="python bench.py --steps 1 --warmup 0 --cfm-steps 6 --no-cpu-baseline --no-roofline --no-graph --no-batched"
         i=0
         for ctrs in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
           i=$((i+1)); rm -rf /tmp/pmc$i
           run pmc$i 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc$i -- $B || true
           python scripts/pmc_summary.py /tmp/pmc$i gpurun_out/pmc${i}_summary.csv
         done ;;
    small) for t in 3 2 1 0; do
             export V2A_GEMM_SMALL=$t
             TAILN=0 run small_$t 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-batched
             echo "--- small cfg $t: $(grep -E 'timed' gpurun_out/small_$t.log)"
           done; unset V2A_GEMM_SMALL ;;
    bench2) V2A_BENCH_BACKEND=gloo run bench2 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 ;;
    auxw) for t in 0 2; do
             export V2A_GEMM_AUXW=$t
             TAILN=0 run auxw_$t 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --shapes
             echo "--- auxw $t: $(grep -E 'timed' gpurun_out/auxw_$t.log)"; grep -E "geglu,bf16> 1564x10240|resid,f32> 1564x1280x5120|store,bf16> 1564x3088x1280" gpurun_out/auxw_$t.log
           done; unset V2A_GEMM_AUXW ;;
    batch) for b in 2 4 8; do
             TAILN=0 run batch_$b 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --clips-per-gpu $b
             echo "--- B=$b: $(grep -E 'timed' gpurun_out/batch_$b.log) $(grep -o '"value": [0-9.]*' gpurun_out/batch_$b.log)"
           done ;;
    shapes8) run shapes8 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --shapes --single-stream --clips-per-gpu 8 ;;
    prio) for t in 0 8; do
             export V2A_GEMM_DBG=$t
             TAILN=0 run prio_$t 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --shapes
             echo "--- prio $t: $(grep -E 'timed' gpurun_out/prio_$t.log)"; grep -E "geglu,bf16> 1564x10240|resid,f32> 1564x1280x5120|store,bf16> 1564x3088x1280" gpurun_out/prio_$t.log
             TAILN=0 run prio8_$t 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --clips-per-gpu 8
             echo "--- prio $t B=8: $(grep -E 'timed' gpurun_out/prio8_$t.log)"
           done; unset V2A_GEMM_DBG ;;
    prof) rm -rf /tmp/prof; run prof 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-batched
          mkdir -p gpurun_out/prof; cp /tmp/prof/*/*_kernel_stats.csv gpurun_out/prof/kernel_stats_multistream.csv ;;
    prof1) rm -rf /tmp/prof1; run prof1 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --single-stream --no-batched
          mkdir -p gpurun_out/prof; cp /tmp/prof1/*/*_kernel_stats.csv gpurun_out/prof/kernel_stats_singlestream.csv ;;
    *) echo "unknown step $s" ;;
  esac
done
exit 0
