#!/bin/bash
# Runs the given steps on the GPU box one after another; a step that times out (rc 124/137) ends the
# run (no further GPU step after a hang).  Each step logs to gpurun_out/<name>.log.
# usage: scripts/gpu_ci.sh step1 step2 ...   steps: kernels sampler full alltests smoke bench bench_nocpu bench_shapes bench2 batch shapes8 configs prof prof1 pmc
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
# DT=bf16x3 (or fp32): the profile / trace steps run that compute mode and tag their outputs with it
DTA=""; DTS=""; if [ -n "$DT" ]; then DTA="--dtype $DT"; DTS="_$DT"; fi
for s in "$@"; do
  case $s in
    trace) rm -rf /tmp/trace; TAILN=3 run trace$DTS 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace -- python bench.py $DTA ${TRACE_EXTRA:-} --steps 1 --warmup 1 --cfm-steps 8 --no-cpu-baseline --no-roofline --no-batched --no-parity-mode --no-configs --no-video2roll --no-vocoder
           python scripts/timeline_summary.py /tmp/trace > gpurun_out/timeline$DTS${TRACE_TAG:-}.txt 2>&1; tail -n ${TL_TAIL:-90} gpurun_out/timeline$DTS${TRACE_TAG:-}.txt ;;
    side) SW=${SIDE_SWEEP:-"-1:-1 0:-1 5:-1 6:-1 0:0 6:0"}
         for v in $SW; do
           st=${v%%:*}; mt=${v##*:}
           TAILN=0 run side_${st}_${mt} 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder --no-batched --side-tile $st --gemm-force-tile $mt
           echo "--- side tile $st main tile $mt: $(grep -o '"value": [0-9.]*' gpurun_out/side_${st}_${mt}.log)"
         done ;;
    ilv) for v in 0 1 0 1; do
           TAILN=0 run ilv_x 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder --no-batched --interleave-capture $v
           echo "--- interleave $v: $(grep -o '"value": [0-9.]*' gpurun_out/ilv_x.log)"
         done ;;
    xcd) for v in "" "--xcd-1x8" "" "--xcd-1x8"; do
           TAILN=0 run xcd_x 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder $v
           echo "--- xcd grid $v: $(grep -o '"value": [0-9.]*' gpurun_out/xcd_x.log) batched $(grep -o '"mel_frames_per_s": [0-9.]*' gpurun_out/xcd_x.log | head -1)"
         done ;;
    cross) for v in "" "--cross-on-sides"; do
           TAILN=0 run cross_x 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder --no-batched --side-tile 0 $v
           echo "--- side tile 0 $v: $(grep -o '"value": [0-9.]*' gpurun_out/cross_x.log)"
         done ;;
    rowbench) TAILN=10 run rowbench 300 python scripts/microbench_rowops.py ;;
    kdw) run kdw 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -k "dwconv or rmsnorm" ;;
    extev) TAILN=12 run extev 120 python scripts/probes/ext_event_probe.py ;;
    k8p) run k8p 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -k "8phase" ;;
    probe) TAILN=40 run probe 600 python scripts/gemm_probe.py ${PROBE_ARGS:-} ;;
    probe_geglu) TAILN=40 run probe_geglu 600 python scripts/gemm_probe.py --epi geglu 1564x8192x1024 1564x10240x1280 1564x4096x512 12512x8192x1024 ;;
    x3qkv) for v in ${X3_SWEEP:-"_" "a.qkv=5,t.qkv=5" "a.qkv=5,t.qkv=5,f.qkv=5" "_"}; do
           a=""; [ "$v" != "_" ] && a="--side-tiles $v"
           TAILN=0 run x3_x 300 python bench.py --dtype bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder --no-batched $a
           echo "--- bf16x3 split tiles [$v]: $(grep -o '"value": [0-9.]*' gpurun_out/x3_x.log | head -1)"
         done ;;
    soak) TAILN=12 run soak 1100 python scripts/probes/soak_probe.py ;;
    kernels) run kernels 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x --tb=short ;;
    kernels_all) run kernels 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=line ;;
    sampler) run sampler 600 python -m pytest tests/test_sampler_gpu.py -q -m gpu -x --tb=short -s ;;
    sampler_all) run sampler 600 python -m pytest tests/test_sampler_gpu.py -q -m gpu --tb=short -s ;;
    v2r) run v2r 600 python -m pytest tests/test_video2roll_gpu.py -q -m gpu -x --tb=short -s ;;
    enc) run enc 600 python -m pytest tests/test_encodec_gpu.py -q -m gpu -x --tb=short -s ;;
    full) TAILN=40 run full 900 python -m pytest tests/test_full_shape_gpu.py -q -m gpu --tb=short -s ;;
    alltests) run alltests 1100 python -m pytest tests -q -m gpu -x --tb=short ;;
    alltests_s) TAILN=60 run alltests_s 1100 python -m pytest tests -q -m gpu --tb=short -s ;;
    smoke) run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) TAILN=4 run bench 900 python bench.py --steps 5 --warmup 2 ;;
    ab8) for v in "0 0" "1 200" "1 80"; do set -- $v
           TAILN=0 run ab8_$1_$2 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder --gemm-8phase $1 --gemm-8phase-min-tiles $2
           echo "--- 8phase=$1 min_tiles=$2: $(grep -o '"value": [0-9.]*' gpurun_out/ab8_$1_$2.log) batched $(grep -o '"mel_frames_per_s": [0-9.]*' gpurun_out/ab8_$1_$2.log | head -1)"
         done ;;
    bench_nocpu) run bench_nocpu 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ;;
    bench_shapes) run bench_shapes 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --shapes ;;
    bench8) run bench8 600 python bench.py --steps 2 --warmup 1 --clips-per-gpu 8 --no-cpu-baseline ;;
    pmc) B="python bench.py $DTA --steps 1 --warmup 0 --cfm-steps 6 --no-cpu-baseline --no-roofline --no-graph --no-batched --no-video2roll --no-vocoder --no-parity-mode --no-configs ${PMC_EXTRA:-}"
         i=0
         for ctrs in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
           i=$((i+1)); rm -rf /tmp/pmc$i
           run pmc$i 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc$i -- $B || true
           python scripts/pmc_summary.py /tmp/pmc$i gpurun_out/pmc${i}${PMC_TAG:-}_summary.csv
         done ;;
    pmcdw) i=0
         for ctrs in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
           i=$((i+1)); rm -rf /tmp/pmcdw$i
           TAILN=3 run pmcdw$i 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmcdw$i -- python scripts/probes/dwconv_probe.py ${DW_ARGS:-16 782 1024 0} || true
           python scripts/pmc_summary.py /tmp/pmcdw$i gpurun_out/pmcdw${i}_summary.csv || true
         done ;;
    foldab) for v in "" "--no-fold-norm"; do
           TAILN=0 run fold_x 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder $v
           echo "--- fold [$v]: $(grep -o '"value": [0-9.]*' gpurun_out/fold_x.log | head -1) batched $(grep -o '"mel_frames_per_s": [0-9.]*' gpurun_out/fold_x.log | head -1)"
         done ;;
    foldtab) for v in fold nofold; do
           f=""; [ $v = nofold ] && f="--no-fold-norm"
           TAILN=0 run ft_$v 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-configs --no-video2roll --no-vocoder $f
         done ;;
    sidetiles) for v in ${ST_SWEEP:-"_" "t.qkv=1" "t.ff1=1" "f.cross=1,f.out=1,f.ff2=1" "f.cross=3,f.out=3,f.ff2=3" "_"}; do
           a=""; [ "$v" != "_" ] && a="--side-tiles $v"
           TAILN=0 run st_x 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder --no-batched $a ${ST_EXTRA:-}
           echo "--- side tiles [$v]: $(grep -o '"value": [0-9.]*' gpurun_out/st_x.log | head -1)"
         done ;;
    bigtiles) for v in ${BT_SWEEP:-"_" "a.qkv=0,t.qkv=0" "f.qkv=0" "a.ff2=0" "_"}; do
           a=""; [ "$v" != "_" ] && a="--big-tiles $v"
           TAILN=0 run bt_x 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder --no-batched --clips-per-gpu 8 $a
           echo "--- big tiles [$v]: $(grep -o '"value": [0-9.]*' gpurun_out/bt_x.log | head -1)"
         done ;;
    misc1) for v in "" "--interleave-capture 1" "" "--interleave-capture 1" "" "--interleave-capture 1"; do
           TAILN=0 run m1_x 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder $v
           echo "--- [$v]: $(grep -o '"value": [0-9.]*' gpurun_out/m1_x.log | head -1) batched $(grep -o '"mel_frames_per_s": [0-9.]*' gpurun_out/m1_x.log | head -1)"
         done ;;
    x3tiles) for v in ${X3T_SWEEP:-"_" "a.qkv=4,t.qkv=4" "a.qkv=6,t.qkv=6" "a.qkv=6,t.qkv=6,t.ff1=6" "a.qkv=6,t.qkv=6,t.ff1=3" "a.qkv=6,t.qkv=6,t.ff1=6,a.ff1=6,f.ff1=6" "_"}; do
           a=""; [ "$v" != "_" ] && a="--side-tiles $v"
           TAILN=0 run x3_x 300 python bench.py --dtype bf16x3 --steps ${X3T_STEPS:-5} --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder ${X3T_EXTRA:---no-batched} $a
           echo "--- bf16x3 tiles [$v] ${X3T_EXTRA:-}: $(grep -o '"value": [0-9.]*' gpurun_out/x3_x.log | head -1) $(grep -o '"mel_frames_per_s": [0-9.]*' gpurun_out/x3_x.log | head -1)"
         done ;;
    prof_modes) for m in bf16x3 fp32; do
           rm -rf /tmp/profm; run prof_$m 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profm -- python bench.py --dtype $m --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-batched --no-video2roll --no-vocoder --no-parity-mode --no-configs
           mkdir -p gpurun_out/prof; cp /tmp/profm/*/*_kernel_stats.csv gpurun_out/prof/kernel_stats_$m.csv
         done ;;
    maintile) SW="${MAIN_SWEEP:--1 1 2 7}"
         for v in $SW; do
           TAILN=0 run mt_$v 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder --no-batched --main-tile $v
           echo "--- main_tile=$v: $(grep -o '"value": [0-9.]*' gpurun_out/mt_$v.log | head -1)"
         done ;;
    dist2) # two ranks of the REAL sampler (graph on) + one all-gather on the one device, gloo; the launcher touches no GPU
           for m in bf16 bf16x3; do
             TAILN=6 run dist2_$m 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 scripts/dist_rehearsal.py $m 5
           done
           cat gpurun_out/dist2_bf16.log gpurun_out/dist2_bf16x3.log | grep -E "^rank|REHEARSAL" > gpurun_out/dist_rehearsal.txt ;;
    bench2) V2A_BENCH_BACKEND=gloo run bench2 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 ;;
    batch) for b in 2 4 8; do
             TAILN=0 run batch_$b 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --clips-per-gpu $b
             echo "--- B=$b: $(grep -E 'timed' gpurun_out/batch_$b.log) $(grep -o '"value": [0-9.]*' gpurun_out/batch_$b.log)"
           done ;;
    shapes8) TAILN=45 run shapes8 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --shapes --single-stream --clips-per-gpu 8 --no-parity-mode --no-configs --no-video2roll --no-vocoder ;;
    shapes1) TAILN=45 run shapes1 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --shapes --single-stream --no-batched --no-parity-mode --no-configs --no-video2roll --no-vocoder ;;
    configs) TAILN=2 run cfg_v2p 300 python bench.py --steps 2 --warmup 1 --v2p --no-cpu-baseline --no-roofline --no-batched
             TAILN=2 run cfg_cascade 300 python bench.py --steps 2 --warmup 1 --cascade 3 --clips-per-gpu 4 --no-cpu-baseline --no-roofline --no-batched ;;
    big) for t in 0 1; do
           export V2A_GEMM_BIG=$t
           TAILN=0 run big_$t 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-batched --clips-per-gpu 8
           echo "--- big $t B=8: $(grep -E 'timed' gpurun_out/big_$t.log)"
         done; unset V2A_GEMM_BIG ;;
    prof) rm -rf /tmp/prof; run prof$DTS 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python bench.py $DTA --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-batched --no-video2roll --no-vocoder --no-parity-mode --no-configs
          mkdir -p gpurun_out/prof; cp /tmp/prof/*/*_kernel_stats.csv gpurun_out/prof/kernel_stats${DTS}_multistream.csv ;;
    prof1) rm -rf /tmp/prof1; run prof1$DTS 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python bench.py $DTA --steps 3 --warmup 1 --no-cpu-baseline --single-stream --no-roofline --no-batched --no-video2roll --no-vocoder --no-parity-mode --no-configs
          mkdir -p gpurun_out/prof; cp /tmp/prof1/*/*_kernel_stats.csv gpurun_out/prof/kernel_stats${DTS}_singlestream.csv ;;
    prof8) rm -rf /tmp/prof8; run prof8$DTS 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof8 -- python bench.py $DTA --steps 2 --warmup 1 --clips-per-gpu 8 --no-cpu-baseline --no-roofline --no-batched --no-video2roll --no-vocoder --no-parity-mode --no-configs
          mkdir -p gpurun_out/prof; cp /tmp/prof8/*/*_kernel_stats.csv gpurun_out/prof/kernel_stats${DTS}_8clips.csv ;;
    prof8s) rm -rf /tmp/prof8s; run prof8s$DTS 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof8s -- python bench.py $DTA --steps 2 --warmup 1 --clips-per-gpu 8 --single-stream --no-cpu-baseline --no-roofline --no-batched --no-video2roll --no-vocoder --no-parity-mode --no-configs
          mkdir -p gpurun_out/prof; cp /tmp/prof8s/*/*_kernel_stats.csv gpurun_out/prof/kernel_stats${DTS}_8clips_singlestream.csv ;;
    prof1_old) rm -rf /tmp/prof1; run prof1 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --single-stream --no-batched --no-video2roll --no-vocoder
          mkdir -p gpurun_out/prof; cp /tmp/prof1/*/*_kernel_stats.csv gpurun_out/prof/kernel_stats_singlestream.csv ;;
    prof_v2r) rm -rf /tmp/profv; run prof_v2r 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profv -- python bench.py --steps 1 --warmup 0 --cfm-steps 4 --no-cpu-baseline --no-roofline --no-batched
          mkdir -p gpurun_out/prof; cp /tmp/profv/*/*_kernel_stats.csv gpurun_out/prof/kernel_stats_video2roll.csv ;;
    q1) # quick one-clip / 8-clip throughput of the headline mode (or DT=...): no extras
        TAILN=0 run q1$DTS 300 python bench.py $DTA --steps ${Q_STEPS:-5} --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder ${Q_EXTRA:-}
        echo "--- q1 ${DT:-bf16x3} ${Q_EXTRA:-}: one clip $(grep -o '"value": [0-9.]*' gpurun_out/q1$DTS.log | head -1)  8 clips $(grep -o '"mel_frames_per_s": [0-9.]*' gpurun_out/q1$DTS.log | head -1)" ;;
    sprobe) TAILN=60 run sprobe${SP_TAG:-} 900 python scripts/split_probe.py ${SP_ARGS:-} ;;
    ksplit) run ksplit 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -k "split" ;;
    ab) # generic A/B of bench flags, alternating: AB_A="" AB_B="--flag" [AB_EXTRA="--clips-per-gpu 8"] [AB_N=2]
        for i in $(seq 1 ${AB_N:-2}); do for v in "${AB_A:-}" "${AB_B:-}"; do
          TAILN=0 run ab_x 400 python bench.py $DTA --steps ${AB_STEPS:-5} --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder ${AB_EXTRA:---no-batched} $v
          echo "--- ab [$v] ${AB_EXTRA:-}: $(grep -o '"value": [0-9.]*' gpurun_out/ab_x.log | head -1) $(grep -o '"mel_frames_per_s": [0-9.]*' gpurun_out/ab_x.log | head -1)"
        done; done ;;
    kattn) run kattn 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -k "attention or cfg_euler" ;;
    cutime) # chip-time accounting of one evaluation: single-stream trace (every kernel alone) + three-stream trace
        rm -rf /tmp/tr1 /tmp/tr3
        TAILN=2 run cut1$DTS 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -- python bench.py $DTA ${CU_EXTRA:-} --single-stream --steps 1 --warmup 1 --cfm-steps 8 --no-cpu-baseline --no-roofline --no-batched --no-parity-mode --no-configs --no-video2roll --no-vocoder
        TAILN=2 run cut3$DTS 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr3 -- python bench.py $DTA ${CU_EXTRA:-} --steps 1 --warmup 1 --cfm-steps 8 --no-cpu-baseline --no-roofline --no-batched --no-parity-mode --no-configs --no-video2roll --no-vocoder
        python scripts/cu_time.py /tmp/tr1 /tmp/tr3 > gpurun_out/cu_time$DTS${CU_TAG:-}.txt 2>&1; cat gpurun_out/cu_time$DTS${CU_TAG:-}.txt
        head -1 /tmp/tr1/*/*_kernel_trace.csv ;;
    *) echo "unknown step $s" ;;
  esac
done
exit 0
