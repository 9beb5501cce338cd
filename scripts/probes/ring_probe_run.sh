mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x --tb=short -k "split" > gpurun_out/k5.log 2>&1; echo rc=$?; tail -4 gpurun_out/k5.log
timeout -k 10 300 python scripts/probes/xcd_probe.py \
 1564x8192x1024:4:ff:s3geglu 1564x8192x1024:2:ff:s3geglu 1564x10240x1280:4:ff:s3geglu 1564x10240x1280:2:ff:s3geglu 1564x4096x512:4:ff:s3geglu 1564x4096x512:2:ff:s3geglu 1564x4096x512:3:ff:s3geglu \
 1564x3088x1024:0:ff:s3store 1564x3088x1024:1:ff:s3store 1564x3088x1024:2:ff:s3store 1564x3088x1024:3:ff:s3store 1564x3088x1024:4:ff:s3store \
 1564x3088x1280:2:ff:s3store 1564x3088x1280:3:ff:s3store 1564x3088x1280:4:ff:s3store 1564x1552x512:3:ff:s3store 1564x1552x512:0:ff:s3store 1564x1552x512:4:ff:s3store \
 1564x1024x2816:3:ff:s3resid 1564x1024x2816:0:ff:s3resid 1564x512x2048:3:ff:s3resid 1564x512x2048:0:ff:s3resid 1564x512x512:3:ff:s3resid 1564x512x512:0:ff:s3resid 782x1040x1024:3:ff:s3store 782x1040x1024:0:ff:s3store \
 > gpurun_out/s3_probe2.log 2>&1
echo rc=$?; cat gpurun_out/s3_probe2.log
timeout -k 10 300 python bench.py --dtype bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder > gpurun_out/bx3.log 2>&1; grep -o '"value": [0-9.]*\|"mel_frames_per_s": [0-9.]*' gpurun_out/bx3.log
