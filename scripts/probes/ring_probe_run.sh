mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_video2roll_gpu.py tests/test_encodec_gpu.py -q -m gpu -x --tb=short > gpurun_out/k3.log 2>&1; echo rc=$?; tail -5 gpurun_out/k3.log
timeout -k 10 300 python scripts/probes/xcd_probe.py \
 1564x1024x4096:3 1564x1024x64:3 1564x1024x1024:3 1564x1024x1024:15 1564x1024x2816:3 1564x1280x5120:15 1564x1280x1024:15 1564x512x2048:3 1564x512x512:3 \
 1564x3088x1024:0:ff:store 1564x3088x1024:3:ff:store 782x1040x1024:3:ff:store \
 1564x8192x1024:6:ff:geglu 1564x10240x1280:6:ff:geglu 12512x1024x4096:6 \
 > gpurun_out/ring_probe6.log 2>&1
echo rc=$?; cat gpurun_out/ring_probe6.log
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder > gpurun_out/b6.log 2>&1; grep -o '"value": [0-9.]*\|"mel_frames_per_s": [0-9.]*' gpurun_out/b6.log
