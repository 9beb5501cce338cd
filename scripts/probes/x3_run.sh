mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x --tb=short > gpurun_out/all4.log 2>&1; echo rc=$?; tail -15 gpurun_out/all4.log
timeout -k 10 300 python bench.py --dtype bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder > gpurun_out/bx3.log 2>&1; grep -o '"value": [0-9.]*\|"mel_frames_per_s": [0-9.]*' gpurun_out/bx3.log
