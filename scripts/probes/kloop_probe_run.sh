mkdir -p gpurun_out
timeout -k 10 400 python scripts/probes/kloop_probe.py > gpurun_out/kloop_probe.log 2>&1; echo rc=$?
cat gpurun_out/kloop_probe.log
hipcc --offload-arch=gfx950 -O3 -o /tmp/fill_probe scripts/probes/fill_probe.hip && for k in 1024 4096 16384; do timeout -k 10 120 /tmp/fill_probe $k; done > gpurun_out/fill_probe.log 2>&1
cat gpurun_out/fill_probe.log
