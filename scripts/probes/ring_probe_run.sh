mkdir -p gpurun_out
timeout -k 10 300 python scripts/probes/xcd_probe.py \
 1564x10240x1280:6:ff:geglu 1536x10240x1280:6:ff:geglu 28x10240x1280:3:ff:geglu 28x10240x1280:2:ff:geglu 28x10240x1280:15:ff:geglu 28x10240x1280:-1:ff:geglu \
 1564x8192x1024:6:ff:geglu 1536x8192x1024:6:ff:geglu 28x8192x1024:3:ff:geglu \
 1564x4096x512:6:ff:geglu 1536x4096x512:6:ff:geglu 1564x4096x512:0:ff:geglu 1564x4096x512:12:ff:geglu \
 1536x10240x1280:6:ff:geglu+28x10240x1280:3:ff:geglu \
 > gpurun_out/rowsplit_probe.log 2>&1
echo rc=$?; cat gpurun_out/rowsplit_probe.log
