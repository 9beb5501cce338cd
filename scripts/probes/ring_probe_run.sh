mkdir -p gpurun_out
S=""
for shp in 12512x1024x1024 12512x1024x4096 12512x1024x3840 12512x1280x1024 12512x1280x5120 12512x1280x2304 12512x512x512 12512x512x2048 12512x512x1536 6256x1040x1024; do
  for t in -1 0 1 12 14 15 6; do S="$S $shp:$t"; done
done
for shp in 12512x3088x1024 12512x3088x1280 12512x1552x512; do
  for t in -1 0 1 12 15 6; do S="$S $shp:$t:ff:store"; done
done
for shp in 12512x8192x1024 12512x10240x1280 12512x4096x512; do
  for t in -1 0 12 6; do S="$S $shp:$t:ff:geglu"; done
done
timeout -k 10 500 python scripts/probes/xcd_probe.py $S > gpurun_out/clips8_probe.log 2>&1
echo rc=$?; cat gpurun_out/clips8_probe.log
