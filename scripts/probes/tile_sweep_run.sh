# end-to-end A/B of per-(stream, op) tile choices on one box: bench.py one clip, 5 timed samples each
mkdir -p gpurun_out
: > gpurun_out/tile_sweep.txt
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-configs --no-video2roll --no-vocoder --no-batched"
i=0
while read -r spec; do
  [ -z "$spec" ] && continue
  i=$((i+1))
  args=""
  [ "$spec" != "_" ] && args="$spec"
  timeout -k 10 200 $B $args > gpurun_out/ts_$i.log 2>&1
  rc=$?
  v=$(grep -o '"value": [0-9.]*' gpurun_out/ts_$i.log | head -1)
  echo "[$spec] rc=$rc $v" | tee -a gpurun_out/tile_sweep.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 99; fi
done <<'LIST'
_
--split-cross
_
--split-cross
_
--split-cross
LIST
cat gpurun_out/tile_sweep.txt
