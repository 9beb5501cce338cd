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
--side-tiles t.cross=12,t.out=12,t.ff2=12,f.cross=12,f.out=12,f.ff2=12
--main-tile 12 --side-tiles t.cross=12,t.out=12,t.ff2=12,f.cross=12,f.out=12,f.ff2=12
--side-tiles t.cross=12,t.out=12,t.ff2=12,f.cross=12,f.out=12,f.ff2=12,a.qkv=3
--side-tiles t.cross=12,t.out=12,t.ff2=12,f.cross=12,f.out=12,f.ff2=12,t.qkv=12
--side-tiles t.cross=12,t.out=12,t.ff2=12,f.cross=3,f.out=3,f.ff2=3
--side-tiles t.cross=15,t.out=15,t.ff2=15,f.cross=3,f.out=3,f.ff2=3
--main-tile 15 --side-tiles t.cross=12,t.out=12,t.ff2=12,f.cross=12,f.out=12,f.ff2=12
--main-tile 14 --side-tiles t.cross=12,t.out=12,t.ff2=12,f.cross=12,f.out=12,f.ff2=12
--side-tiles t.cross=12,t.out=12,t.ff2=12,f.cross=12,f.out=12,f.ff2=12,f.qkv=3,f.ff1=0
_
LIST
cat gpurun_out/tile_sweep.txt
