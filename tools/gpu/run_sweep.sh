# Sweep decoder contexts x frames in flight for bench.py on one box.  Output: gpurun_out/sweep.log
ulimit -c 0
mkdir -p gpurun_out; : > gpurun_out/sweep.log
for cfg in "8 128" "12 64" "16 64" "12 96" "16 32" "6 128" "10 128"; do
  set -- $cfg
  echo "== contexts $1 inflight $2" >> gpurun_out/sweep.log
  timeout 600 python bench.py --no-cpu-baseline --contexts $1 --inflight $2 --batch $((($1*$2+255)/256*256)) --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_flight'])" >> gpurun_out/sweep.log 2>&1
done
cat gpurun_out/sweep.log
