# Sweep decoder contexts x frames in flight for bench.py on one box.  Usage: bash tools/gpu/run_sweep.sh "8 128" "12 128" ...   Output: gpurun_out/sweep.log
ulimit -c 0
mkdir -p gpurun_out; : > gpurun_out/sweep.log
for cfg in "$@"; do
  set -- $cfg
  echo "== contexts $1 inflight $2" >> gpurun_out/sweep.log
  timeout 600 python bench.py --no-cpu-baseline --contexts $1 --inflight $2 --batch $(($1*$2)) --steps 5 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_flight'])" >> gpurun_out/sweep.log 2>&1
done
cat gpurun_out/sweep.log
