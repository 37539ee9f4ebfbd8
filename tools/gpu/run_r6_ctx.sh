ulimit -c 0
for cfg in "16 64" "16 48" "16 32" "12 64" "20 64" "24 48" "16 64"; do
  set -- $cfg
  timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --contexts $1 --inflight $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[ctx] contexts $1 inflight $2 value', d['value'], 'ms/step', d['ms_per_step'], d['roofline']['stage_ms_per_flight'])"
done
