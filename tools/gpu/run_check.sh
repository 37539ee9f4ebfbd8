ulimit -c 0
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for cfg in "128 4 3072" "64 8 3072"; do set -- $cfg; timeout 900 python bench.py --no-cpu-baseline --inflight $1 --contexts $2 --steps $3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight $1 ctx $2 steps $3 value',d['value'],d['roofline']['stage_ms_per_flight'], d.get('single_frame_ms'))"; done
