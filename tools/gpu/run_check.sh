ulimit -c 0
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run() { timeout 900 python bench.py --no-cpu-baseline --steps $3 --inflight $1 --contexts $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight $1 ctx $2 value',d['value'],d['roofline']['stage_ms_per_flight'], d['roofline'].get('kernel_ms_all_launches'))"; }
run 128 8 4096
run 128 8 8192
