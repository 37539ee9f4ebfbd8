ulimit -c 0
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run() { timeout 900 python bench.py --no-cpu-baseline --steps 4096 $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 value',d['value'],d['roofline']['stage_ms_per_flight'])"; }
for rep in 1 2 3; do run default; done
