ulimit -c 0
run() { timeout 900 python bench.py --no-cpu-baseline --steps 4096 $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 value',d['value'],d['roofline']['stage_ms_per_flight'])"; }
for rep in 1 2; do for g in 3,2,2 4,2,2 3,0,2 3,2,3 2,2,2 0,0,0; do JXLAMD_STAGE_GATES=$g run $g; done; done
