ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="${BENCH_ARGS:-}"
cd /tmp; rm -rf /tmp/prof
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py $ARGS --no-cpu-baseline > /tmp/bench.log 2>&1
grep -v "^[WE]2026" /tmp/bench.log | tail -1 | cut -c1-200
python - /tmp/prof/bench_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]: print(r['Name'][:50].ljust(50), r['Calls'], 'avg ms %.2f'%(float(r['AverageNs'])/1e6), 'min %.2f'%(float(r['MinNs'])/1e6), 'max %.2f'%(float(r['MaxNs'])/1e6), r['Percentage'])
PY
