ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp; rm -rf /tmp/prof1
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o single -- python $R/tools/prof_decode.py 5 > /tmp/single.log 2>&1
python - /tmp/prof1/single_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]: print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'], r['Percentage'])
PY
cd $R
for cfg in "128 3"; do set -- $cfg; timeout 600 python bench.py --no-cpu-baseline --inflight $1 --contexts $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight $1 ctx $2 value',d['value'],d['roofline']['stage_ms_per_flight'])"; done
