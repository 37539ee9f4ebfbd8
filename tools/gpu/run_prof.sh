ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="${BENCH_ARGS:---inflight 128 --contexts 3}"
cd /tmp; rm -rf /tmp/prof
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py $ARGS --no-cpu-baseline > /tmp/bench.log 2>&1
grep -v "^[WE]2026" /tmp/bench.log | tail -1 | cut -c1-200
mkdir -p $R/gpurun_out/prof; cp /tmp/prof/bench_kernel_stats.csv $R/gpurun_out/prof/
python - /tmp/prof/bench_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]: print(r['Name'][:60].ljust(60), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
head -1 /tmp/prof/bench_kernel_trace.csv
python $R/tools/gpu/trace_summary.py /tmp/prof/bench_kernel_trace.csv 120
