ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="${BENCH_ARGS:---inflight 128 --contexts 3}"
cd /tmp; rm -rf /tmp/prof
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py $ARGS --no-cpu-baseline > /tmp/bench.log 2>&1
tail -3 /tmp/bench.log | cut -c1-400
grep -v "^W2026" /tmp/bench.log | tail -5 | cut -c1-300; find /tmp/prof | head; f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); echo "stats: $f"; mkdir -p $R/gpurun_out/prof; cp "$f" $R/gpurun_out/prof/bench_kernel_stats.csv
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]: print(r['Name'][:60].ljust(60), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
