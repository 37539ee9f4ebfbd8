# round 5: kernel stats of ONE flight alone (one context, 64 frames) and of the quick bench (16 contexts), rocprofv3 --kernel-trace --stats
ulimit -c 0; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5c; mkdir -p $O
cd /tmp; rm -rf /tmp/prof1 /tmp/prof2
summ() { python - "$1" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]: print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), 'avg ms %8.3f'%(float(r['AverageNs'])/1e6), 'min %8.3f'%(float(r['MinNs'])/1e6), 'max %8.3f'%(float(r['MaxNs'])/1e6), r['Percentage'])
PY
}
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o f -- python $R/tools/prof_flight.py 64 > $O/flight64.log 2>&1
cp /tmp/prof1/f_kernel_stats.csv $O/kernel_stats_flight64_alone.csv; echo "--- one flight of 64 alone"; summ $O/kernel_stats_flight64_alone.csv
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o b -- python $R/bench.py --no-cpu-baseline --distinct 0 --steps 12 --warmup 3 > /tmp/bench.log 2>&1
grep -v "^[WE]2026" /tmp/bench.log | tail -1 > $O/bench_under_rocprof.json; cut -c1-120 $O/bench_under_rocprof.json
cp /tmp/prof2/b_kernel_stats.csv $O/kernel_stats_bench.csv; echo "--- bench"; summ $O/kernel_stats_bench.csv
