# the bench command under rocprofv3 --kernel-trace --stats + the line it printed (the two figures that must agree)
ulimit -c 0; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp; rm -rf /tmp/prof
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --no-cpu-baseline > /tmp/bench.log 2>&1
grep -v "^[WE]2026" /tmp/bench.log | tail -1 > $O/bench_under_rocprof.json; cut -c1-200 $O/bench_under_rocprof.json
cp /tmp/prof/bench_kernel_stats.csv $O/kernel_stats_bench.csv; head -4 $O/kernel_stats_bench.csv | cut -c1-170
