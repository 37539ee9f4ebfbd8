ulimit -c 0; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp; rm -rf /tmp/prof
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --workload c5 --no-cpu-baseline --steps 16 --warmup 4 > /tmp/bench.log 2>&1
grep -v "^[WE]2026" /tmp/bench.log | tail -1 | cut -c1-150
cp /tmp/prof/bench_kernel_stats.csv $O/kernel_stats_c5.csv
