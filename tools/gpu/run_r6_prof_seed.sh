# kernel stats of the bench on ONE cycled frame (JXLAMD_BENCH_SEEDS=$1): which kernels a slow frame spends its time in
ulimit -c 0; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof; mkdir -p $O
for s in $SEEDS; do
cd /tmp; rm -rf /tmp/prof
JXLAMD_BENCH_SEEDS=$s PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --no-cpu-baseline --distinct 0 --steps 12 --warmup 3 > /tmp/bench.log 2>&1
grep -v "^[WE]2026" /tmp/bench.log | tail -1 | cut -c1-150
cp /tmp/prof/bench_kernel_stats.csv $O/kernel_stats_seed$s.csv; head -12 $O/kernel_stats_seed$s.csv | cut -c1-60,100-200
done
