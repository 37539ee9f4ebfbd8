# Collect the round's evidence on the GPU box: default bench line, rocprofv3 kernel stats of the bench command and of
# sequential single-frame decodes.  Outputs land in gpurun_out/prof/ (copied into profiles/ by hand).
ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_n1.log 2>&1; tail -1 $O/bench_n1.log > $O/bench_n1.json; cut -c1-400 $O/bench_n1.json
cd /tmp; rm -rf /tmp/prof
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --no-cpu-baseline > /tmp/bench.log 2>&1
grep -v "^[WE]2026" /tmp/bench.log | tail -1 > $O/bench_under_rocprof.json; cut -c1-300 $O/bench_under_rocprof.json
cp /tmp/prof/bench_kernel_stats.csv $O/kernel_stats_bench.csv
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o single -- python $R/tools/prof_decode.py 5 > /tmp/single.log 2>&1
cp /tmp/prof1/single_kernel_stats.csv $O/kernel_stats_single.csv
head -12 $O/kernel_stats_bench.csv | cut -c1-160
head -12 $O/kernel_stats_single.csv | cut -c1-160
# round 3 additions: the config-5 line (PQ-16 EPF=3 -> tone map -> F16, post stages fused) and the driver's own command line
cd $R
timeout 900 python bench.py --workload c5 > $O/bench_c5.log 2>&1; tail -1 $O/bench_c5.log > $O/bench_c5.json; cut -c1-300 $O/bench_c5.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2>&1; tail -1 $O/bench_driver.log > $O/bench_driver_cmd.json; cut -c1-300 $O/bench_driver_cmd.json
JXLAMD_BENCH_FILES=bench_data/real4k_summer_nature.jxl timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | tail -1 > $O/bench_real4k.json; cut -c1-200 $O/bench_real4k.json
