# round-6 record: the full -m gpu suite, the driver's bench command (+ the same under rocprofv3 --kernel-trace --stats, + with the flight trace)
ulimit -c 0; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O
cd $R
if [ -z "$SKIP_TESTS" ]; then timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt; fi
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_err.txt | tail -1 > $O/bench_driver_command.json; cut -c1-260 $O/bench_driver_command.json
JXLAMD_TRACE_FLIGHT=1 timeout 900 python bench.py --no-cpu-baseline --gpus 1 --steps 20 --warmup 5 2>$O/flight_trace.txt | tail -1 > $O/bench_traced.json; python tools/gpu/flight_summary.py $O/flight_trace.txt
cd /tmp; rm -rf /tmp/prof
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --no-cpu-baseline --gpus 1 --steps 20 --warmup 5 > /tmp/bench.log 2>&1
grep -v "^[WE]2026" /tmp/bench.log | tail -1 > $O/bench_under_rocprof.json; cut -c1-200 $O/bench_under_rocprof.json
cp /tmp/prof/bench_kernel_stats.csv $O/kernel_stats_bench.csv; head -5 $O/kernel_stats_bench.csv | cut -c1-150
