# config 5 under rocprofv3 --kernel-trace --stats, A10 + A11 as a pass and inside the writer: per-kernel times of the two forms
ulimit -c 0; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/c5prof; mkdir -p $O
cd /tmp
for mode in pass writer; do
  rm -rf /tmp/prof_$mode
  PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o bench -- python $R/bench.py --workload c5 --c5-post $mode --no-cpu-baseline --steps 6 --warmup 2 > /tmp/bench_$mode.log 2>&1
  grep -v "^[WE]2026" /tmp/bench_$mode.log | tail -1 > $O/c5_${mode}_under_rocprof.json; cut -c1-160 $O/c5_${mode}_under_rocprof.json
  cp /tmp/prof_$mode/bench_kernel_stats.csv $O/kernel_stats_c5_$mode.csv; head -14 $O/kernel_stats_c5_$mode.csv | cut -c1-150
done
