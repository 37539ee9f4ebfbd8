# Round 5 additions to the evidence: the mixed-content line (with its CPU baseline), the same command under rocprofv3 --kernel-trace --stats, kernel stats of a
# default-settings RGBA 4K frame (run_rgba4k_prof.sh) and of single decodes of one 1080p screenshot (patch dictionary)
ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 900 python bench.py --workload mixed > $O/bench_mixed.log 2>&1; tail -1 $O/bench_mixed.log > $O/bench_mixed.json; cut -c1-300 $O/bench_mixed.json
cd /tmp; rm -rf /tmp/profm
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profm -o mixed -- python $R/bench.py --workload mixed --no-cpu-baseline > /tmp/mixed.log 2>&1
grep -v "^[WE]2026" /tmp/mixed.log | tail -1 > $O/bench_mixed_under_rocprof.json; cut -c1-200 $O/bench_mixed_under_rocprof.json
cp /tmp/profm/mixed_kernel_stats.csv $O/kernel_stats_mixed.csv; head -14 $O/kernel_stats_mixed.csv | cut -c1-160
cd $R
bash tools/gpu/run_rgba4k_prof.sh > $O/rgba4k.txt 2>&1; grep "4k " $O/rgba4k.txt | head -2
cp gpurun_out/rgba4k/kernel_stats_rgba4k_d1.csv $O/kernel_stats_rgba4k_d1.csv; cp gpurun_out/rgba4k/kernel_stats_rgba4k_lossless_e3.csv $O/kernel_stats_rgba4k_lossless_e3.csv
cp /tmp/jxlamd_bench_frames/mixed_seed0.jxl /tmp/shot1080.jxl
cd /tmp; rm -rf /tmp/profs
JXLAMD_PROF_FILE=/tmp/shot1080.jxl PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs -o shot -- python $R/tools/prof_decode.py 5 > $O/screenshot1080.log 2>&1
cp /tmp/profs/shot_kernel_stats.csv $O/kernel_stats_screenshot_1080p.csv; grep "4k " $O/screenshot1080.log | tail -2; head -8 $O/kernel_stats_screenshot_1080p.csv | cut -c1-160
