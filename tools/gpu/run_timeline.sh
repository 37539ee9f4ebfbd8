# kernel timeline of the bench (rocprofv3 --kernel-trace): per-dispatch start / end -> tools/gpu/timeline_analyse.py
ulimit -c 0; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/timeline; mkdir -p $O
cd /tmp; rm -rf /tmp/tl
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o bench -- python $R/bench.py --no-cpu-baseline --distinct 0 --steps 12 --warmup 3 > /tmp/tl.log 2>&1
grep -v "^[WE]2026" /tmp/tl.log | tail -1 | cut -c1-200
f=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
python $R/tools/gpu/timeline_analyse.py "$f" > $O/timeline_summary.txt 2>&1
cat $O/timeline_summary.txt
