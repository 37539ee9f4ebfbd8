# round 5: --workload mixed baseline (composed frames still one by one), with the flight trace and a rocprof kernel summary
ulimit -c 0
mkdir -p gpurun_out/r5o
timeout 600 python bench.py --workload mixed --steps 4 --warmup 1 2>gpurun_out/r5o/mixed_err.txt | tail -1 > gpurun_out/r5o/mixed.json
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r5o/mixed.json")); print("mixed", d["value"], d["ms_per_step"], d["config"]["single_frame_latency_ms"], d["config"]["stage_ms_per_flight"], d["cpu_baseline"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r5o/mixed_err.txt").read()[-2500:])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r5o/prof -o mixed -- python /root/repo/bench.py --workload mixed --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/r5o/mixed_prof.json 2>/root/repo/gpurun_out/r5o/mixed_prof_err.txt
cd /root/repo
f=$(ls gpurun_out/r5o/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -25 "$f"
rm -f gpurun_out/r5o/prof/*kernel_trace.csv gpurun_out/r5o/prof/*agent_info.csv
