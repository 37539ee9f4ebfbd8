# one GPU call: -m gpu tests (TESTS=1), the bench line (quick: 8 committed frames; FULL=1: the default command incl. 256 distinct frames and
# the CPU baseline), optionally the full-size config-4 check (C4=1)
ulimit -c 0
mkdir -p gpurun_out/round
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/round/pytest_gpu.txt; tail -5 gpurun_out/round/pytest_gpu.txt; fi
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("bench value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"], "single", c["single_frame_latency_ms"], "h2d", c.get("h2d_included_MPps"), "distinct", c.get("distinct_frames"), "cpu", d["cpu_baseline"]["value"])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/round/bench_err.txt").read()[-1500:])
PY
}
timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 8 --warmup 2 $BENCH_ARGS 2>gpurun_out/round/bench_err.txt | tail -1 > gpurun_out/round/bench_quick.json; show gpurun_out/round/bench_quick.json
if [ -n "$FULL" ]; then timeout 900 python bench.py 2>gpurun_out/round/bench_err.txt | tail -1 > gpurun_out/round/bench.json; show gpurun_out/round/bench.json; fi
if [ -n "$C4" ]; then mkdir -p gpurun_out/c4; timeout 1500 python tools/gpu/c4_full.py 32768 32768 8 > gpurun_out/c4/c4_32768.txt 2>&1; echo c4 rc=$?; tail -16 gpurun_out/c4/c4_32768.txt; fi
