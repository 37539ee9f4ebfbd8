# round 5: full -m gpu suite on the tree with block-form trees + composed frames in flights, then the mixed line (with its CPU baseline) and the RGBA 4K kernel stats
ulimit -c 0
mkdir -p gpurun_out/r5v
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 900 python bench.py --workload mixed 2>gpurun_out/r5v/mixed_err.txt | tail -1 > gpurun_out/r5v/bench_mixed.json
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r5v/bench_mixed.json")); print("mixed", d["value"], d["ms_per_step"], d["config"]["single_frame_latency_ms"], d["cpu_baseline"]["value"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r5v/mixed_err.txt").read()[-2500:])
PY
bash tools/gpu/run_rgba4k_prof.sh > gpurun_out/r5v/rgba4k.txt 2>&1; grep "4k " gpurun_out/r5v/rgba4k.txt | head -3
