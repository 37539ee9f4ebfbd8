# round 5: MA trees in block form (kBig) — parity of the alpha streams, then the mixed bench line again
ulimit -c 0
mkdir -p gpurun_out/r5p
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "block_form or round3_kinds or round4 or golden_vectors or rgba" 2>&1 | tail -8
timeout 600 python bench.py --workload mixed --steps 4 --warmup 1 --no-cpu-baseline 2>gpurun_out/r5p/mixed_err.txt | tail -1 > gpurun_out/r5p/mixed.json
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r5p/mixed.json")); print("mixed", d["value"], d["ms_per_step"], d["config"]["single_frame_latency_ms"], d["config"]["stage_ms_per_flight"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r5p/mixed_err.txt").read()[-2500:])
PY
