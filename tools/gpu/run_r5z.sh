# round 5: the weighted-predictor LF loop for codes of more than 64 clusters (libjxl's one-shot files: 128)
ulimit -c 0
mkdir -p gpurun_out/r5z
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "block_form or kinds or rgba or golden_vectors or batch_equals or lossless or composed" 2>&1 | tail -3
python tools/gpu/which_general.py 2>&1 | tail -3
timeout 600 python bench.py --workload mixed --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5z/mixed.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5z/mixed.json")); print("mixed", d["value"], d["ms_per_step"], d["config"]["single_frame_latency_ms"], d["config"]["stage_ms_per_flight"])
PY
bash tools/gpu/run_rgba4k_prof.sh 2>&1 | grep "4k " | sed -n '2p;5p'
