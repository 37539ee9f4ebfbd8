# round 5: frames with patch references ride in flights — parity of everything batched / composed, then the mixed line A/B
ulimit -c 0
mkdir -p gpurun_out/r5x
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_post_stages.py -x -q -m gpu -k "batch or flight or kinds or upsampl or noise or spline or jpeg or golden_vectors or writer or anim or patch or screenshot" 2>&1 | tail -5
for v in 1 2 0; do
JXLAMD_COMPOSE_IN_FLIGHTS=$v timeout 600 python bench.py --workload mixed --steps 6 --warmup 1 --no-cpu-baseline 2>gpurun_out/r5x/mixed_err_$v.txt | tail -1 > gpurun_out/r5x/mixed_$v.json
python - $v <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r5x/mixed_{v}.json")); print("mixed compose_in_flights", v, d["value"], d["ms_per_step"], d["config"]["stage_ms_per_flight"])
except Exception as e:
    print("failed", e); print(open(f"gpurun_out/r5x/mixed_err_{v}.txt").read()[-2500:])
PY
done
