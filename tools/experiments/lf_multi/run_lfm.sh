# multi-stream LF-coefficient kernel: flight parity tests, then the quick bench with and without it
ulimit -c 0
mkdir -p gpurun_out/lfm
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flight or batch or smoke" 2>&1 | tail -8 > gpurun_out/lfm/pytest.txt; tail -4 gpurun_out/lfm/pytest.txt
for m in 1 0; do
  JXLAMD_LF_MULTI=$m timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 8 --warmup 2 2>gpurun_out/lfm/err_$m.txt | tail -1 > gpurun_out/lfm/bench_$m.json
  python - $m <<'PY'
import json, sys
m = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/lfm/bench_{m}.json")); print("LF_MULTI", m, "value", d["value"], "ms/step", d["ms_per_step"], d["roofline"].get("stage_ms_per_flight"), "single", d["config"].get("single_frame_latency_ms"))
except Exception as e:
    print("bench failed", m, e); print(open(f"gpurun_out/lfm/err_{m}.txt").read()[-1500:])
PY
done
