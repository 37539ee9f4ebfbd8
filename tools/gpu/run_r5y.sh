# round 5: group-stream workgroups with a table pool sized by the frame's tree (dynamic LDS): parity of everything Modular, then the mixed line and the RGBA 4K frame
ulimit -c 0
mkdir -p gpurun_out/r5y
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lossless or block_form or kinds or rgba or modular or anim or composed or golden_vectors or batch_equals" 2>&1 | tail -4
for i in 1 2; do
timeout 600 python bench.py --workload mixed --no-cpu-baseline 2>gpurun_out/r5y/mixed_err.txt | tail -1 > gpurun_out/r5y/mixed_$i.json
python - $i <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r5y/mixed_{sys.argv[1]}.json")); print("mixed", d["value"], d["ms_per_step"], d["config"]["stage_ms_per_flight"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r5y/mixed_err.txt").read()[-2500:])
PY
done
bash tools/gpu/run_rgba4k_prof.sh 2>&1 | grep "4k " | sed -n '2p;5p'
