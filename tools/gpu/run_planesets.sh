# sub-flight size of the HF phase (JXLAMD_PLANE_SETS: frames per reconstruction + filter launch; 16 by default): fewer, larger launches vs memory
ulimit -c 0
mkdir -p gpurun_out/planesets
for ps in 16 32 8 16 32; do
  JXLAMD_PLANE_SETS=$ps timeout 900 python bench.py --no-cpu-baseline --distinct 0 --steps 16 --warmup 4 2>gpurun_out/planesets/err_$ps.txt | tail -1 > gpurun_out/planesets/b_$ps.json
  python - $ps <<'PY'
import json, sys
ps = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/planesets/b_{ps}.json")); print("plane_sets", ps, "value", d["value"], "ms/step", d["ms_per_step"], d["roofline"].get("stage_ms_per_flight"))
except Exception as e:
    print("plane_sets", ps, "failed", e); print(open(f"gpurun_out/planesets/err_{ps}.txt").read()[-400:])
PY
done
