# reconstruction families of a flight on three streams (JXLAMD_RECON_FORK=1, default) vs one after another (0); hardware queues 16 / 48
ulimit -c 0
mkdir -p gpurun_out/fork
for cfg in "1 16" "0 16" "1 48" "0 16" "1 16" "1 48"; do
  set -- $cfg
  JXLAMD_RECON_FORK=$1 GPU_MAX_HW_QUEUES=$2 timeout 900 python bench.py --no-cpu-baseline --steps ${STEPS:-16} --warmup 4 2>gpurun_out/fork/err.txt | tail -1 > gpurun_out/fork/b_$1_$2.json
  python - $1 $2 <<'PY'
import json, sys
f, q = sys.argv[1:3]
try:
    d = json.load(open(f"gpurun_out/fork/b_{f}_{q}.json")); print("fork", f, "queues", q, "value", d["value"], "ms/step", d["ms_per_step"], d["roofline"].get("stage_ms_per_flight"))
except Exception as e:
    print("fork", f, q, "failed", e); print(open("gpurun_out/fork/err.txt").read()[-600:])
PY
done
