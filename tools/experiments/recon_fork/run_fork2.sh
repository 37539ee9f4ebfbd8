ulimit -c 0
mkdir -p gpurun_out/fork
for cfg in "1 24 8 64" "1 32 10 64" "0 16 8 64" "1 24 8 128" "1 16 5 128"; do
  set -- $cfg
  JXLAMD_RECON_FORK=$1 GPU_MAX_HW_QUEUES=$2 timeout 900 python bench.py --no-cpu-baseline --steps 16 --warmup 4 --contexts $3 --inflight $4 2>gpurun_out/fork/err.txt | tail -1 > gpurun_out/fork/c_$1_$2_$3_$4.json
  python - $1 $2 $3 $4 <<'PY'
import json, sys
f, q, c, p = sys.argv[1:5]
try:
    d = json.load(open(f"gpurun_out/fork/c_{f}_{q}_{c}_{p}.json")); print("fork", f, "queues", q, "contexts", c, "x", p, "value", d["value"], "ms/step", d["ms_per_step"], d["roofline"].get("stage_ms_per_flight"))
except Exception as e:
    print("fork", f, q, c, p, "failed", e); print(open("gpurun_out/fork/err.txt").read()[-600:])
PY
done
