# contexts x flight-size sweep with the multi-stream LF kernel (JXLAMD_LF_MULTI=1) and without
ulimit -c 0
mkdir -p gpurun_out/lfm
for cfg in "1 16 64" "1 32 32" "1 24 40" "1 32 48" "1 48 24" "1 64 16" "0 32 32" "0 16 64"; do
  set -- $cfg
  JXLAMD_LF_MULTI=$1 timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 8 --warmup 2 --contexts $2 --inflight $3 2>gpurun_out/lfm/err_sweep.txt | tail -1 > gpurun_out/lfm/sweep_$1_$2_$3.json
  python - $1 $2 $3 <<'PY'
import json, sys
m, c, f = sys.argv[1:4]
try:
    d = json.load(open(f"gpurun_out/lfm/sweep_{m}_{c}_{f}.json")); print("LF_MULTI", m, "contexts", c, "inflight", f, "value", d["value"], "ms/step", d["ms_per_step"], d["roofline"].get("stage_ms_per_flight"))
except Exception as e:
    print("bench failed", m, c, f, e); print(open("gpurun_out/lfm/err_sweep.txt").read()[-800:])
PY
done
