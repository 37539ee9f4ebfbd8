# more contexts with smaller HF pools (sub-flights): does more LF-phase overlap raise the throughput?
ulimit -c 0
mkdir -p gpurun_out/lfm
for cfg in "0 32 64 32 8" "1 32 64 32 8" "0 24 64 32 16" "1 24 64 32 16" "0 32 64 16 8" "1 48 64 16 8" "0 48 64 16 8"; do
  set -- $cfg
  JXLAMD_LF_MULTI=$1 JXLAMD_HF_SETS=$4 JXLAMD_PLANE_SETS=$5 timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 8 --warmup 2 --contexts $2 --inflight $3 2>gpurun_out/lfm/err_sweep.txt | tail -1 > gpurun_out/lfm/sweep2_$1_$2_$3_$4.json
  python - $1 $2 $3 $4 $5 <<'PY'
import json, sys
m, c, f, h, p = sys.argv[1:6]
try:
    d = json.load(open(f"gpurun_out/lfm/sweep2_{m}_{c}_{f}_{h}.json")); print("LF_MULTI", m, "contexts", c, "inflight", f, "hf_sets", h, "plane_sets", p, "value", d["value"], "ms/step", d["ms_per_step"], d["roofline"].get("stage_ms_per_flight"))
except Exception as e:
    print("bench failed", m, c, f, h, e); print(open("gpurun_out/lfm/err_sweep.txt").read()[-300:])
PY
done
