# LF workgroups per CU capped through their LDS request (JXLAMD_LF_LDS_FLOOR): does freeing LDS for the data-parallel kernels raise the rate?
ulimit -c 0
mkdir -p gpurun_out/lffloor
for cfg in "0 16 1" "55000 16 1" "82000 16 1" "41000 16 1" "55000 24 2" "82000 24 2" "55000 32 2"; do
  set -- $cfg
  JXLAMD_LF_LDS_FLOOR=$1 timeout 900 python bench.py --no-cpu-baseline --distinct 0 --steps 16 --warmup 4 --contexts $2 --share $3 2>gpurun_out/lffloor/err.txt | tail -1 > gpurun_out/lffloor/b_$1_$2_$3.json
  python - $1 $2 $3 <<'PY'
import json, sys
a, c, s = sys.argv[1:4]
try:
    d = json.load(open(f"gpurun_out/lffloor/b_{a}_{c}_{s}.json")); print("floor", a, "contexts", c, "share", s, "value", d["value"], "ms/step", d["ms_per_step"], d["roofline"].get("stage_ms_per_flight"))
except Exception as e:
    print("bench failed", a, c, s, e); print(open("gpurun_out/lffloor/err.txt").read()[-300:])
PY
done
