# shared HF pools: contexts x share sweep (quick bench, 8 cycled frames, longer runs so that every context sees several flights)
ulimit -c 0
mkdir -p gpurun_out/share
for cfg in "16 1 64" "32 2 64" "24 2 64" "48 3 64" "32 2 32" "32 2 48"; do
  set -- $cfg
  timeout 900 python bench.py --no-cpu-baseline --distinct 0 --steps 16 --warmup 4 --contexts $1 --share $2 --inflight $3 2>gpurun_out/share/err.txt | tail -1 > gpurun_out/share/b_$1_$2_$3.json
  python - $1 $2 $3 <<'PY'
import json, sys
c, s, f = sys.argv[1:4]
try:
    d = json.load(open(f"gpurun_out/share/b_{c}_{s}_{f}.json")); print("contexts", c, "share", s, "inflight", f, "value", d["value"], "ms/step", d["ms_per_step"], d["roofline"].get("stage_ms_per_flight"), "h2d", d["config"].get("h2d_included_MPps"))
except Exception as e:
    print("bench failed", c, s, f, e); print(open("gpurun_out/share/err.txt").read()[-400:])
PY
done
