# round 5: the mixed workload against contexts / flight length
ulimit -c 0
mkdir -p gpurun_out/r5w
for cfg in "8 16" "16 16" "16 8" "24 8" "16 32"; do set -- $cfg
timeout 600 python bench.py --workload mixed --contexts $1 --inflight $2 --steps 6 --warmup 1 --no-cpu-baseline 2>gpurun_out/r5w/err_$1_$2.txt | tail -1 > gpurun_out/r5w/mixed_$1_$2.json
python - $1 $2 <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r5w/mixed_{sys.argv[1]}_{sys.argv[2]}.json")); print("contexts", sys.argv[1], "inflight", sys.argv[2], d["value"], d["ms_per_step"], d["config"]["stage_ms_per_flight"])
except Exception as e:
    print("failed", e); print(open(f"gpurun_out/r5w/err_{sys.argv[1]}_{sys.argv[2]}.txt").read()[-1500:])
PY
done
