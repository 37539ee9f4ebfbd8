# contexts x flight size x hardware queues
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ctx; mkdir -p $O; cd $R
run() { GPU_MAX_HW_QUEUES=$1 python bench.py --no-cpu-baseline --distinct 0 --steps 40 --warmup 4 --contexts $2 --inflight $3 2>$O/err.txt > $O/c_$1_$2_$3.json || tail -3 $O/err.txt
  python - $O/c_$1_$2_$3.json "$1 queues, $2 contexts x $3 frames" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"])
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
run 16 16 64
run 16 12 96
run 16 10 128
run 16 16 48
run 20 20 48
run 16 16 64
