# config 5 with A10 + A11 as a pass over the stored RGBA16 (rounds 2-3) and inside the decoder's writer (round 4, jxlamd_decoder_set_writer_post), same box;
# CFGS = "mode contexts inflight" triples
ulimit -c 0
mkdir -p gpurun_out/c5post
CFGS=${CFGS:-"pass 16 32;writer 16 32;pass 16 32;writer 16 32"}
IFS=';' read -ra LIST <<< "$CFGS"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  timeout 900 python bench.py --workload c5 --no-cpu-baseline --c5-post $1 --contexts $2 --inflight $3 --steps ${STEPS:-8} --warmup 2 2>gpurun_out/c5post/err_$1.txt | tail -1 > gpurun_out/c5post/c5_$1_$2_$3.json
  python - $1 $2 $3 <<'PY'
import json, sys
m, c, p = sys.argv[1:4]
try:
    d = json.load(open(f"gpurun_out/c5post/c5_{m}_{c}_{p}.json")); print("c5", m, c, "x", p, "value", d["value"], "ms/step", d["ms_per_step"], d["roofline"].get("stage_ms_per_flight"), "single", d["config"]["single_frame_latency_ms"])
except Exception as e:
    print("c5", m, c, p, "failed", e); print(open(f"gpurun_out/c5post/err_{m}.txt").read()[-1500:])
PY
done
