# config 5 with A10 + A11 as a pass over the stored RGBA16 (rounds 2-3) and inside the decoder's writer (round 4, jxlamd_decoder_set_writer_post), same box
ulimit -c 0
mkdir -p gpurun_out/c5post
for mode in pass writer pass writer; do
  timeout 900 python bench.py --workload c5 --no-cpu-baseline --c5-post $mode --steps 8 --warmup 2 2>gpurun_out/c5post/err_$mode.txt | tail -1 > gpurun_out/c5post/c5_$mode.json
  python - $mode <<'PY'
import json, sys
m = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/c5post/c5_{m}.json")); print("c5", m, "value", d["value"], "ms/step", d["ms_per_step"], d["roofline"].get("stage_ms_per_flight"), "single", d["config"]["single_frame_latency_ms"])
except Exception as e:
    print("c5", m, "failed", e); print(open(f"gpurun_out/c5post/err_{m}.txt").read()[-1500:])
PY
done
