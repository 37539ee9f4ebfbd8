# stage floors of the flight pipeline: the driver's bench command on ONE cycled frame (so that stale slot contents are a frame of the same layout),
# with stages left out of the timed steps by jxlamd_debug_set_ablate.  1 LF, 2 PassGroup, 4 reconstruction + filters + writer.
ulimit -c 0
mkdir -p gpurun_out/ablate
for m in 0 1 2 4 3 5 6; do
  JXLAMD_BENCH_SEEDS=3 JXLAMD_BENCH_ABLATE=$m timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 20 --warmup 5 2>gpurun_out/ablate/err_$m.txt | tail -1 > gpurun_out/ablate/ablate_$m.json
  python - $m <<'PY'
import json, sys
m = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/ablate/ablate_{m}.json")); print("[ablate]", m, "ms/step", d["ms_per_step"], "value", d["value"], d["roofline"]["stage_ms_per_flight"])
except Exception as e:
    print("[ablate]", m, "failed", e); print(open(f"gpurun_out/ablate/err_{m}.txt").read()[-800:])
PY
done
