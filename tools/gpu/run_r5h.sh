# round 5: context-map prefetch in k_pass_flat; C-ABI sharded-local decode
ulimit -c 0
mkdir -p gpurun_out/r5h
timeout 1200 python -m pytest tests/test_band_sharded.py tests/test_gpu_parity.py -x -q -m gpu -k "sharded_local or bands_equal or sparse or flat_passgroup or corrupt or config3 or batch_equals" 2>&1 | tail -5
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("bench value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"], "single", c["single_frame_latency_ms"], "h2d", c.get("h2d_included_MPps"))
except Exception as e:
    print("bench failed", e); print(open(sys.argv[1].replace(".json", "_err.txt")).read()[-1500:])
PY
}
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 12 --warmup 3 "$@" 2>gpurun_out/r5h/bench_${tag}_err.txt | tail -1 > gpurun_out/r5h/bench_$tag.json; echo $tag; show gpurun_out/r5h/bench_$tag.json; }
run a
run b
run c
