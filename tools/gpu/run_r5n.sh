# round 5: k_recon_dct32_b with its cosine operand in registers (12.7 KB of LDS instead of 16.8)
ulimit -c 0
mkdir -p gpurun_out/r5n
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_vectors or batch_equals or config3 or demo_assets or sparse or 4k_frame" 2>&1 | tail -3
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("bench value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"], "h2d", c.get("h2d_included_MPps"), "pool", c["lf_pool_bytes"], c["flights_repeated_for_lf_pool"])
except Exception as e:
    print("bench failed", e); print(open(sys.argv[1].replace(".json", "_err.txt")).read()[-1500:])
PY
}
run() { tag=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" 2>gpurun_out/r5n/bench_${tag}_err.txt | tail -1 > gpurun_out/r5n/bench_$tag.json; echo $tag; show gpurun_out/r5n/bench_$tag.json; }
run quick --distinct 0 --steps 12 --warmup 3
run quick2 --distinct 0 --steps 12 --warmup 3
run quick3 --distinct 0 --steps 12 --warmup 3
