# round 5: after the contraction-proof colour transform: writer-post == decode + post_fused (sweep frames incl. zero-luma rows), parity, quick bench
ulimit -c 0
mkdir -p gpurun_out/r5k
timeout 1500 python -m pytest tests/test_post_stages.py tests/test_gpu_parity.py -x -q -m gpu -k "writer_post or config5 or pipeline or golden_vectors or bench_line or 16bit or batch_equals or demo_assets" 2>&1 | tail -5
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("bench value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"], "h2d", c.get("h2d_included_MPps"), "pool", c["lf_pool_bytes"], c["flights_repeated_for_lf_pool"])
except Exception as e:
    print("bench failed", e); print(open(sys.argv[1].replace(".json", "_err.txt")).read()[-1500:])
PY
}
run() { tag=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" 2>gpurun_out/r5k/bench_${tag}_err.txt | tail -1 > gpurun_out/r5k/bench_$tag.json; echo $tag; show gpurun_out/r5k/bench_$tag.json; }
run quick --distinct 0 --steps 12 --warmup 3
run default
run c5 --workload c5
