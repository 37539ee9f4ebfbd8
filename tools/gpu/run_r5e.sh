# round 5: the whole -m gpu suite + quick bench (exact LF pool floor)
ulimit -c 0
mkdir -p gpurun_out/r5e
timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5e/pytest_gpu.txt; tail -5 gpurun_out/r5e/pytest_gpu.txt
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("bench value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"], "single", c["single_frame_latency_ms"], "h2d", c.get("h2d_included_MPps"), "pool", c["lf_pool_bytes"], "pool retries", c["flights_repeated_for_lf_pool"])
except Exception as e:
    print("bench failed", e); print(open(sys.argv[1].replace(".json", "_err.txt")).read()[-1500:])
PY
}
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 12 --warmup 3 "$@" 2>gpurun_out/r5e/bench_${tag}_err.txt | tail -1 > gpurun_out/r5e/bench_$tag.json; echo $tag; show gpurun_out/r5e/bench_$tag.json; }
run a
run b
