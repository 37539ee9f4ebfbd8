# round 5: LF table pool no longer raised by the HF-metadata channels' compact tables; parity hygiene tests
ulimit -c 0
mkdir -p gpurun_out/r5g
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "16bit or forced_epf or pq16 or golden_vectors or config3" 2>&1 | grep -v "^$" | tail -12
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("bench value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"], "single", c["single_frame_latency_ms"], "h2d", c.get("h2d_included_MPps"), "pool", c["lf_pool_bytes"], "pool retries", c["flights_repeated_for_lf_pool"])
except Exception as e:
    print("bench failed", e); print(open(sys.argv[1].replace(".json", "_err.txt")).read()[-1500:])
PY
}
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 12 --warmup 3 "$@" 2>gpurun_out/r5g/bench_${tag}_err.txt | tail -1 > gpurun_out/r5g/bench_$tag.json; echo $tag; show gpurun_out/r5g/bench_$tag.json; }
run a
run b
run c
