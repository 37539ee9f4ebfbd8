# round 5, second GPU call: the sweep's fast writer + row prefetch — parity, A/B on one box, the flight trace, contexts x flight size with sparse lists
ulimit -c 0
mkdir -p gpurun_out/r5b
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_vectors or 4k_frame or batch_equals or sparse or flat_passgroup or config3 or 16bit or demo_assets or band" 2>&1 | tail -15 > gpurun_out/r5b/pytest.txt; tail -5 gpurun_out/r5b/pytest.txt
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("bench value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"], "single", c["single_frame_latency_ms"], "h2d", c.get("h2d_included_MPps"), "ctx", c["decoder_contexts"], "P", c["frames_in_flight"])
except Exception as e:
    print("bench failed", e); print(open(sys.argv[1].replace(".json", "_err.txt")).read()[-1500:])
PY
}
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 12 --warmup 3 "$@" 2>gpurun_out/r5b/bench_${tag}_err.txt | tail -1 > gpurun_out/r5b/bench_$tag.json; echo $tag; show gpurun_out/r5b/bench_$tag.json; }
run fast
JXLAMD_SWEEP_FAST=0 run general
run fast2
JXLAMD_SWEEP_FAST=0 run general2
run c16x128 --contexts 16 --inflight 128
run c24x64 --contexts 24 --inflight 64
run c32x64 --contexts 32 --inflight 64
run c12x128 --contexts 12 --inflight 128
JXLAMD_TRACE_FLIGHT=1 timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 6 --warmup 2 2>gpurun_out/r5b/trace.txt | tail -1 > gpurun_out/r5b/bench_trace.json; grep "LF streams" gpurun_out/r5b/trace.txt | tail -12
