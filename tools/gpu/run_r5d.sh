# round 5: sub-batch size of the data-parallel stages (frames per launch of the reconstruction / filter kernels)
ulimit -c 0
mkdir -p gpurun_out/r5d
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("bench value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"], "single", c["single_frame_latency_ms"], "h2d", c.get("h2d_included_MPps"), "ctx", c["decoder_contexts"], "P", c["frames_in_flight"])
except Exception as e:
    print("bench failed", e); print(open(sys.argv[1].replace(".json", "_err.txt")).read()[-1500:])
PY
}
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 12 --warmup 3 "$@" 2>gpurun_out/r5d/bench_${tag}_err.txt | tail -1 > gpurun_out/r5d/bench_$tag.json; echo $tag; show gpurun_out/r5d/bench_$tag.json; }
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch_equals or sparse or subflights or config3 or config5 or corrupt" 2>&1 | tail -4
run sets64
JXLAMD_PLANE_SETS=16 run sets16
JXLAMD_PLANE_SETS=32 run sets32
run sets64b
JXLAMD_PLANE_SETS=16 run sets16b
run c24 --contexts 24
run c20 --contexts 20
run c16x96 --inflight 96
