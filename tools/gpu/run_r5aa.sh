# round 5: the flights' data-parallel stage on a high-priority stream — parity of the batch paths, then A/B on the quick bench
ulimit -c 0
mkdir -p gpurun_out/r5aa
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_post_stages.py -x -q -m gpu -k "batch or flight or kinds or sparse or writer or composed or config" 2>&1 | tail -3
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("bench value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"], "h2d", c.get("h2d_included_MPps"))
except Exception as e:
    print("bench failed", e); print(open(sys.argv[1].replace(".json", "_err.txt")).read()[-1500:])
PY
}
run() { tag=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" 2>gpurun_out/r5aa/bench_${tag}_err.txt | tail -1 > gpurun_out/r5aa/bench_$tag.json; echo $tag; show gpurun_out/r5aa/bench_$tag.json; }
run hi --distinct 0 --steps 12 --warmup 3
JXLAMD_REST_PRIORITY=0 run off --distinct 0 --steps 12 --warmup 3
run hi2 --distinct 0 --steps 12 --warmup 3
JXLAMD_REST_PRIORITY=0 run off2 --distinct 0 --steps 12 --warmup 3
