# round 5: HBM accesses of the entropy loops as global loads / stores instead of FLAT ones (bit reader, block-form tree loop, flat PassGroup kernel)
ulimit -c 0
mkdir -p gpurun_out/r5t
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "block_form or golden_vectors or batch_equals or config3 or sparse or kinds or 4k_frame or jpeg" 2>&1 | tail -3
[ -f gpurun_out/rgba4k/rgba4k_d1.jxl ] && cp gpurun_out/rgba4k/rgba4k_d1.jxl /tmp/rgba4k_d1.jxl
JXLAMD_PROF_FILE=/tmp/rgba4k_d1.jxl timeout 300 python tools/prof_decode.py 3 2>&1 | grep "4k \|lf group 0" | tail -2
timeout 300 python tools/prof_decode.py 3 2>&1 | grep "4k \|lf group 0" | tail -2
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("bench value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"], "h2d", c.get("h2d_included_MPps"), "pool", c["lf_pool_bytes"], c["flights_repeated_for_lf_pool"])
except Exception as e:
    print("bench failed", e); print(open(sys.argv[1].replace(".json", "_err.txt")).read()[-1500:])
PY
}
run() { tag=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" 2>gpurun_out/r5t/bench_${tag}_err.txt | tail -1 > gpurun_out/r5t/bench_$tag.json; echo $tag; show gpurun_out/r5t/bench_$tag.json; }
run quick --distinct 0 --steps 12 --warmup 3
run quick2 --distinct 0 --steps 12 --warmup 3
