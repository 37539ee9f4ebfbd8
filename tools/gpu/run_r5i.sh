# round 5: more contexts now that the HF-phase memory is small (sparse lists, 3-plane sets): contexts x pool sharing
ulimit -c 0
mkdir -p gpurun_out/r5i
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("bench value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"], "ctx", c["decoder_contexts"], "share", c["contexts_per_pool_set"], "P", c["frames_in_flight"], "h2d", c.get("h2d_included_MPps"))
except Exception as e:
    print("bench failed", e); print(open(sys.argv[1].replace(".json", "_err.txt")).read()[-600:])
PY
}
run() { tag=$1; shift; timeout 150 python bench.py --no-cpu-baseline --distinct 0 --steps 12 --warmup 3 "$@" 2>gpurun_out/r5i/bench_${tag}_err.txt | tail -1 > gpurun_out/r5i/bench_$tag.json; echo $tag; show gpurun_out/r5i/bench_$tag.json; }
run c16
run c24 --contexts 24
run c32s2 --contexts 32 --share 2
run c24s2 --contexts 24 --share 2
run c20x48 --contexts 20 --inflight 48
run c16b
