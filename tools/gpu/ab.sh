# One parametrised A/B runner for the GPU box (replaces the per-experiment run_r5*.sh scripts).
#   VARIANTS="name1:ENV1=a,ENV2=b name2: name3:LIB=path/to/other/libjxlamd.so"   each variant = a set of environment variables and / or another build of the library
#   REPS=2  BENCH_ARGS="--steps 20 --warmup 5"  OUT=gpurun_out/ab   TRACE=1 (JXLAMD_TRACE_FLIGHT summaries)
# Variants alternate (A B A B ...) so that box drift shows up as spread inside a variant, not as a difference between them.
ulimit -c 0
OUT=${OUT:-gpurun_out/ab}; mkdir -p $OUT
REPS=${REPS:-2}
ARGS="${BENCH_ARGS:---steps 20 --warmup 5}"
cp jxl_coder_amd/libjxlamd.so /tmp/ab_base.so
for rep in $(seq 1 $REPS); do for v in $VARIANTS; do
  name=${v%%:*}; envs=${v#*:}
  lib=""; envline=""
  for kv in $(echo "$envs" | tr ',' ' '); do
    case $kv in LIB=*) lib=${kv#LIB=};; *) envline="$envline $kv";; esac
  done
  if [ -n "$lib" ]; then cp "$lib" jxl_coder_amd/libjxlamd.so; else cp /tmp/ab_base.so jxl_coder_amd/libjxlamd.so; fi
  env $envline ${TRACE:+JXLAMD_TRACE_FLIGHT=1} timeout 900 python bench.py --no-cpu-baseline $ARGS 2>$OUT/${name}_$rep.err | tail -1 > $OUT/${name}_$rep.json
  python - $OUT/${name}_$rep.json $name $rep <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print("[ab]", sys.argv[2], "rep", sys.argv[3], "value", d["value"], "ms/step", d["ms_per_step"], d["roofline"].get("stage_ms_per_flight"))
except Exception as e:
    print("[ab]", sys.argv[2], "FAILED", e)
PY
  if [ -n "$TRACE" ]; then python tools/gpu/flight_summary.py $OUT/${name}_$rep.err; fi
done; done
cp /tmp/ab_base.so jxl_coder_amd/libjxlamd.so
