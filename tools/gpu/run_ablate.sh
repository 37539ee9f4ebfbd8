# What each stage costs the 16-context mix: bench with a stage left out from the third flight of every context on
# (experiment builds: tools/build_variant.sh ablN decoder.hip -DJXL_ABLATE_MASK=N; 1 LF, 2 PassGroup, 4 reconstruction, 8 filters+writer)
ulimit -c 0
mkdir -p gpurun_out/ablate
for v in "" 1 2 3 4 8 12 13 14; do
  lib=jxl_coder_amd/libjxlamd${v:+_abl$v}.so
  JXLAMD_LIB=$PWD/$lib timeout 600 python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>gpurun_out/ablate/err_$v.txt | tail -1 > gpurun_out/ablate/abl_${v:-0}.json
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/ablate/abl_${v:-0}.json"))
    print("mask ${v:-0}: value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"])
except Exception as e:
    print("mask ${v:-0}: failed", e)
PY
done
