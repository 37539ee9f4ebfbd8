ulimit -c 0
mkdir -p gpurun_out/diag
for v in "" 14 3; do
  lib=jxl_coder_amd/libjxlamd${v:+_abl$v}.so
  echo "== mask ${v:-0}"
  JXLAMD_BENCH_CLOCKS=1 JXLAMD_LIB=$PWD/$lib timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 8 --warmup 2 2>gpurun_out/diag/err_${v:-0}.txt | tail -1 > gpurun_out/diag/b_${v:-0}.json
  grep clocks gpurun_out/diag/err_${v:-0}.txt
  python -c "
import json; d=json.load(open('gpurun_out/diag/b_${v:-0}.json')); print('value', d['value'], d['roofline']['stage_ms_per_flight'])"
done
