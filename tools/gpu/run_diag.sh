ulimit -c 0
mkdir -p gpurun_out/diag
timeout 900 python -m pytest tests/test_post_stages.py -x -q -m gpu 2>&1 | grep -E "^E|passed|failed" | head -12
for rep in 1 2; do for v in "" pool30 noprio; do
  lib=jxl_coder_amd/libjxlamd${v:+_$v}.so
  for m in full lfonly; do
    if [ $m = lfonly ]; then continue; fi
    JXLAMD_LIB=$PWD/$lib timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${v:-default}', 'value', d['value'], d['roofline']['stage_ms_per_flight'])"
  done
done; done
