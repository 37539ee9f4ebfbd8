#!/bin/bash
# JXLAMD_ENTROPY_CUS sweep: entropy stages of every context on the first N compute units, data-parallel stages on the others
mkdir -p gpurun_out
out=gpurun_out/cusplit.txt; : > $out
for n in 0 96 128 160 64 192; do
  for q in 16 32; do
    [ $n = 0 ] && [ $q = 32 ] && continue
    echo "== JXLAMD_ENTROPY_CUS=$n GPU_MAX_HW_QUEUES=$q" >> $out
    JXLAMD_ENTROPY_CUS=$n GPU_MAX_HW_QUEUES=$q timeout 240 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>>gpurun_out/cusplit.err | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['value'], d['ms_per_step'], d['roofline'].get('stage_ms_per_flight'))
" >> $out
  done
done
cat $out
