#!/bin/bash
# JXLAMD_ENTROPY_PRIORITY=1: entropy stages on a highest-priority stream, data-parallel stages on a lowest-priority stream (no CU masks)
mkdir -p gpurun_out
out=gpurun_out/prio.txt; : > $out
for cfg in "0 16" "1 16" "1 32" "0 16"; do
  set -- $cfg
  echo "== JXLAMD_ENTROPY_PRIORITY=$1 GPU_MAX_HW_QUEUES=$2" >> $out
  JXLAMD_ENTROPY_PRIORITY=$1 GPU_MAX_HW_QUEUES=$2 timeout 240 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>>gpurun_out/prio.err | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['value'], d['ms_per_step'], d['roofline'].get('stage_ms_per_flight'))
" >> $out
done
JXLAMD_ENTROPY_PRIORITY=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "batch or flight or config3 or concurrent" 2>&1 | tail -2 >> $out
cat $out
