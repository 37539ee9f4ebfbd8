ulimit -c 0; mkdir -p gpurun_out/mixed
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "composed_frames or patch or round4_kinds or round3_kinds or concurrent_contexts" 2>&1 | tail -3
for v in 0 1 0 1; do for st in 16 8; do
  JXLAMD_REFS_ASYNC=$v timeout 900 python bench.py --workload mixed --no-cpu-baseline --steps $st 2>gpurun_out/mixed/err.txt | tail -1 > gpurun_out/mixed/m_${v}_$st.json
  python -c "import json; d=json.load(open('gpurun_out/mixed/m_${v}_$st.json')); print('[mixed] async=$v steps=$st value', d['value'], 'ms/step', d['ms_per_step'])" || tail -5 gpurun_out/mixed/err.txt
done; done
