ulimit -c 0; mkdir -p gpurun_out/mixed
for st in 8 16 32 48 32 48; do
  timeout 900 python bench.py --workload mixed --no-cpu-baseline --steps $st 2>gpurun_out/mixed/err.txt | tail -1 > gpurun_out/mixed/steps_$st.json
  python -c "import json; d=json.load(open('gpurun_out/mixed/steps_$st.json')); print('[mixed] steps=$st value', d['value'], 'ms/step', d['ms_per_step'])" || tail -5 gpurun_out/mixed/err.txt
done
