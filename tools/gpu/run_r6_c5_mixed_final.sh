# config 5 and the mixed workload on the round's final code (two runs each)
ulimit -c 0; mkdir -p gpurun_out/c5m
for r in 1 2; do
  timeout 900 python bench.py --workload c5 --no-cpu-baseline --steps 32 --warmup 8 2>gpurun_out/c5m/c5_err_$r.txt | tail -1 > gpurun_out/c5m/c5_$r.json
  python -c "import json; d=json.load(open('gpurun_out/c5m/c5_$r.json')); print('[c5] run $r value', d['value'], 'ms/step', d['ms_per_step'])" || tail -5 gpurun_out/c5m/c5_err_$r.txt
  timeout 900 python bench.py --workload mixed --no-cpu-baseline --steps 48 --warmup 8 2>gpurun_out/c5m/mixed_err_$r.txt | tail -1 > gpurun_out/c5m/mixed_$r.json
  python -c "import json; d=json.load(open('gpurun_out/c5m/mixed_$r.json')); print('[mixed] run $r value', d['value'], 'ms/step', d['ms_per_step'])" || tail -5 gpurun_out/c5m/mixed_err_$r.txt
done
