# BASELINE configs[4] as a reproducible number: the same command several times, steps sized so that every context runs >= 4 flights
ulimit -c 0; mkdir -p gpurun_out/c5
for r in 1 2 3 4; do
  timeout 900 python bench.py --workload c5 --no-cpu-baseline --steps ${STEPS:-32} --warmup 8 2>gpurun_out/c5/err_$r.txt | tail -1 > gpurun_out/c5/c5_$r.json
  python -c "import json; d=json.load(open('gpurun_out/c5/c5_$r.json')); print('[c5] run $r value', d['value'], 'ms/step', d['ms_per_step'], d['roofline']['stage_ms_per_flight'], d['config'].get('frames_in_flight'), d['config'].get('decoder_contexts'))" || tail -5 gpurun_out/c5/err_$r.txt
done
