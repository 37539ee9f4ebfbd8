for s in 0 1 3 5 6 "0,1,2,3,4,5,6,7"; do
  JXLAMD_BENCH_SEEDS=$s timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[seeds] $s value', d['value'], 'ms/step', d['ms_per_step'], d['roofline']['stage_ms_per_flight'], 'pool', d['config']['lf_pool_bytes'], 'bytes', d['config']['frame_bytes_mean'])"
done
