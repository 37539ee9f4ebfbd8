# A/B two builds of libjxlamd.so on the same box: tools/gpu/ab/libjxlamd_A.so vs the in-tree build (B)
ulimit -c 0
cp jxl_coder_amd/libjxlamd.so /tmp/B.so
ARGS="${BENCH_ARGS:---contexts 4 --steps 4096}"
for rep in 1 2; do for v in A B; do
  if [ $v = A ]; then cp tools/gpu/ab/libjxlamd_A.so jxl_coder_amd/libjxlamd.so; else cp /tmp/B.so jxl_coder_amd/libjxlamd.so; fi
  timeout 900 python bench.py --no-cpu-baseline $ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v value',d['value'],d['roofline']['stage_ms_per_flight'])"
done; done
cp /tmp/B.so jxl_coder_amd/libjxlamd.so
