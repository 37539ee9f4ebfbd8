ulimit -c 0
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
JXLAMD_HF_SETS=16 JXLAMD_PLANE_SETS=2 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "batch" 2>&1 | tail -2
run() { timeout 900 python bench.py --no-cpu-baseline --steps $3 --inflight $1 --contexts $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight $1 ctx $2 value',d['value'],d['roofline']['stage_ms_per_flight'])"; }
run 128 8 4096
run 128 8 8192
run 256 8 8192
run 192 8 6144
run 384 6 9216
