# round 5, after the retry fix (a flight whose LF stage stopped for a larger pool is repeated whatever its later stages flagged): the repro, the mixed-content evidence, the -m gpu suite
ulimit -c 0
O=gpurun_out/ai; mkdir -p $O
FAST=1 timeout 600 python tools/gpu/mixed_repro.py 64 2>&1 | grep -v amdgpu.ids | grep -v "max diff vs reference -1" > $O/repro.txt; grep -c "differ from their single decode: \[\]" $O/repro.txt; grep -v "differ from their single decode: \[\]" $O/repro.txt | tail -5
bash tools/gpu/run_profiles_mixed.sh > $O/mixed.txt 2>&1; tail -32 $O/mixed.txt | cut -c1-220
( time timeout 1300 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
