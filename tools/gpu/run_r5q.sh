# round 5: block-form trees, properties through LDS: parity again, then the RGBA 4K kernel times
ulimit -c 0
mkdir -p gpurun_out/r5q
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "block_form or round3_kinds or round4 or rgba" 2>&1 | tail -4
bash tools/gpu/run_rgba4k_prof.sh 2>&1 | grep -v "^\"void\|fillBuffer\|copyBuffer" | head -12
