# round 5: the bit reader's look-ahead word through an unconditional (clamped) load — parity, then the frames whose loops read bits through it
ulimit -c 0
mkdir -p gpurun_out/r5ab
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_vectors or block_form or kinds or rgba or lossless or batch_equals or config3 or previous" 2>&1 | tail -3
python tools/gpu/which_general.py 2>&1 | tail -3
bash tools/gpu/run_rgba4k_prof.sh 2>&1 | grep "4k " | sed -n '2p;5p'
timeout 300 python tools/prof_decode.py 3 2>&1 | grep "4k " | tail -1
timeout 600 python bench.py --workload mixed --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
timeout 600 python bench.py --no-cpu-baseline --distinct 0 --steps 12 --warmup 3 2>/dev/null | tail -1 | cut -c1-260
