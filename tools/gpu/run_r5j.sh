# round 5: writer-post through the column sweep (bit-identical to decode + post_fused), then the PMC passes (flight of 128; single 4K frame)
ulimit -c 0
timeout 1200 python -m pytest tests/test_post_stages.py tests/test_gpu_parity.py -x -q -m gpu -k "writer_post or config5 or pipeline or golden_vectors or bench_line" 2>&1 | tail -5
bash tools/gpu/run_pmc_batch.sh 2>&1 | tail -30
bash tools/gpu/run_pmc.sh 2>&1 | grep "FETCH\|WRITE" | tail -30
