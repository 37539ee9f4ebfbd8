# round 5: where does the block-form loop spend its 1.4 us per sample?  JXLAMD_DEBUG_MOD (measurement only): 1 = every symbol from cluster 0's tables
# (L1-hot: the alias lookups' latency share), 2 = block 0's exits taken as leaves (the later blocks' share)
ulimit -c 0
mkdir -p gpurun_out/r5s
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "jpeg_transcodes or block_form" 2>&1 | tail -3
[ -f gpurun_out/rgba4k/rgba4k_d1.jxl ] && cp gpurun_out/rgba4k/rgba4k_d1.jxl /tmp/rgba4k_d1.jxl
python - <<'PY'
import os, sys; sys.path[:0] = ['.', 'oracle', 'tools']
if not os.path.exists('/tmp/rgba4k_d1.jxl'):
    import jxl_ref, synth
    open('/tmp/rgba4k_d1.jxl', 'wb').write(jxl_ref.encode(synth.photo_like(3840, 2160, seed=4, channels=4), effort=7, distance=1.0))
PY
for v in 0 1 2 3; do echo "JXLAMD_DEBUG_MOD=$v"; JXLAMD_DEBUG_MOD=$v JXLAMD_PROF_FILE=/tmp/rgba4k_d1.jxl timeout 300 python tools/prof_decode.py 3 2>&1 | grep "4k " | tail -1; done
