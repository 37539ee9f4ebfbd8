# a default-settings lossy RGBA frame at 4K (VarDCT colour + squeezed, quantised alpha) and a lossless RGBA 1080p e7 frame: wall time of five decodes, A (tools/gpu/ab/libjxlamd_A.so) vs the in-tree build
ulimit -c 0; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python - <<'PY'
import sys; sys.path[:0] = ['.', 'oracle', 'tools']
import jxl_ref, synth
open('/tmp/rgba4k_d1.jxl', 'wb').write(jxl_ref.encode(synth.photo_like(3840, 2160, seed=4, channels=4), effort=7, distance=1.0))
open('/tmp/rgba1080_lossless_e7.jxl', 'wb').write(jxl_ref.encode(synth.photo_like(1920, 1080, seed=4, channels=4), lossless=True, effort=7))
PY
cp jxl_coder_amd/libjxlamd.so /tmp/B.so
for v in A B A B; do
  if [ $v = A ]; then cp tools/gpu/ab/libjxlamd_A.so jxl_coder_amd/libjxlamd.so; else cp /tmp/B.so jxl_coder_amd/libjxlamd.so; fi
  for f in /tmp/rgba4k_d1.jxl /tmp/rgba1080_lossless_e7.jxl; do
    echo "[rgba] $v $(basename $f): $(JXLAMD_PROF_FILE=$f python tools/prof_decode.py 5 2>&1 | grep '4k ' | tail -2 | tr '\n' ' ' | cut -c1-330)"
  done
done
cp /tmp/B.so jxl_coder_amd/libjxlamd.so
md5sum /tmp/rgba4k_d1.jxl
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alpha or previous_channel or rgba or squeeze or lossless" 2>&1 | tail -2
