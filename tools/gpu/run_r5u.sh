ulimit -c 0
python - <<'PY'
import os, sys; sys.path[:0] = ['.', 'oracle', 'tools']
import jxl_ref, synth
open('/tmp/rgba4k_d1.jxl', 'wb').write(jxl_ref.encode(synth.photo_like(3840, 2160, seed=4, channels=4), effort=7, distance=1.0))
PY
for v in 4; do echo "JXLAMD_DEBUG_MOD=$v"; JXLAMD_DEBUG_MOD=$v JXLAMD_PROF_FILE=/tmp/rgba4k_d1.jxl timeout 300 python tools/prof_decode.py 3 2>&1 | grep "4k \|block-form" | tail -4; done
