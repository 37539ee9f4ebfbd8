# round 5: removal switches on the block-form loop (measurement only, wrong pixels): 2 = no blocks behind block 0, 8 = no weighted predictor, 16 = no symbol read
# (no table load, no rANS step, no bits), 32 = no tree walk at all; RGBA 4K frame, PassGroup + alpha stage (k_pass_group 20.5 ms + k_mod_group)
ulimit -c 0
python - <<'PY'
import os, sys; sys.path[:0] = ['.', 'oracle', 'tools']
import jxl_ref, synth
open('/tmp/rgba4k_d1.jxl', 'wb').write(jxl_ref.encode(synth.photo_like(3840, 2160, seed=4, channels=4), effort=7, distance=1.0))
PY
for v in 0 2 8 16 32 24 56; do echo -n "JXLAMD_DEBUG_MOD=$v  "; JXLAMD_DEBUG_MOD=$v JXLAMD_PROF_FILE=/tmp/rgba4k_d1.jxl timeout 300 python tools/prof_decode.py 3 2>&1 | grep "4k " | tail -1 | sed 's/.*pass_groups_ms.: \([0-9.]*\).*/pass_groups_ms \1/'; done
