# round 5: variants of the general lock-step loop (tools/build_variant.sh): interior / edge bodies ("split"), sample store once per row ("rowstore") — stage time of the
# PassGroup + alpha stage of the default lossy RGBA 4K frame, a lossless RGBA 4K e7 / e3 frame; md5 of the decoded pixels (must agree across variants)
ulimit -c 0
python - <<'PY'
import sys; sys.path[:0] = ['.', 'oracle', 'tools']
import jxl_ref, synth
img = synth.photo_like(3840, 2160, seed=4, channels=4)
open('/tmp/rgba4k_d1.jxl', 'wb').write(jxl_ref.encode(img, effort=7, distance=1.0))
open('/tmp/rgba4k_ll7.jxl', 'wb').write(jxl_ref.encode(img[:1080, :1920].copy(), lossless=True, effort=7))
open('/tmp/rgba4k_ll3.jxl', 'wb').write(jxl_ref.encode(img, lossless=True, effort=3))
PY
cat > /tmp/ab.py <<'PY'
import os, sys, time, hashlib; sys.path.insert(0, '.')
import jxl_coder_amd as J
dec = J.JxlDecoder(0)
for f in ('/tmp/rgba4k_d1.jxl', '/tmp/rgba4k_ll7.jxl', '/tmp/rgba4k_ll3.jxl'):
    data = open(f, 'rb').read(); best = None
    for i in range(4):
        out, info = dec.decode_one_shot(data); t = dec.last_timing()
        tot = t.get('device_total_ms', 0)
        if best is None or tot < best[0]: best = (tot, t)
    print(os.path.basename(f), hashlib.md5(out.tobytes()).hexdigest()[:12], 'total %.1f' % best[0], {k: round(v, 1) for k, v in best[1].items() if k.endswith('_ms') and v > 0.5})
PY
for v in ""; do
  lib=jxl_coder_amd/libjxlamd$v.so
  [ -f $lib ] || continue
  echo "== $lib"; JXLAMD_LIB=$PWD/$lib timeout 300 python /tmp/ab.py 2>&1 | tail -4
done
