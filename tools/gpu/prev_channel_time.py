"""cjxl -E files (MA-tree properties of previous channels) on the MI355X: time and parity against the reference run live."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")]
import jxl_ref, synth
import jxl_coder_amd as J
dec = J.JxlDecoder(0)
g = J.api.lib().jxlamd_debug_modular
g.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 2)]
for (w, h, seed, E) in ((1920, 1080, 3, 3), (700, 500, 5, 2), (640, 480, 6, 1)):
    img = synth.photo_like(w, h, seed=seed)
    data = jxl_ref.encode(img, lossless=True, effort=7, extra=((29, E),))
    ref = jxl_ref.decode(data, threads=64)[0]
    s0 = (C.c_uint64 * 2)(); g(dec._h, C.byref(s0))
    for i in range(2):
        t = time.time(); out, info = dec.decode_one_shot(data); dt = time.time() - t
    s1 = (C.c_uint64 * 2)(); g(dec._h, C.byref(s1))
    print("lossless e7 -E %d %dx%d: %d bytes, GPU %.1f ms, bit-exact: %s, serial streams %d, block-form channels %d" % (E, w, h, len(data), dt * 1e3, np.array_equal(out, ref), (s1[0] - s0[0]) // 2, (s1[1] - s0[1]) // 2))
