import os, sys, time
import numpy as np
sys.path[:0] = ['/root/repo', '/root/repo/oracle', '/root/repo/tools']
import jxl_ref, synth
import jxl_coder_amd as J
dec = J.JxlDecoder(0)
for (w, h, seed) in ((1920, 1080, 2), (700, 500, 5)):
    shot = synth.screenshot(w, h, seed=seed)
    data = jxl_ref.encode(shot, lossless=True, effort=7, extra=((23, 1),))
    ref = jxl_ref.decode(data, threads=64)[0]
    for i in range(3):
        t = time.time(); out, info = dec.decode_one_shot(data); dt = time.time() - t
    print("lossy palette screenshot %dx%d: %d bytes, GPU %.1f ms, equal to the reference: %s" % (w, h, len(data), dt * 1e3, np.array_equal(out, ref)), dec.last_timing())
