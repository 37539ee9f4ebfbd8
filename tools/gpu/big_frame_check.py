#!/usr/bin/env python3
"""One-off robustness check on the GPU box: a large multi-LF-group frame (default 8192x6144 = 50 MP, 12 LF groups, 768 groups)
encoded with the reference's encoder (oracle/_ref, checker), decoded by the MI355X path and by the reference; prints parity + timings."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import jxl_ref, synth
import jxl_coder_amd as J
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 6144)
img = synth.photo_like(w, h, seed=3)
t = time.time(); data = jxl_ref.encode(img, effort=7, distance=1.0); print("encoded", len(data), "bytes in %.1f s" % (time.time() - t))
t = time.time(); ref = jxl_ref.decode(data, threads=0 if False else 64, allow16=True)[0]; t_ref = time.time() - t
dec = J.JxlDecoder(0)
for _ in range(2):
    t = time.time(); out, info = dec.decode_one_shot(data); t_gpu = time.time() - t
d = np.abs(out.astype(np.int16) - ref.astype(np.int16))
print("shape", out.shape, "max|diff|", int(d.max()), "mean %.4f" % d.mean(), "| GPU %.0f ms (%s) | reference CPU %.0f ms" % (t_gpu * 1e3, {k: round(v, 1) for k, v in dec.last_timing().items()}, t_ref * 1e3))
