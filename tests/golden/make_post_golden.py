#!/usr/bin/env python3
"""Generates tests/golden/post_golden.npz: outputs of the REFERENCE's own post-decode stages (oracle/_ref/libref_post.so, built
from /root/reference/jxlcoder/src/main/cpp/{imagebit,colorspaces}/*.cpp by oracle/ref_post/Makefile) on seeded inputs.
Run in the build container (needs the reference sources once, to build the library); the .npz is data, not source."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.path.join(ROOT, "oracle/_ref/libref_post.so"))
L.refpost_color_matrix.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p]
rng = np.random.default_rng(20260928)
h, w = 9, 53
p8 = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
p16 = rng.integers(0, 65536, (h, w, 4), dtype=np.uint16)
p8[2, 5, :3] = 0; p8[4, 0, :3] = 0; p16[1, 9, :3] = 0; p8[6, :, 3] = 0; p16[7, :, 3] = 65535
vp = lambda a: C.c_void_p(a.ctypes.data)
out = {"p8": p8, "p16": p16}
a = p8.copy(); L.refpost_associate8(vp(a), w * 4, w, h); out["associate8"] = a
a = p16.copy(); L.refpost_associate16(vp(a), w * 8, w, h, 16); out["associate16"] = a
d = np.zeros((h, w, 4), np.uint16); L.refpost_u16_to_f16(vp(p16), w * 8, vp(d), w * 8, w, h, 16); out["u16_to_f16"] = d
for att in (0, 1):
    d = np.zeros((h, w, 4), np.uint16); L.refpost_rgba8_to_f16(vp(p8), w * 4, vp(d), w * 8, w, h, att); out[f"rgba8_to_f16_{att}"] = d
    d = np.zeros((h, w), np.uint16); L.refpost_rgba8_to_565(vp(p8), w * 4, vp(d), w * 2, w, h, att); out[f"rgba8_to_565_{att}"] = d
    d = np.zeros((h, w), np.uint32); L.refpost_rgba8_to_1010102(vp(p8), w * 4, vp(d), w * 4, w, h, att); out[f"rgba8_to_1010102_{att}"] = d
d = np.zeros((h, w, 4), np.uint8); L.refpost_rgba16_to_8(vp(p16), w * 8, vp(d), w * 4, w, h, 16); out["rgba16_to_8"] = d
d = np.zeros((h, w), np.uint16); L.refpost_rgba16_to_565(vp(p16), w * 8, vp(d), w * 2, w, h, 16); out["rgba16_to_565"] = d
d = np.zeros((h, w), np.uint32); L.refpost_rgba16_to_1010102(vp(p16), w * 8, vp(d), w * 4, w, h, 16); out["rgba16_to_1010102"] = d
xy = (C.c_double * 8)(0.64, 0.33, 0.21, 0.71, 0.15, 0.06, 0.3127, 0.329)
out["custom_xy"] = np.array(list(xy))
for prim, tf, it in [(9, 16, 10000.0), (9, 18, 1000.0), (1, 13, 255.0), (11, 13, 255.0), (1, 1, 255.0), (2, 65535, 255.0), (11, 17, 255.0)]:
    for src, is16 in ((p8, 0), (p16, 1)):
        m = np.zeros(9, np.float32); a = src.copy()
        L.refpost_color_matrix(a.ctypes.data, w * (8 if is16 else 4), w, h, is16, 16 if is16 else 8, prim, tf, xy, it, m.ctypes.data)
        out[f"cm_{prim}_{tf}_{16 if is16 else 8}"] = a
        out[f"cm_matrix_{prim}"] = m
np.savez_compressed(os.path.join(ROOT, "tests/golden/post_golden.npz"), **out)
print("wrote", len(out), "arrays")
