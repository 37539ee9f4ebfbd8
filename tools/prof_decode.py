#!/usr/bin/env python3
"""Decode the 4K bench frame a few times (for rocprofv3 runs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jxl_coder_amd as J
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dec = J.JxlDecoder(0)
data = open(os.path.join(ROOT, os.environ.get("JXLAMD_PROF_FILE", "bench_data/syn4k_q90_seed0.jxl")), "rb").read()      # (an absolute JXLAMD_PROF_FILE wins: os.path.join)
for i in range(n):
    t = time.time()
    try:
        out, info = dec.decode_one_shot(data)
    except Exception as e:      # timing experiments (tools/build_variant.sh) decode garbage on purpose: the phase stamps below still count
        print("decode failed:", e)
    dt = time.time() - t
    print("4k %.1f ms wall" % (dt * 1e3), dec.last_timing())

import ctypes as C, numpy as np
L = J.api.lib()
L.jxlamd_debug_lf_phases.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
t = np.zeros((4, 8), np.uint64)
L.jxlamd_debug_lf_phases(dec._h, 4, t.ctypes.data)
names = ["open+stage", "LF coeffs", "meta open+stage", "meta decode", "place", "epilogue"]
for g in range(4):
    d = (t[g, 1:7].astype(np.int64) - t[g, 0:6].astype(np.int64)) / 1e5
    wall_ms = float(int(t[g, 6]) - int(t[g, 0])) / 1e5
    print("lf group", g, {n: round(float(v), 2) for n, v in zip(names, d)}, "ms; shader clock %.0f MHz" % (float(t[g, 7]) / max(wall_ms, 1e-9) / 1e3))
