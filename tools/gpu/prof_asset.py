#!/usr/bin/env python3
"""Decode a golden fixture repeatedly (for rocprofv3 runs on natural-image content)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import jxl_coder_amd as J
name = sys.argv[1] if len(sys.argv) > 1 else "asset_wide_gamut"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dec = J.JxlDecoder(0)
data = open(os.path.join(ROOT, "tests/golden", name + ".jxl"), "rb").read()
for i in range(n):
    t = time.time(); out, info = dec.decode_one_shot(data); dt = time.time() - t
    print(name, out.shape, "%.1f ms wall" % (dt * 1e3), {k: round(v, 3) for k, v in dec.last_timing().items()})
