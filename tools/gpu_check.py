#!/usr/bin/env python3
"""Quick on-GPU check: decode every tests/golden VarDCT fixture + the 4K bench frame through libjxlamd.so."""
import glob, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jxl_coder_amd as J
dec = J.JxlDecoder(0)
bad = 0
for f in sorted(glob.glob(os.path.join(ROOT, "tests/golden/v*.jxl"))):
    exp = np.load(f[:-4] + ".npz")["rgba"]
    data = open(f, "rb").read()
    try:
        out, info = dec.decode_one_shot(data)
    except Exception as e:
        print(os.path.basename(f), "ERR", type(e).__name__, e); bad += 1; continue
    d = np.abs(out.astype(int) - exp.astype(int))
    print(os.path.basename(f), out.shape, "max", d.max(), "mean %.4f" % d.mean(), dec.last_timing())
    bad += d.max() > (1 if exp.dtype == np.uint8 else 11500)
data = open(os.path.join(ROOT, "bench_data/syn4k_q90_seed0.jxl"), "rb").read()
for i in range(3):
    t = time.time(); out, info = dec.decode_one_shot(data); dt = time.time() - t
    print("4k", out.shape, "%.1f ms wall" % (dt * 1e3), dec.last_timing())
meta = json.load(open(os.path.join(ROOT, "tests/golden/golden.json")))["syn4k_q90_seed0"]
rs = [int(x) for x in out[::240].astype(np.int64).sum(axis=(1, 2))]
print("row-sum max rel dev vs reference:", max(abs(a - b) / b for a, b in zip(rs, meta["row_sums"])))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
try:
    import jxl_ref
    ref, _, _ = jxl_ref.decode(data)
    d = np.abs(out.astype(int) - ref.astype(int))
    print("4k vs reference libjxl: max", d.max(), "mean %.4f" % d.mean())
    bad += d.max() > 1
except Exception as e:
    print("reference unavailable:", e)
print("BAD" if bad else "ALL OK")
