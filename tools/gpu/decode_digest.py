"""Digest of the pixels of every fixture, for comparing two builds of the library bit for bit on one GPU box:

    JXLAMD_LIB=tools/gpu/ab/libjxlamd_A.so python tools/gpu/decode_digest.py > a.txt
    python tools/gpu/decode_digest.py > b.txt ; diff a.txt b.txt

(one process per build: the library is bound at import).  Files that a build refuses print their error text instead of a digest."""
import glob, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import jxl_coder_amd as J

files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.jxl")) + glob.glob(os.path.join(ROOT, "bench_data", "*.jxl")))
for f in files:
    data = open(f, "rb").read()
    try:
        img = J.JxlCoder.decode(data)
        a = np.asarray(img.pixels if hasattr(img, "pixels") else img)
        print(os.path.basename(f), a.shape, a.dtype, hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest())
    except Exception as e:      # noqa: BLE001 — the text is the result
        print(os.path.basename(f), "ERROR", str(e)[:100])
