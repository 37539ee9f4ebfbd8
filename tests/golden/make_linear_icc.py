#!/usr/bin/env python3
"""Fixtures for enum-coded LINEAR-light images (run in the build container): the reference does not handle the linear transfer function itself —
it asks libjxl for the data profile, which libjxl synthesises, and converts through Little CMS (interop/JxlDecoding.cpp:126-141,
JniDecoding.cpp:103-114).  Stored: the .jxl from the reference's encoder, the reference's decoded pixels (.npz) and the profile bytes its
libjxl returned (.icc) — the product's synthesised profile must equal them byte for byte."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import jxl_ref  # noqa: E402
import synth  # noqa: E402

CASES = {
    "vlin96x64_e3": (dict(seed=7), dict(distance=1.0, effort=3, transfer=8)),
    "vlin2100_96x64_e3": (dict(seed=7), dict(distance=1.0, effort=3, transfer=8, primaries=9)),
    "vlinp3_96x64_e3": (dict(seed=7), dict(distance=1.0, effort=3, transfer=8, primaries=11)),
    "llin96x64_e3": (dict(seed=7), dict(lossless=True, effort=3, transfer=8)),
    "vlingrey96x64_e3": (dict(seed=7, grey=True), dict(distance=1.0, effort=3, transfer=8)),
}
meta = json.load(open(os.path.join(HERE, "golden.json")))
for name, (sk, ek) in CASES.items():
    sk = dict(sk); grey = sk.pop("grey", False)
    img = synth.photo_like(96, 64, **sk)
    if grey:
        img = np.ascontiguousarray(img[..., :1])
    data = jxl_ref.encode(img, **ek)
    out, info, icc = jxl_ref.decode(data)
    open(os.path.join(HERE, name + ".jxl"), "wb").write(data)
    open(os.path.join(HERE, name + ".icc"), "wb").write(icc)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), rgba=out)
    info = {k: (v if isinstance(v, list) else float(v) if isinstance(v, float) else int(v)) for k, v in info.items()}
    meta[name] = dict(bytes=len(data), shape=list(out.shape), dtype=str(out.dtype), info=info, encode=ek, synth=dict(sk, grey=grey), icc_size=len(icc),
                      icc_sha256=hashlib.sha256(icc).hexdigest())
    print(name, len(data), len(icc), info.get("prefer_encoding"))
json.dump(meta, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
