#!/usr/bin/env python3
"""Seeded corruption fuzz of the MI355X decode path (one-off robustness run, not part of pytest): every variant must either decode
or raise a JXL exception; afterwards the untouched file must still decode to the very same pixels (context survives)."""
import os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_case
import jxl_coder_amd as J
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rnd = random.Random(seed)
dec = J.JxlDecoder(0)
dec_ok = rej = 0
t0 = time.time()
for name in ("v256_e7", "v264x520_e7", "l200x120_e7", "va300x520_e7", "v64_hard_e7", "l700x500_e7", "asset_first_jxl", "l530x300_e1", "v300x300_e7_d3",
             "j420_200x136", "j444_200x136", "jgrey_160x120", "an_blend_lossless", "an_modes_d2_e5", "u96x64_lf_frame", "vlf600x410_e7", "vs400x300_e7_d1", "vu400x300_e7_d10",
             # round 4, last part: delta palettes, Modular channels spread over passes, two levels of LF frames, noise on an upsampled frame, previous-channel properties, upsampled animation layers
             "lpl400x300_e7_nopatch", "lpl200x136_e7_photo", "vapr400x300_e7", "vaqr520x300_e7", "vlfq600x410_e7", "vlf2_600x410_e7_d2", "vnu523x267_e7_d12", "lpc200x136_e7_prev3", "an_blend_d12_e7",      # round 4: JPEGs, animations, LF frames, patches, upsampling
             # round 5: the writer's files (splines, DCT128 / 256, custom upsampling weights, preview, dequant encodings, 6 / 11 passes) and — made here by the reference's
             # encoder when it travelled — an RGBA photograph whose alpha streams run from the block form of a 459-leaf tree
             "w_spline_b", "w_dct_mix_a", "w_dct256", "w_up4_custom", "w_preview", "w_dequant_a", "w_dequant_b", "w_passes6", "w_passes11", "FUZZ_LIVE_RGBA"):
    if name == "FUZZ_LIVE_RGBA":
        sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
        import jxl_ref, synth
        if not jxl_ref.available():
            continue
        d0 = jxl_ref.encode(synth.photo_like(700, 523, seed=2002, channels=4), effort=7, distance=1.0, threads=0)
    else:
        d0 = open(os.path.join(ROOT, "tests", "golden", name + ".jxl"), "rb").read()
    ref = dec.decode_one_shot(d0)[0]
    for it in range(n):
        d = bytearray(d0)
        mode = rnd.randrange(4)
        if mode == 0:
            for _ in range(rnd.randrange(1, 4)): d[rnd.randrange(len(d))] ^= 1 << rnd.randrange(8)
        elif mode == 1:
            for _ in range(rnd.randrange(1, 8)): d[rnd.randrange(len(d))] = rnd.randrange(256)
        elif mode == 2:
            d = d[:rnd.randrange(1, len(d))]
        else:
            a = rnd.randrange(len(d)); b = min(len(d), a + rnd.randrange(1, 64)); d[a:b] = bytes(rnd.randrange(256) for _ in range(b - a))
        try:
            dec.decode_one_shot(bytes(d)); dec_ok += 1
        except (J.InvalidJXLException, J.UnsupportedJXLFeature, J.InvalidImageSizeException, ValueError):
            rej += 1
    again = dec.decode_one_shot(d0)[0]
    assert np.array_equal(again, ref), name
print("fuzz seed", seed, ": decoded", dec_ok, "rejected", rej, "in %.1f s; context intact" % (time.time() - t0))
