#!/usr/bin/env python3
"""One-off check on the GPU box: the file kinds of round 4's last part at 4K size (many groups, several LF groups) — encoded with the reference's encoder
(oracle/_ref, checker), decoded by the MI355X path and by the reference; prints parity per file."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import jxl_ref, synth
import jxl_coder_amd as J

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
rgb = synth.photo_like(W, H, seed=3)
rgba = synth.photo_like(W, H, seed=4, channels=4)
grain = np.clip(rgb.astype(int) + np.random.default_rng(7).normal(0, 6, rgb.shape), 0, 255).astype(np.uint8)
shot = synth.screenshot(W // 2, H // 2, seed=2)
f16 = (rgba.astype(np.float32) / 255).astype(np.float16)
CASES = [
    ("cjxl -p on RGBA (qprogressive AC + squeezed alpha)", rgba, dict(distance=1.0, extra=((18, 1), (16, 1)))),
    ("progressive DC 2 + progressive AC", rgb, dict(distance=2.0, extra=((19, 2), (17, 1)))),
    ("noise + 2x upsampling", grain, dict(distance=2.0, extra=((6, 1), (2, 2)))),
    ("RGBA d12 + noise", np.dstack([grain, rgba[..., 3]]), dict(distance=12.0, extra=((6, 1),))),
    ("lossy palette screenshot", shot, dict(lossless=True, effort=7, extra=((23, 1),))),
    ("lossless e7 previous-channel properties (half size)", rgb[: H // 2, : W // 2], dict(lossless=True, effort=7, extra=((29, 3),))),
    ("lossless RGBA responsive (28 group channels)", rgba[: H // 2], dict(lossless=True, effort=3, extra=((16, 1),))),
    ("float16 RGBA lossless e3 (half size)", f16[: H // 2, : W // 2], dict(lossless=True, effort=3)),
    ("float16 RGBA d1", f16, dict(distance=1.0)),
]
dec = J.JxlDecoder(0)
bad = 0
for name, img, kw in CASES:
    t = time.time(); data = jxl_ref.encode(img, **kw); t_enc = time.time() - t
    ref = jxl_ref.decode(data, threads=64, allow16=True)[0]
    try:
        t = time.time(); out, info = dec.decode_one_shot(data); t_gpu = time.time() - t
    except Exception as e:      # noqa: BLE001
        print(f"{name:56s} {len(data):9d} B  FAILED {e}"); bad += 1
        continue
    d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    lossless = kw.get("lossless") and not any(k == 23 for k, _ in kw.get("extra", ()))
    tol = 0 if kw.get("lossless") else (1 if out.dtype == np.uint8 else 256)
    ok = d.max() <= tol
    bad += not ok
    print(f"{name:56s} {len(data):9d} B  {out.shape[1]}x{out.shape[0]} {out.dtype}  max {int(d.max())} mean {d.mean():.4f}  alpha max {int(d[..., 3].max())}  GPU {t_gpu * 1e3:.0f} ms  {'ok' if ok else 'OUT OF TOLERANCE'}", flush=True)
print("files out of tolerance:", bad)
