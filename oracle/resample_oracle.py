"""CPU restatement (numpy, float64) of the decodeSampled resampler — TEST INFRASTRUCTURE ONLY.

Geometry follows the reference exactly: resolve_dimensions (weaver/src/scale.rs:94-130) and the Fit / Fill / Resize window
(scale.rs:202-234; f64 arithmetic, Rust's round = half away from zero).  The filter arithmetic is `pic-scale 0.7.6`, an un-vendored
crate (weaver/Cargo.toml:6-9, Cargo.lock:132-135): PARITY UNPINNED for the sample values — this file restates the published
definitions of the ten filters (triangle, nearest, Keys cubic a=-0.5 / -0.75, Mitchell-Netravali and the other BC-splines, Lanczos3)
as a separable convolution with clamp-to-edge and kernel widening on minification, premultiplied alpha when the origin has alpha;
tests/test_resample.py holds the HIP kernels to it within +-1 LSB."""
import math
import numpy as np


def rust_round(v):
    return int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))


def geometry(w, h, new_w, new_h, mode):
    if new_w > 0 and new_h == -1:
        nw, nh = new_w, max(1, rust_round(h * (new_w / w)))
    elif new_w > 0 and new_h == -2:
        nw, nh = new_w, (max(1, rust_round(h * (new_w / w))) + 1) & ~1
    elif new_w == -1 and new_h > 0:
        nh, nw = new_h, max(1, rust_round(w * (new_h / h)))
    elif new_w == -2 and new_h > 0:
        nh, nw = new_h, (max(1, rust_round(w * (new_h / h))) + 1) & ~1
    else:
        nw, nh = max(1, new_w), max(1, new_h)
    if mode == 3:
        return nw, nh, 0, 0, nw, nh
    xf, yf = nw / w, nh / h
    sc = max(xf, yf) if mode == 2 else min(xf, yf)
    sw, sh = max(1, rust_round(w * sc)), max(1, rust_round(h * sc))
    cx, cy = max(0, (sw - nw) // 2 if sw >= nw else -((nw - sw) // 2)), max(0, (sh - nh) // 2 if sh >= nh else -((nh - sh) // 2))
    return sw, sh, cx, cy, min(nw, sw), min(nh, sh)


def _bc(x, B, C):
    x = np.abs(x)
    a = ((12 - 9 * B - 6 * C) * x ** 3 + (-18 + 12 * B + 6 * C) * x ** 2 + (6 - 2 * B)) / 6
    b = ((-B - 6 * C) * x ** 3 + (6 * B + 30 * C) * x ** 2 + (-12 * B - 48 * C) * x + (8 * B + 24 * C)) / 6
    return np.where(x < 1, a, np.where(x < 2, b, 0.0))


def _keys(x, a):
    x = np.abs(x)
    return np.where(x < 1, (a + 2) * x ** 3 - (a + 3) * x ** 2 + 1, np.where(x < 2, a * x ** 3 - 5 * a * x ** 2 + 8 * a * x - 4 * a, 0.0))


def _weight(f, x):
    if f == 1:
        return np.maximum(0.0, 1 - np.abs(x))
    if f == 3:
        return _keys(x, -0.5)
    if f == 4:
        return _bc(x, 1 / 3, 1 / 3)
    if f in (5, 9):
        return np.where(np.abs(x) < 3, np.sinc(x) * np.sinc(x / 3), 0.0)
    if f == 6:
        return _bc(x, 0.0, 0.5)
    if f == 7:
        return _bc(x, 0.0, 0.0)
    if f == 8:
        return _bc(x, 1.0, 0.0)
    return _keys(x, -0.75)


RADIUS = {1: 1.0, 2: 0.5, 3: 2.0, 4: 2.0, 5: 3.0, 6: 2.0, 7: 1.0, 8: 2.0, 9: 3.0, 10: 2.0}


def _matrix(in_len, out_full, crop0, out_len, f):
    """[out_len, in_len] resampling matrix (rows normalised)."""
    M = np.zeros((out_len, in_len))
    scale = np.float32(in_len) / np.float32(out_full)
    fscale = max(scale, np.float32(1.0))
    for o in range(out_len):
        center = np.float32((np.float32(o + crop0) + np.float32(0.5)) * scale)          # the device computes the tap window in f32
        if f == 2:
            j = min(max(int(math.floor(center)), 0), in_len - 1)
            M[o, j] = 1.0
            continue
        radius = np.float32(RADIUS[f]) * fscale
        j0, j1 = int(math.floor(np.float32(center - radius))), int(math.ceil(np.float32(center + radius)))
        js = np.arange(j0, j1)
        wg = _weight(f, (js + 0.5 - float(center)) / float(fscale))
        for j, g in zip(js, wg):
            M[o, min(max(j, 0), in_len - 1)] += g
        M[o] /= M[o].sum()
    return M


def rescale(px, depth, new_w, new_h, mode, sampler, premultiply):
    """px: [h, w, 4] u8 / u16 -> resampled array of the same dtype."""
    h, w = px.shape[:2]
    sw, sh, cx, cy, cw, ch = geometry(w, h, new_w, new_h, mode)
    maxv = float((1 << depth) - 1)
    v = px.astype(np.float64)
    if premultiply:
        v[..., :3] *= v[..., 3:4] / maxv
    Mx = _matrix(w, sw, cx, cw, sampler)
    My = _matrix(h, sh, cy, ch, sampler)
    t = np.einsum("ow,hwc->hoc", Mx, v)
    r = np.einsum("ph,hoc->poc", My, t)
    if premultiply:
        a = r[..., 3:4]
        r[..., :3] = np.where(a > 0, r[..., :3] * (maxv / np.where(a > 0, a, 1)), 0.0)
    return np.clip(np.rint(r), 0, maxv).astype(px.dtype)
