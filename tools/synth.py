"""Seeded synthetic 'photo-like' images: low-frequency sinusoid gradients per channel +
band-limited texture + ~2 % noise. Pure numpy, deterministic for a given (w, h, seed)."""
import numpy as np


def photo_like(w, h, seed=0, bits=8, channels=3, hard=False):
    rng = np.random.Generator(np.random.PCG64(0x9E3779B97F4A7C15 ^ seed))
    maxv = (1 << bits) - 1
    if hard:
        img = rng.integers(0, maxv + 1, size=(h, w, channels))
        return img.astype(np.uint16 if bits > 8 else np.uint8)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.zeros((h, w, channels), np.float32)
    for c in range(channels):
        acc = np.full((h, w), 0.5, np.float32)
        for _ in range(4):  # low-frequency gradients
            fx, fy = rng.uniform(-2.5, 2.5, 2) / max(w, h) * 2 * np.pi
            acc += rng.uniform(0.05, 0.18) * np.sin(fx * x + fy * y + rng.uniform(0, 6.28)).astype(np.float32)
        for _ in range(6):  # band-limited texture
            fx, fy = rng.uniform(-0.9, 0.9, 2)
            acc += rng.uniform(0.01, 0.05) * np.sin(fx * x + fy * y + rng.uniform(0, 6.28)).astype(np.float32)
        # a few hard edges (rectangles) so that EPF / large transforms get exercised
        for _ in range(3):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            x1, y1 = x0 + int(rng.integers(8, max(9, w // 3))), y0 + int(rng.integers(8, max(9, h // 3)))
            acc[y0:y1, x0:x1] += rng.uniform(-0.2, 0.2)
        acc += rng.uniform(-0.02, 0.02, size=(h, w)).astype(np.float32)
        out[..., c] = acc
    if channels == 4:
        out[..., 3] = 0.5 + 0.5 * np.sin(x * 0.05) * np.cos(y * 0.04)
    out = np.clip(out, 0, 1) * maxv
    return np.round(out).astype(np.uint16 if bits > 8 else np.uint8)


# ---- non-photographic content (what the reference's encoder compresses with palettes / patches instead of plain VarDCT or WP-predicted Modular)
_GLYPHS = ["01110100011000110001111111000110001", "11110100011111010001100011000111110", "01111100001000010000100001000001111",
           "11110100011000110001100011000111110", "11111100001111010000100001000011111", "10001100011111110001100011000110001",
           "00100001000010000100001000010000100", "10001110011010110011100011000110001", "01110100011000110001100011000101110"]


def screenshot(w, h, seed=0, channels=3):
    """UI-like image: flat panels in a handful of colours, 1-px borders, rows of repeated 5x7 'glyphs' (text)."""
    rng = np.random.Generator(np.random.PCG64(0xC0FFEE ^ seed))
    pal = np.array([[245, 245, 245], [32, 32, 36], [0, 120, 215], [255, 255, 255], [200, 60, 50], [90, 90, 96], [230, 200, 40]], np.uint8)
    img = np.empty((h, w, 3), np.uint8); img[:] = pal[0]
    for _ in range(6):
        x0, y0 = int(rng.integers(0, max(1, w - 16))), int(rng.integers(0, max(1, h - 16)))
        x1, y1 = min(w, x0 + int(rng.integers(16, max(17, w // 2)))), min(h, y0 + int(rng.integers(16, max(17, h // 2))))
        img[y0:y1, x0:x1] = pal[int(rng.integers(2, 7))]
        img[y0:y1, x0] = pal[5]; img[y0:y1, x1 - 1] = pal[5]; img[y0, x0:x1] = pal[5]; img[y1 - 1, x0:x1] = pal[5]
    g = [np.array([int(ch) for ch in s], np.uint8).reshape(7, 5).astype(bool) for s in _GLYPHS]
    for ty in range(6, h - 9, 11):
        n = int(rng.integers(w // 12, max(w // 12 + 1, w // 7)))
        x = 4
        for _ in range(n):
            k = int(rng.integers(0, len(g) + 2))
            if k < len(g) and x + 5 < w:
                blk = img[ty:ty + 7, x:x + 5]; blk[g[k]] = pal[1]
            x += 6
            if x + 6 >= w:
                break
    if channels == 4:
        a = np.full((h, w, 1), 255, np.uint8); a[: h // 4, : w // 4] = 0; a[h // 2:, w // 2:] = 128
        img = np.concatenate([img, a], axis=2)
    return img


def flat(w, h, colour=(37, 150, 190)):
    img = np.empty((h, w, 3), np.uint8); img[:] = np.array(colour, np.uint8)
    return img


def gradient(w, h):
    """plain horizontal/vertical ramp (at most 256 distinct colours per axis)"""
    y, x = np.mgrid[0:h, 0:w]
    r = (x * 255 // max(1, w - 1)).astype(np.uint8); g = (y * 255 // max(1, h - 1)).astype(np.uint8); b = ((x + y) * 255 // max(1, w + h - 2)).astype(np.uint8)
    return np.dstack([r, g, b])


def two_colour(w, h, seed=0, grey=True):
    rng = np.random.Generator(np.random.PCG64(0xB17 ^ seed))
    m = np.zeros((h, w), bool)
    for _ in range(12):
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        m[y0:y0 + int(rng.integers(4, max(5, h // 3))), x0:x0 + int(rng.integers(4, max(5, w // 3)))] ^= True
    v = np.where(m, 220, 30).astype(np.uint8)
    return v[..., None] if grey else np.dstack([v, v, v])


def many_colours(w, h, seed=0, n=1000):
    """n random colours in sorted runs: more palette entries than an 8-bit index holds"""
    rng = np.random.default_rng(seed)
    lut = rng.integers(0, 256, size=(n, 3)).astype(np.uint8)
    return lut[np.sort(rng.integers(0, n, size=(h, w)), axis=1)]


def hard_edged(w, h, seed=0, channels=3):
    """Hard-edged, saturated content: axis-aligned rectangles in saturated and near-saturated colours, a band of 1-px stripes, a corner
    of uniform noise.  What photographs-like fixtures do not have: pixels with one channel near 0 beside two near 1 right at an edge, where libjxl's inverse
    opsin matrix amplifies any 1e-4 difference upstream (the EPF's reciprocal) into several 8-bit codes.  channels: 1 (grey), 3, or 4 (RGB + a hard-edged alpha)."""
    rng = np.random.default_rng(0xED6E ^ seed)
    c = 1 if channels == 1 else 3
    sat = np.array([[255, 255, 0], [0, 255, 255], [255, 0, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [10, 210, 210], [240, 240, 20]], np.uint8)
    img = np.zeros((h, w, c), np.uint8)
    img[:] = rng.integers(0, 256, c)
    for _ in range(60):
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        x1, y1 = x0 + int(rng.integers(4, max(5, w // 4))), y0 + int(rng.integers(4, max(5, h // 4)))
        img[y0:y1, x0:x1] = sat[rng.integers(0, len(sat))][:c]
    img[h // 2: h // 2 + 40, ::4] = sat[1][:c]
    img[h // 2: h // 2 + 40, 1::4] = sat[3][:c]
    img[: h // 6, : w // 6] = rng.integers(0, 256, (h // 6, w // 6, c))
    if channels == 4:
        a = np.full((h, w, 1), 255, np.uint8)
        for _ in range(12):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            a[y0:y0 + int(rng.integers(4, max(5, h // 3))), x0:x0 + int(rng.integers(4, max(5, w // 3)))] = int(rng.choice([0, 64, 128, 200]))
        img = np.concatenate([img, a], axis=2)
    return img
