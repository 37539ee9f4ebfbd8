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
