#!/usr/bin/env python3
"""One decoder context, one flight of N 4K frames (default 128): LF-stage phase stamps of frame 0's LF groups, to compare a stream's
time inside a full flight with its time in a single decode (tools/prof_decode.py)."""
import os, sys, time
import ctypes as C, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jxl_coder_amd as J
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dec = J.JxlDecoder(0)
datas = [open(os.path.join(ROOT, f"bench_data/syn4k_q90_seed{i % 8}.jxl"), "rb").read() for i in range(n)]
outs = [torch.empty(3840 * 2160 * 4, dtype=torch.uint8, device="cuda") for _ in range(n)]
for rep in range(2):
    t = time.time(); dec.decode_batch_to_device(datas, [o.data_ptr() for o in outs], [o.numel() for o in outs]); dt = time.time() - t
    print("flight of %d: %.1f ms wall" % (n, dt * 1e3), dec.last_timing())
L = J.api.lib()
L.jxlamd_debug_lf_phases_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
tt = np.zeros((n, 4, 8), np.uint64)
for f in range(n):
    L.jxlamd_debug_lf_phases_frame(dec._h, f, 4, tt[f].ctypes.data)
start = tt[:, :, 0].min()
tot = (tt[:, :, 6].astype(np.int64) - tt[:, :, 0].astype(np.int64)) / 1e5
coef = (tt[:, :, 2].astype(np.int64) - tt[:, :, 1].astype(np.int64)) / 1e5
begin = (tt[:, :, 0].astype(np.int64) - int(start)) / 1e5
print("per-stream total ms (group 0): min %.1f max %.1f; LF coeffs min %.1f max %.1f; latest start %.1f ms; latest end %.1f ms" % (
    tot[:, 0].min(), tot[:, 0].max(), coef[:, 0].min(), coef[:, 0].max(), begin.max(), ((tt[:, :, 6].astype(np.int64) - int(start)) / 1e5).max()))
meta = (tt[:, :, 4].astype(np.int64) - tt[:, :, 3].astype(np.int64)) / 1e5
place = (tt[:, :, 5].astype(np.int64) - tt[:, :, 4].astype(np.int64)) / 1e5
for f in range(min(n, 8)):
    print(" frame", f, "g0 coeffs %.1f meta %.1f place %.1f | g1 coeffs %.1f meta %.1f place %.1f | bytes %d" % (coef[f, 0], meta[f, 0], place[f, 0], coef[f, 1], meta[f, 1], place[f, 1], len(datas[f])))
for f in range(min(n, 8)):
    print(" frame", f, "g0 total %.1f coeffs %.1f start %.1f | g1 total %.1f start %.1f" % (tot[f, 0], coef[f, 0], begin[f, 0], tot[f, 1], begin[f, 1]))
L.jxlamd_debug_lf_phases.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
t = np.zeros((4, 8), np.uint64)
L.jxlamd_debug_lf_phases(dec._h, 4, t.ctypes.data)
names = ["open+stage", "LF coeffs", "meta open+stage", "meta decode", "place", "epilogue"]
for g in range(4):
    d = (t[g, 1:7].astype(np.int64) - t[g, 0:6].astype(np.int64)) / 1e5
    wall_ms = float(int(t[g, 6]) - int(t[g, 0])) / 1e5
    print("frame 0 lf group", g, {k: round(float(v), 2) for k, v in zip(names, d)}, "ms; shader clock %.0f MHz" % (float(t[g, 7]) / max(wall_ms, 1e-9) / 1e3))
