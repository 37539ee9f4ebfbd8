#!/usr/bin/env python3
"""BASELINE config 4 in miniature on ONE GPU: a large VarDCT frame (default 8192 x 8192 = 67 MP, 32 group rows = 4 LF-group rows, from
the reference's encoder) decoded whole and as N bands (default 4, LF-group aligned) with the halo exchange going through device buffers —
the bytes RCCL would carry.  Prints parity (bands == whole, whole vs the reference) and per-stage times of one band: with one band per GPU
those run concurrently, so max-over-bands is what an N-GPU node would take (plus two sub-3-MB sendrecv pairs per border)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import jxl_ref, synth
import jxl_coder_amd as J
from jxl_coder_amd.shard import DeviceBand, band_rows, HALO_LF, HALO_PIXELS
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 8192)
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
t = time.time(); data = jxl_ref.encode(synth.photo_like(w, h, seed=41), effort=7, distance=1.0); print("encoded %d bytes in %.1f s" % (len(data), time.time() - t))
t = time.time(); ref = jxl_ref.decode(data, threads=64)[0]; t_ref = time.time() - t
dec = J.JxlDecoder(0)
for _ in range(2):
    t = time.time(); whole, info = dec.decode_one_shot(data, size_guard=False); t_whole = time.time() - t
print("whole frame: %.0f ms on one GPU %s | reference CPU (64 threads) %.0f ms" % (t_whole * 1e3, {k: round(v, 1) for k, v in dec.last_timing().items()}, t_ref * 1e3))
d = np.abs(whole.astype(np.int16) - ref.astype(np.int16)); print("whole vs reference: max %d mean %.4f" % (d.max(), d.mean()))
rows = band_rows((h + 255) // 256, nb)
decs = [J.JxlDecoder(0) for _ in range(nb)]
for rep in range(2):
    stage = np.zeros((nb, 3)); bands = {}
    for b in range(nb):
        t = time.time(); bands[b] = DeviceBand(decs[b], data, rows[b], w, h, 4, "cuda:0"); stage[b, 0] = time.time() - t
    for b in range(nb - 1):
        bands[b + 1].import_(HALO_LF, 0, bands[b].export(HALO_LF, 1)); bands[b].import_(HALO_LF, 1, bands[b + 1].export(HALO_LF, 0))
    for b in range(nb):
        t = time.time(); decs[b].band_reconstruct(); stage[b, 1] = time.time() - t
    for b in range(nb - 1):
        bands[b + 1].import_(HALO_PIXELS, 0, bands[b].export(HALO_PIXELS, 1)); bands[b].import_(HALO_PIXELS, 1, bands[b + 1].export(HALO_PIXELS, 0))
    for b in range(nb):
        t = time.time(); decs[b].band_finish(); stage[b, 2] = time.time() - t
torch.cuda.synchronize()
img = np.concatenate([bands[b].out.cpu().numpy().reshape(-1, w, 4) for b in range(nb)])
print("bands == whole:", bool(np.array_equal(img, whole)))
print("per band (ms) begin[parse+LF] / reconstruct[smooth+PassGroup+IDCT] / finish[filters+writer]:")
for b in range(nb):
    print("  band %d rows %s: %.0f / %.0f / %.0f" % (b, rows[b], *(stage[b] * 1e3)))
print("one band per GPU: max-over-bands total %.0f ms  (whole frame on one GPU %.0f ms); halo messages: LF %d B, pixels %d B per border and direction"
      % (stage.sum(axis=1).max() * 1e3, t_whole * 1e3, decs[0].band_halo_bytes(HALO_LF) if False else (w // 8) * 14, 3 * 4 * (w // 8 * 8) * 3))
