#!/usr/bin/env python3
"""One-off C5-shaped check on the GPU box (BASELINE config 5): 4K Rec.2100 PQ 16-bit VarDCT frame with EPF=3, encoded by the
reference's encoder; a flight of 64 decodes to RGBA16 in HBM, then per frame the API<34 colour pipeline (PQ -> Rec.2408 tone map ->
Rec.709 -> sRGB) and the RGBA_F16 reformat, all on the device.  Prints parity of one frame against the reference decoder (+ the
numpy post-stage oracle) and the throughput of the whole chain."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import jxl_ref, synth, post_oracle as P
import jxl_coder_amd as J
w, h, n = 3840, 2160, 64
img = synth.photo_like(w, h, seed=21, bits=16)
data = jxl_ref.encode(img, effort=7, distance=1.0, epf=3, primaries=9, transfer=16, intensity_target=10000.0)
ref = jxl_ref.decode(data, threads=64, allow16=True)[0]
dec = J.JxlDecoder(0)
out1, info = dec.decode_one_shot(data, allowed_floats=True)
d = np.abs(out1.astype(np.int32) - ref.astype(np.int32))
print("frame", len(data), "bytes; decode vs reference: u16 mean|diff| %.2f, >256: %.5f" % (d.mean(), (d > 256).mean()), "tf", info["transfer_function"], "prim", info["primaries"])
outs = [torch.empty(w * h * 8, dtype=torch.uint8, device="cuda") for _ in range(n)]
f16 = [torch.empty(w * h * 8, dtype=torch.uint8, device="cuda") for _ in range(n)]
torch.cuda.synchronize()
def chain():
    dec.decode_batch_to_device([data] * n, [o.data_ptr() for o in outs], [o.numel() for o in outs])
    for o, f in zip(outs, f16):
        dec.color_matrix_device(o.data_ptr(), w, h, True, 16, 9, 16, info["intensity_target"])
        dec.reformat_device(o.data_ptr(), w, h, True, 16, J.PreferredColorConfig.RGBA_F16, False, False, 29, f.data_ptr(), f.numel())
    torch.cuda.synchronize()
chain()
t = time.time(); chain(); dt = time.time() - t
print("flight of %d frames, decode + colour matrix/tone map + F16 reformat: %.0f ms = %.0f MP/s (one decoder context)" % (n, dt * 1e3, n * w * h / 1e6 / dt), dec.last_timing())
got = f16[0].cpu().numpy().view(np.uint16).reshape(h, w, 4)
exp = P.u16_to_f16(P.color_matrix(out1, 16, 9, 16, None, info["intensity_target"]), 16)
print("post stages vs oracle on the GPU decode: differing samples %.5f" % (got != exp).mean())
