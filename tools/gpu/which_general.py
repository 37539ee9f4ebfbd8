"""Which of the mixed workload's frame kinds send a decoder context to the general LF kernel (kErrNeedGeneral), and what their LF stage costs in a flight of 8?"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")]
import torch
import jxl_coder_amd as J
import make_bench_frames as M
f = J.api.lib().jxlamd_debug_lf_general
f.argtypes = [C.c_void_p]; f.restype = C.c_int
g = J.api.lib().jxlamd_debug_modular
g.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 2)]
for seed, kind in ((0, "screenshot"), (1, "rgba"), (2, "photo d12")):
    data = M.encode(seed, 8, "mixed")
    dec = J.JxlDecoder(0)
    n = C.c_size_t(); J.api.lib().jxlamd_output_size(data, len(data), 0, C.byref(n))
    outs = [torch.empty(int(n.value), dtype=torch.uint8, device="cuda") for _ in range(8)]
    for rep in range(3):
        t = time.time(); dec.decode_batch_to_device([data] * 8, [o.data_ptr() for o in outs], [outs[0].numel()] * 8); dt = time.time() - t
    st = (C.c_uint64 * 2)(); g(dec._h, C.byref(st))
    print(kind, "general LF build:", int(f(dec._h)), "flight of 8: %.1f ms" % (dt * 1e3), dec.last_timing(), "serial streams", int(st[0]), "block-form channels", int(st[1]))
