"""Secondary drop-in boundary: the libjxl C-API subset of include/jxl_amd_libjxl.h, exported by
jxl_coder_amd/compat/libjxl.so + libjxl_threads.so.

CPU: every declared symbol is exported; the ABI structs / enum values equal the reference's vendored headers (compiled side by side when the
reference tree is present); the reference's OWN interop/JxlDecoding.cpp, compiled unchanged against the compat libraries, answers
DecodeBasicInfo (host-only path).  GPU: the same compiled reference driver decodes the golden files through the HIP kernels and returns
what the primary boundary returns."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_CPP = "/root/reference/jxlcoder/src/main/cpp"
DRIVER_SO = os.path.join(ROOT, "tests", "boundary", "libref_driver_on_compat.so")
ANIM_SO = os.path.join(ROOT, "tests", "boundary", "libref_anim_on_compat.so")


def _compat():
    import jxl_coder_amd as J
    from jxl_coder_amd import api
    J.build()
    return api.compat_dir()


def build_ref_driver():
    """The reference's interop/JxlDecoding.cpp + tests/boundary/ref_driver_entry.cpp -> tests/boundary/libref_driver_on_compat.so (build container only)."""
    d = _compat()
    if not os.path.isdir(REF_CPP):
        return os.path.exists(DRIVER_SO)
    srcs = [os.path.join(ROOT, "tests", "boundary", "ref_driver_entry.cpp"), os.path.join(REF_CPP, "interop", "JxlDecoding.cpp")]
    deps = srcs + [os.path.join(ROOT, "tests", "boundary", "boundary_entry.inc"), os.path.join(d, "libjxl.so"), os.path.join(d, "libjxl_threads.so")]
    if not os.path.exists(DRIVER_SO) or os.path.getmtime(DRIVER_SO) < max(os.path.getmtime(f) for f in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-include", "cstring", "-include", "cstdint", "-I", REF_CPP, "-I", os.path.join(REF_CPP, "jxl"),
                        "-I", os.path.join(REF_CPP, "interop"), "-o", DRIVER_SO] + srcs +
                       ["-L" + d, "-ljxl", "-ljxl_threads", "-Wl,-rpath,$ORIGIN/../../jxl_coder_amd/compat"], check=True)
    return True


def build_ref_anim():
    """The reference's interop/JxlAnimatedDecoder.cpp + tests/boundary/ref_anim_entry.cpp -> tests/boundary/libref_anim_on_compat.so (build container only)."""
    d = _compat()
    if not os.path.isdir(REF_CPP):
        return os.path.exists(ANIM_SO)
    srcs = [os.path.join(ROOT, "tests", "boundary", "ref_anim_entry.cpp"), os.path.join(REF_CPP, "interop", "JxlAnimatedDecoder.cpp")]
    deps = srcs + [os.path.join(d, "libjxl.so"), os.path.join(d, "libjxl_threads.so")]
    if not os.path.exists(ANIM_SO) or os.path.getmtime(ANIM_SO) < max(os.path.getmtime(f) for f in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-include", "cstring", "-include", "cstdint", "-include", "mutex", "-I", REF_CPP, "-I", os.path.join(REF_CPP, "jxl"),
                        "-I", os.path.join(REF_CPP, "interop"), "-o", ANIM_SO] + srcs +
                       ["-L" + d, "-ljxl", "-ljxl_threads", "-Wl,-rpath,$ORIGIN/../../jxl_coder_amd/compat"], check=True)
    return True


def test_every_declared_symbol_is_exported():
    d = _compat()
    hdr = open(os.path.join(ROOT, "include", "jxl_amd_libjxl.h")).read()
    names = set(re.findall(r"\b(Jxl(?:Decoder|Signature|ResizableParallelRunner)\w*)\s*\(", hdr.split("JXL_AMD_LIBJXL_NO_PROTOTYPES", 1)[1]))
    assert len(names) == 27, sorted(names)            # 17 + 5 of the still decode path, 5 more of the animated decoder's
    exported = set()
    for lib in ("libjxl.so", "libjxl_threads.so"):
        out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(d, lib)], capture_output=True, text=True, check=True).stdout
        exported |= {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert names <= exported, names - exported


def test_abi_layout_equals_the_reference_headers(tmp_path):
    if not os.path.isdir(REF_CPP):
        pytest.skip("needs the reference's vendored libjxl headers")
    src = tmp_path / "layout.cpp"
    fields = {"JxlBasicInfo": ["have_container", "xsize", "ysize", "bits_per_sample", "exponent_bits_per_sample", "intensity_target", "min_nits",
                               "relative_to_max_display", "linear_below", "uses_original_profile", "have_preview", "have_animation", "orientation",
                               "num_color_channels", "num_extra_channels", "alpha_bits", "alpha_exponent_bits", "alpha_premultiplied", "preview", "animation",
                               "intrinsic_xsize", "intrinsic_ysize", "padding"],
              "JxlColorEncoding": ["color_space", "white_point", "white_point_xy", "primaries", "primaries_red_xy", "primaries_green_xy", "primaries_blue_xy",
                                   "transfer_function", "gamma", "rendering_intent"],
              "JxlPixelFormat": ["num_channels", "data_type", "endianness", "align"],
              "JxlAnimationHeader": ["tps_numerator", "tps_denominator", "num_loops", "have_timecodes"],
              "JxlBlendInfo": ["blendmode", "source", "alpha", "clamp"],
              "JxlLayerInfo": ["have_crop", "crop_x0", "crop_y0", "xsize", "ysize", "blend_info", "save_as_reference"],
              "JxlFrameHeader": ["duration", "timecode", "name_length", "is_last", "layer_info"]}
    mine = {"JxlBasicInfo": "JxlcBasicInfo", "JxlColorEncoding": "JxlcColorEncoding", "JxlPixelFormat": "JxlcPixelFormat", "JxlAnimationHeader": "JxlcAnimationHeader",
            "JxlBlendInfo": "JxlcBlendInfo", "JxlLayerInfo": "JxlcLayerInfo", "JxlFrameHeader": "JxlcFrameHeader"}
    lines = ['#include <cstddef>', '#include "jxl/decode.h"', '#include "jxl/resizable_parallel_runner.h"', '#define JXL_AMD_LIBJXL_NO_PROTOTYPES', '#include "jxl_amd_libjxl.h"']
    for ref, flds in fields.items():
        lines.append(f"static_assert(sizeof({ref}) == sizeof({mine[ref]}), \"size of {ref}\");")
        for f in flds:
            lines.append(f"static_assert(offsetof({ref}, {f}) == offsetof({mine[ref]}, {f}), \"{ref}.{f}\");")
    for a, b in (("JXL_DEC_SUCCESS", "JXLC_DEC_SUCCESS"), ("JXL_DEC_ERROR", "JXLC_DEC_ERROR"), ("JXL_DEC_NEED_MORE_INPUT", "JXLC_DEC_NEED_MORE_INPUT"),
                 ("JXL_DEC_NEED_IMAGE_OUT_BUFFER", "JXLC_DEC_NEED_IMAGE_OUT_BUFFER"), ("JXL_DEC_BASIC_INFO", "JXLC_DEC_BASIC_INFO"),
                 ("JXL_DEC_COLOR_ENCODING", "JXLC_DEC_COLOR_ENCODING"), ("JXL_DEC_FRAME", "JXLC_DEC_FRAME"), ("JXL_DEC_FULL_IMAGE", "JXLC_DEC_FULL_IMAGE"),
                 ("JXL_TYPE_UINT8", "JXLC_TYPE_UINT8"), ("JXL_TYPE_UINT16", "JXLC_TYPE_UINT16"), ("JXL_TYPE_FLOAT", "JXLC_TYPE_FLOAT"), ("JXL_TYPE_FLOAT16", "JXLC_TYPE_FLOAT16"),
                 ("JXL_SIG_CODESTREAM", "JXLC_SIG_CODESTREAM"), ("JXL_SIG_CONTAINER", "JXLC_SIG_CONTAINER"), ("JXL_SIG_INVALID", "JXLC_SIG_INVALID"),
                 ("JXL_SIG_NOT_ENOUGH_BYTES", "JXLC_SIG_NOT_ENOUGH_BYTES")):
        lines.append(f"static_assert((int){a} == (int){b}, \"{a}\");")
    lines.append("static_assert(sizeof(JxlParallelRunner) == sizeof(JxlcParallelRunner), \"runner pointer\");")
    lines.append("int main() { return 0; }")
    src.write_text("\n".join(lines))
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", REF_CPP, "-I", os.path.join(ROOT, "include"), str(src)], check=True)


def test_event_order_and_header_calls_on_the_host():
    """The state machine without a GPU: BASIC_INFO -> COLOR_ENCODING -> NEED_IMAGE_OUT_BUFFER; sizes, colour encoding, error on garbage."""
    d = _compat()
    L = C.CDLL(os.path.join(d, "libjxl.so"))
    L.JxlDecoderCreate.restype = C.c_void_p
    for f in ("JxlDecoderDestroy", "JxlDecoderSubscribeEvents", "JxlDecoderSetInput", "JxlDecoderCloseInput", "JxlDecoderProcessInput", "JxlDecoderGetBasicInfo",
              "JxlDecoderImageOutBufferSize", "JxlDecoderGetColorAsEncodedProfile", "JxlDecoderGetICCProfileSize"):
        getattr(L, f).argtypes = [C.c_void_p] + {"JxlDecoderSubscribeEvents": [C.c_int], "JxlDecoderSetInput": [C.c_char_p, C.c_size_t], "JxlDecoderGetBasicInfo": [C.c_void_p],
                                                  "JxlDecoderImageOutBufferSize": [C.c_void_p, C.c_void_p], "JxlDecoderGetColorAsEncodedProfile": [C.c_int, C.c_void_p],
                                                  "JxlDecoderGetICCProfileSize": [C.c_int, C.c_void_p]}.get(f, [])
    data = open(os.path.join(ROOT, "tests/golden/v160x120_16bit_pq2100_epf3.jxl"), "rb").read()
    dec = L.JxlDecoderCreate(None)
    assert L.JxlDecoderSubscribeEvents(dec, 0x40 | 0x100 | 0x1000) == 0
    assert L.JxlDecoderSetInput(dec, data, len(data)) == 0
    L.JxlDecoderCloseInput(dec)
    assert L.JxlDecoderProcessInput(dec) == 0x40
    info = (C.c_uint32 * 64)()
    assert L.JxlDecoderGetBasicInfo(dec, info) == 0
    assert (info[1], info[2], info[3]) == (160, 120, 16)                     # xsize, ysize, bits_per_sample
    fmt16 = (C.c_uint64 * 3)(); C.memmove(fmt16, (C.c_uint32 * 4)(4, 3, 0, 0), 16)      # {4, JXL_TYPE_UINT16, native, align 0}
    size = C.c_size_t(0)
    assert L.JxlDecoderImageOutBufferSize(dec, fmt16, C.byref(size)) == 0 and size.value == 160 * 120 * 8
    assert L.JxlDecoderProcessInput(dec) == 0x100
    ce = (C.c_double * 14)()
    assert L.JxlDecoderGetColorAsEncodedProfile(dec, 1, ce) == 0
    ce32 = C.cast(ce, C.POINTER(C.c_uint32))
    assert ce32[0] == 0 and ce32[6] == 9 and ce32[20] == 16                   # RGB, primaries 2100 (offset 24), transfer PQ (offset 80)
    assert L.JxlDecoderProcessInput(dec) == 5                                 # JXL_DEC_NEED_IMAGE_OUT_BUFFER
    L.JxlDecoderDestroy(dec)
    dec = L.JxlDecoderCreate(None)
    L.JxlDecoderSubscribeEvents(dec, 0x40)
    L.JxlDecoderSetInput(dec, b"\xff\x0a" + b"\x00" * 5, 7); L.JxlDecoderCloseInput(dec)
    assert L.JxlDecoderProcessInput(dec) == 1                                 # JXL_DEC_ERROR, and it stays an error
    assert L.JxlDecoderProcessInput(dec) == 1
    L.JxlDecoderDestroy(dec)


_CHILD = r"""
import ctypes as C, os, sys, numpy as np
ROOT = sys.argv[1]; gpu = sys.argv[2] == "gpu"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_libjxl_abi as T
L = T.driver_handle()
maps = open("/proc/self/maps").read()
assert "jxl_coder_amd/compat/libjxl.so" in maps and "oracle/_ref" not in maps, "the driver must be bound to the compat library, not to the reference's libjxl"
for name, wh in (("v264x520_e7", (264, 520)), ("asset_first_jxl", (768, 768)), ("l512_e7", (512, 512))):
    data = open(os.path.join(ROOT, "tests/golden", name + ".jxl"), "rb").read()
    out = (C.c_uint64 * 2)()
    assert L.boundary_basic_info(data, len(data), C.byref(out)) == 1 and (out[0], out[1]) == wh, name
assert L.boundary_basic_info(b"not a jxl file at all", 21, C.byref((C.c_uint64 * 2)())) == 0
if gpu:
    import torch
    assert torch.cuda.is_available()
    import jxl_coder_amd as J
    from jxl_coder_amd import api
    for name, allowed_floats in (("v264x520_e7", 0), ("va300x520_e7", 0), ("l512_e7", 0), ("v160x120_16bit_pq2100_epf3", 1), ("v160x120_16bit_pq2100_epf3", 0),
                                 ("asset_wide_gamut", 1), ("va400x300_e7_d2", 0), ("lra200x150_e5", 0),
                                 # round 4, last part: delta palette, cjxl -p RGBA, two LF levels, noise on an upsampled frame, previous-channel properties,
                                 # float16 (lossless HDR range; VarDCT + float alpha), premultiplied alpha, grey + alpha, an extra channel besides the alpha
                                 ("lpl400x300_e7", 0), ("vaqr520x300_e7", 0), ("vlf2a520x300_e7", 0), ("vnu523x267_e7_d12", 0), ("lpc200x136_e7_prev3", 0),
                                 ("lf16_300x200_e7_hdr", 0), ("vf16a300x200_e7", 1), ("vpm400x300_e7_premultiplied", 0), ("lga300x200_e7", 0), ("vxs400x300_e7_rgba_spot", 0)):
        data = open(os.path.join(ROOT, "tests/golden", name + ".jxl"), "rb").read()
        dec = J.JxlDecoder(0)
        exp, info = dec.decode_one_shot(data, allowed_floats=bool(allowed_floats))
        dec.close()
        out = np.zeros(exp.nbytes, np.uint8)
        meta = (C.c_uint64 * 12)(); xy = (C.c_double * 8)(); msg = C.create_string_buffer(256)
        rc = L.boundary_decode(data, len(data), allowed_floats, out.ctypes.data, out.size, C.byref(meta), C.byref(xy), msg, 256)
        assert rc == 1, (name, rc, api.lib().jxlamd_last_error(None))
        assert (meta[0], meta[1]) == (exp.shape[1], exp.shape[0]) and meta[3] == info["out_bits"] and meta[2] == int(info["out_bits"] == 16), name
        assert np.array_equal(out.view(exp.dtype).reshape(exp.shape), exp), name            # same kernels underneath: bit for bit
        assert meta[6] == info["prefer_encoding"] and meta[7] == info["has_alpha_in_origin"] and meta[4] == info["alpha_premultiplied"], name
        assert (meta[8], meta[9]) == (info["primaries"], info["transfer_function"]), name
        assert meta[10] == 0, name                                                          # preferEncoding: the ICC vector is cleared (JxlDecoding.cpp:142-144)
    # linear-light enum encodings: not 'preferred' -> the driver fetches the (synthesised) data profile and fails the decode without it (:135-141)
    for name in ("vlin96x64_e3", "llin96x64_e3", "vlingrey96x64_e3"):
        data = open(os.path.join(ROOT, "tests/golden", name + ".jxl"), "rb").read()
        want = np.load(os.path.join(ROOT, "tests/golden", name + ".npz"))["rgba"]
        out = np.zeros(want.nbytes, np.uint8)
        meta = (C.c_uint64 * 12)(); xy = (C.c_double * 8)(); msg = C.create_string_buffer(256)
        rc = L.boundary_decode(data, len(data), 0, out.ctypes.data, out.size, C.byref(meta), C.byref(xy), msg, 256)
        assert rc == 1, (name, rc, api.lib().jxlamd_last_error(None))
        assert meta[6] == 0 and meta[10] == os.path.getsize(os.path.join(ROOT, "tests/golden", name + ".icc")), (name, list(meta))      # preferEncoding false, ICC vector filled
        d = np.abs(out.view(want.dtype).reshape(want.shape).astype(int) - want.astype(int))
        assert (d.max() == 0) if name.startswith("l") else (d.max() <= 1 and d.mean() <= 0.08), (d.max(), d.mean())      # effort-3 file with one EPF iteration, linear-light codes: the rcpps offset of conftest.py (measured 0.051), (name, d.max(), d.mean())
print("driver ok")
"""


_ANIM_CHILD = r"""
import ctypes as C, json, os, sys, numpy as np
ROOT = sys.argv[1]; gpu = sys.argv[2] == "gpu"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_libjxl_abi as T
from jxl_coder_amd import api
api.lib()
for so in ("libjxl_threads.so", "libjxl.so"):           # by path, before the driver asks for them by soname
    C.CDLL(os.path.join(api.compat_dir(), so), mode=C.RTLD_GLOBAL)
L = C.CDLL(T.ANIM_SO)
maps = open("/proc/self/maps").read()
assert "jxl_coder_amd/compat/libjxl.so" in maps and "oracle/_ref" not in maps, "the animated decoder must be bound to the compat library, not to the reference's libjxl"
L.refanim_open.restype = C.c_void_p; L.refanim_open.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
for f in ("refanim_close", "refanim_frames", "refanim_loops"): getattr(L, f).argtypes = [C.c_void_p]
L.refanim_duration.argtypes = [C.c_void_p, C.c_int]; L.refanim_size.argtypes = [C.c_void_p, C.POINTER(C.c_uint32 * 2)]
L.refanim_get_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
meta = json.load(open(os.path.join(ROOT, "tests/golden/golden.json")))
msg = C.create_string_buffer(256)
assert not L.refanim_open(b"GIF89a....", 10, msg, 256) and msg.value == b"Not an JXL image"
for name in ("an_blend_lossless", "an_modes_d2_e5", "asset_animated", "v256_e7", "an_blend_d12_e7", "an_blend_premul_lossless"):
    data = open(os.path.join(ROOT, "tests/golden", name + ".jxl"), "rb").read()
    h = L.refanim_open(data, len(data), msg, 256)
    assert h, (name, msg.value)
    n = L.refanim_frames(h)
    durations = [L.refanim_duration(h, i) for i in range(n)]
    wh = (C.c_uint32 * 2)(); L.refanim_size(h, C.byref(wh))
    if name.startswith("an_"):
        # the reference's own constructor (coalescing off, every frame skipped) over the compat library = what it collects over its libjxl
        assert durations == meta[name]["durations_ms"] and L.refanim_loops(h) == meta[name]["loops"], (name, durations)
        assert (wh[0], wh[1]) == (160, 120)
    elif name == "asset_animated":
        assert n == 48 and (wh[0], wh[1]) == (128, 128)
    else:
        assert n == 1 and durations == [0] and L.refanim_loops(h) == -1
    if gpu and name.startswith("an_"):
        frames = np.load(os.path.join(ROOT, "tests/golden", name + ".npz"))["frames"]
        for i in range(len(frames)):
            out = np.zeros(frames[i].nbytes, np.uint8); dur = C.c_int(); pref = C.c_int()
            assert L.refanim_get_frame(h, i, out.ctypes.data, out.size, C.byref(dur), C.byref(pref), msg, 256) == 1, (name, i, msg.value, api.lib().jxlamd_last_error(None))
            got = out.reshape(frames[i].shape)
            d = np.abs(got.astype(int) - frames[i].astype(int))
            assert (d.max() == 0) if "lossless" in name else (d.max() <= 1 and d.mean() <= 0.05), (name, i, d.max(), d.mean())
            assert pref.value == 1                                                  # sRGB enum encoding: the reference uses it and ignores the ICC bytes
        assert L.refanim_get_frame(h, len(frames) + 3, None, 0, C.byref(dur), C.byref(pref), msg, 256) == 0      # beyond the coalesced frames: an AnimatedDecoderError
    if gpu and name == "v256_e7":
        import jxl_coder_amd as J
        exp, _ = J.JxlDecoder(0).decode_one_shot(data, allowed_floats=False)
        out = np.zeros(exp.nbytes, np.uint8); dur = C.c_int(); pref = C.c_int()
        assert L.refanim_get_frame(h, 0, out.ctypes.data, out.size, C.byref(dur), C.byref(pref), msg, 256) == 1 and np.array_equal(out.reshape(exp.shape), exp)
    L.refanim_close(h)
print("anim ok")
"""


def _run_anim_child(mode):
    if not build_ref_anim():
        pytest.skip("the reference's animated decoder is compiled in the build container (needs the reference tree)")
    r = subprocess.run([sys.executable, "-c", _ANIM_CHILD, ROOT, mode], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "anim ok" in r.stdout, r.stdout[-800:] + r.stderr[-2000:]


def test_reference_animated_decoder_walks_frames_on_the_host():
    """interop/JxlAnimatedDecoder.hpp:68-185 — the reference's own constructor (frame events with coalescing off, JxlDecoderSkipCurrentFrame, durations,
    loop count) compiled unchanged against compat/libjxl.so: no GPU involved."""
    _run_anim_child("cpu")


@pytest.mark.gpu
def test_reference_animated_decoder_gets_frames_through_the_libjxl_abi():
    """JxlAnimatedDecoder::getFrame (interop/JxlAnimatedDecoder.cpp:28-144: JxlDecoderRewind, JxlDecoderSkipFrames, coalescing on) — the reference's own
    object code — over the libjxl-ABI subset: every coalesced frame of the layered fixtures equals the golden frames the reference's libjxl produced."""
    _run_anim_child("gpu")


def driver_handle():
    """(child process) the compiled reference driver, bound to compat/libjxl.so"""
    from jxl_coder_amd import api
    api.lib()
    for so in ("libjxl_threads.so", "libjxl.so"):       # by path, before the driver asks for them by soname
        C.CDLL(os.path.join(api.compat_dir(), so), mode=C.RTLD_GLOBAL)
    L = C.CDLL(DRIVER_SO)
    L.boundary_basic_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64 * 2)]
    L.boundary_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64 * 12), C.POINTER(C.c_double * 8), C.c_char_p, C.c_size_t]
    return L


def _run_child(mode):
    """The driver runs in a process of its own: a process that has already loaded the REFERENCE's libjxl.so (oracle/_ref, the live checker of
    other tests) would hand that library to the driver — the dynamic loader matches sonames — and the test would compare the reference with
    itself."""
    if not build_ref_driver():
        pytest.skip("the reference's driver is compiled in the build container (needs the reference tree)")
    r = subprocess.run([sys.executable, "-c", _CHILD, ROOT, mode], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "driver ok" in r.stdout, r.stdout[-800:] + r.stderr[-2000:]


def test_reference_driver_answers_basic_info_on_the_host():
    """interop/JxlDecoding.cpp:178-225 (DecodeBasicInfo), the reference's own object code, against compat/libjxl.so: no GPU involved."""
    _run_child("cpu")


@pytest.mark.gpu
def test_reference_driver_decodes_through_the_libjxl_abi():
    """DecodeJpegXlOneShot — the reference's own compiled driver loop (JxlDecoding.cpp:46-171) — over the libjxl-ABI subset: pixels and
    out-params equal the primary boundary's (jxlamd_decode through the Python mirror), incl. squeezed alpha and a responsive lossless file."""
    _run_child("gpu")
