"""ctypes wrapper over oracle/libjxo.so — the plain-C CPU restatement of the JPEG XL decode path
(what the reference runs inside libjxl, call site jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:75).
TEST INFRASTRUCTURE ONLY (tests/, bench.py cpu_baseline leg, __graft_entry__.smoke())."""
import ctypes as C
import os
import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libjxo.so")
_lib = None


class Info(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("xsize", "ysize", "bits_per_sample", "exp_bits", "num_color_channels",
                                          "num_extra_channels", "alpha_bits", "alpha_premultiplied", "orientation",
                                          "have_animation", "xyb_encoded")] + \
        [("intensity_target", C.c_float)] + \
        [(n, C.c_uint32) for n in ("want_icc", "color_space", "white_point", "primaries", "transfer_function",
                                   "rendering_intent", "have_gamma")] + [("gamma", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_PATH)
        _lib.jxo_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(Info)]
        _lib.jxo_basic_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Info)]
        _lib.jxo_last_error.restype = C.c_char_p
        _libc = C.CDLL(None)
        _libc.free.argtypes = [C.c_void_p]
        _lib._free = _libc.free
    return _lib


def basic_info(data: bytes):
    info = Info()
    if lib().jxo_basic_info(data, len(data), C.byref(info)) != 0:
        raise ValueError(lib().jxo_last_error().decode())
    return info.as_dict()


def decode(data: bytes, out_bits=8, debug=False, epf_x86=None):
    """epf_x86: True = the EPF normalises with the reference x86 build's rcpps (oracle/jxo_rcp12.h), False = the exact quotient, None = as JXO_EPF_RCPPS says"""
    C.c_int.in_dll(lib(), "jxo_epf_rcp").value = -1 if epf_x86 is None else int(bool(epf_x86))
    out = C.c_void_p()
    n = C.c_size_t()
    info = Info()
    C.c_int.in_dll(lib(), "jxo_debug").value = int(debug)
    rc = lib().jxo_decode(data, len(data), out_bits, C.byref(out), C.byref(n), C.byref(info))
    if rc != 0:
        raise ValueError(lib().jxo_last_error().decode())
    arr = np.frombuffer(C.string_at(out.value, n.value), dtype=np.uint16 if out_bits == 16 else np.uint8)
    arr = arr.reshape(info.ysize, info.xsize, 4).copy()
    lib()._free(out)
    return arr, info.as_dict()
