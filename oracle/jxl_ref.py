"""ctypes wrapper over oracle/_ref/libref_harness.so — the reference's own prebuilt libjxl 0.12.0
(jxlcoder/src/main/cpp/lib/x86_64/libjxl.so) driven with the reference's call sequence
(jxlcoder/src/main/cpp/interop/JxlDecoding.cpp:46-171, JxlEncoding.cpp:54-192).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke().
"""
import ctypes as C
import os
import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
_lib = None


class RefInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "xsize", "ysize", "bits_per_sample", "exponent_bits", "num_color_channels", "num_extra_channels",
        "alpha_bits", "alpha_premultiplied", "orientation", "have_animation", "uses_original_profile", "out_bits")] + \
        [("intensity_target", C.c_float), ("prefer_encoding", C.c_uint32), ("have_encoded_profile", C.c_uint32),
         ("color_space", C.c_uint32), ("white_point", C.c_uint32), ("primaries", C.c_uint32),
         ("transfer_function", C.c_uint32), ("rendering_intent", C.c_uint32), ("gamma", C.c_double),
         ("icc_size", C.c_uint32), ("version", C.c_uint32), ("xy", C.c_double * 8)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "xy"}
        d["xy"] = [float(v) for v in self.xy]
        return d


class RefEncParams(C.Structure):
    _fields_ = [("xsize", C.c_uint32), ("ysize", C.c_uint32), ("num_channels", C.c_uint32), ("bits", C.c_uint32),
                ("lossless", C.c_int32), ("distance", C.c_float), ("effort", C.c_int32), ("decoding_speed", C.c_int32),
                ("gaborish", C.c_int32), ("epf", C.c_int32), ("primaries", C.c_int32), ("transfer", C.c_int32),
                ("intensity_target", C.c_float), ("modular", C.c_int32), ("threads", C.c_int32),
                ("extra", (C.c_int32 * 2) * 8)]


def available():
    return os.path.exists(os.path.join(_DIR, "libref_harness.so")) and os.path.exists(os.path.join(_DIR, "libjxl.so"))


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(_DIR, "libref_harness.so"), mode=C.RTLD_LOCAL)
        _lib.ref_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_size_t), C.POINTER(RefInfo), C.c_void_p, C.c_size_t]
        _lib.ref_encode.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(RefEncParams), C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_size_t)]
        _lib.ref_free.argtypes = [C.c_void_p]
        _lib.ref_set_icc.argtypes = [C.c_char_p, C.c_size_t]
        _lib.ref_version.restype = C.c_int
        _lib.ref_set_custom_xy.argtypes = [C.c_void_p]
        _lib.ref_set_extra_channel.argtypes = [C.c_void_p, C.c_int]
        _lib.ref_set_premultiplied.argtypes = [C.c_int]
        _lib.ref_set_float.argtypes = [C.c_int]
    return _lib


def version():
    return lib().ref_version()


def decode(data: bytes, threads=0, allow16=True, mode=0):
    """-> (pixels ndarray [h,w,4] u8/u16/f32, info dict, icc bytes). Raises ValueError on decode failure."""
    out = C.c_void_p()
    n = C.c_size_t()
    ri = RefInfo()
    icc = C.create_string_buffer(1 << 20)
    rc = lib().ref_decode(data, len(data), threads, int(allow16), mode, C.byref(out), C.byref(n), C.byref(ri), icc, len(icc))
    if rc != 0:
        raise ValueError(f"ref_decode failed rc={rc}")
    dt = {8: np.uint8, 16: np.uint16, 32: np.float32}[ri.out_bits]
    # (C.string_at takes a C int: outputs of 2 GiB and more — BASELINE config 4 — need the array view)
    arr = np.frombuffer((C.c_uint8 * n.value).from_address(out.value), dtype=dt).reshape(ri.ysize, ri.xsize, 4).copy()
    lib().ref_free(out)
    return arr, ri.as_dict(), icc.raw[:ri.icc_size]


def encode(pixels: np.ndarray, lossless=False, distance=1.0, effort=7, decoding_speed=0, gaborish=-1, epf=-1,
           primaries=0, transfer=0, intensity_target=0.0, modular=-1, threads=0, extra=(), icc=None, orientation=1, custom_xy=None, extra_channel=None, premultiplied=False, int_bits=0):
    """pixels: [h,w,c] u8 or u16 — or float32 / float16: a floating-point image (32 bits / 8 exponent bits, 16 / 5), nominal range 0..1 —, c in 1,2,3,4
    (2: grey + alpha). Same sequence as the reference's EncodeJxlOneshot.
    extra_channel: (plane [h,w] u8, JxlExtraChannelType) — one more extra channel behind the alpha (1 depth, 2 spot colour, 3 selection mask, 4 black, ...)."""
    pixels = np.ascontiguousarray(pixels)
    h, w, c = pixels.shape
    p = RefEncParams()
    p.xsize, p.ysize, p.num_channels = w, h, c
    p.bits = 16 if pixels.dtype == np.uint16 else 8
    p.lossless, p.distance, p.effort, p.decoding_speed = int(lossless), distance, effort, decoding_speed
    p.gaborish, p.epf, p.primaries, p.transfer = gaborish, epf, primaries, transfer
    p.intensity_target, p.modular, p.threads = intensity_target, modular, threads
    for i in range(8):
        p.extra[i][0] = -1
    for i, (k, v) in enumerate(extra):
        p.extra[i][0], p.extra[i][1] = k, v
    out = C.c_void_p()
    n = C.c_size_t()
    lib().ref_set_icc(icc or b"", len(icc) if icc else 0)      # JxlEncoderSetICCProfile (interop/JxlEncoding.cpp:125-129) instead of an enum profile
    lib().ref_set_orientation(int(orientation))                # JxlBasicInfo.orientation of the file (the decoder re-orients by default)
    _xy = (C.c_double * 8)(*custom_xy) if custom_xy else None
    lib().ref_set_custom_xy(C.cast(_xy, C.c_void_p) if custom_xy else None)      # custom white point + primaries (white xy, red, green, blue xy) instead of the enum values
    _ecp = np.ascontiguousarray(extra_channel[0], dtype=np.uint8) if extra_channel is not None else None
    lib().ref_set_extra_channel(C.c_void_p(_ecp.ctypes.data) if _ecp is not None else None, int(extra_channel[1]) if extra_channel is not None else 0)
    lib().ref_set_premultiplied(int(premultiplied))          # the alpha is declared premultiplied (pixels taken as they are)
    lib().ref_set_float(1 if pixels.dtype == np.float32 else 2 if pixels.dtype == np.float16 else 0)
    lib().ref_set_int_bits(int(int_bits))      # float32 pixels of an INTEGER image of int_bits bits (17 .. 31): values k / (2^bits - 1)
    rc = lib().ref_encode(pixels.ctypes.data, pixels.nbytes, C.byref(p), C.byref(out), C.byref(n))
    lib().ref_set_float(0)
    lib().ref_set_int_bits(0)
    lib().ref_set_premultiplied(0)
    lib().ref_set_extra_channel(None, 0)
    lib().ref_set_icc(b"", 0)
    lib().ref_set_orientation(1)
    lib().ref_set_custom_xy(None)
    if rc != 0:
        raise ValueError(f"ref_encode failed rc={rc}")
    data = C.string_at(out.value, n.value)
    lib().ref_free(out)
    return data


def encode_jpeg(jpeg: bytes, effort=7):
    """JPEG -> JPEG XL with the reference's JxlConstruction call sequence (interop/JxlConstruction.hpp:46-90)."""
    out = C.c_void_p(); n = C.c_size_t()
    rc = lib().ref_encode_jpeg(jpeg, len(jpeg), int(effort), C.byref(out), C.byref(n))
    if rc != 0:
        raise ValueError(f"ref_encode_jpeg failed rc={rc}")
    data = C.string_at(out.value, n.value)
    lib().ref_free(out)
    return data


def fnv1a64(b: bytes) -> int:
    h = 0xcbf29ce484222325
    # vectorised FNV is awkward; fall back to a C-speed-ish loop via int.from_bytes chunks
    for x in b:
        h = ((h ^ x) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


class RefAnimFrame(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("x0", C.c_int32), ("y0", C.c_int32),
                ("blend_mode", C.c_int32), ("source", C.c_int32), ("save_as_reference", C.c_int32), ("duration", C.c_uint32), ("alpha_blend_mode", C.c_int32)]


def encode_anim(frames, W, H, lossless=True, distance=1.0, effort=3, tps=(100, 1), loops=0, premultiplied=False):
    """frames: list of dicts {rgba: [h,w,4] u8, x0, y0, blend (0 replace, 1 add, 2 blend, 3 muladd, 4 mul), source, save, duration (ticks)} -> an
    animated JPEG XL over a W x H RGBA canvas (libjxl's encoder API, JxlEncoderSetFrameHeader with layer_info)."""
    arr = (RefAnimFrame * len(frames))()
    keep = []
    for i, f in enumerate(frames):
        px = np.ascontiguousarray(f["rgba"], dtype=np.uint8); keep.append(px)
        arr[i].rgba = px.ctypes.data; arr[i].h, arr[i].w = px.shape[:2]
        arr[i].x0, arr[i].y0 = f.get("x0", 0), f.get("y0", 0)
        arr[i].blend_mode, arr[i].source, arr[i].save_as_reference, arr[i].duration = f.get("blend", 0), f.get("source", 0), f.get("save", 0), f.get("duration", 1)
        arr[i].alpha_blend_mode = f.get("alpha_blend", -1)      # the alpha channel's own blend mode (-1: as the colour)
    out = C.c_void_p(); n = C.c_size_t()
    lib().ref_set_premultiplied(int(premultiplied))
    rc = lib().ref_encode_anim(arr, len(frames), W, H, int(lossless), C.c_float(distance), effort, tps[0], tps[1], loops, C.byref(out), C.byref(n))
    lib().ref_set_premultiplied(0)
    if rc != 0:
        raise ValueError(f"ref_encode_anim failed rc={rc}")
    data = C.string_at(out.value, n.value)
    lib().ref_free(out)
    return data


def anim_info(data: bytes):
    """-> (durations_ms list, loops): the frame list of the reference's JxlAnimatedDecoder constructor (coalescing off)"""
    d = (C.c_int32 * 4096)(); loops = C.c_int32()
    n = lib().ref_anim_info(data, len(data), d, 4096, C.byref(loops))
    if n < 0:
        raise ValueError(f"ref_anim_info failed rc={n}")
    return [int(d[i]) for i in range(n)], int(loops.value)


def decode_frame(data: bytes, index: int):
    """coalesced frame `index` as RGBA8 [h,w,4]: the reference's JxlAnimatedDecoder::getFrame sequence"""
    out = C.c_void_p(); n = C.c_size_t(); w = C.c_uint32(); h = C.c_uint32()
    rc = lib().ref_decode_frame(data, len(data), int(index), C.byref(out), C.byref(n), C.byref(w), C.byref(h))
    if rc != 0:
        raise ValueError(f"ref_decode_frame failed rc={rc}")
    arr = np.frombuffer((C.c_uint8 * n.value).from_address(out.value), dtype=np.uint8).reshape(h.value, w.value, 4).copy()
    lib().ref_free(out)
    return arr
