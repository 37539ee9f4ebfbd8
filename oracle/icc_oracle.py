"""convertUseDefinedColorSpace (cpp/colorspaces/colorspace.cpp:38-86) restated over Little CMS through ctypes — TEST INFRASTRUCTURE ONLY.
Same calls and parameters as the reference: cmsOpenProfileFromMem, cmsCreate_sRGBProfile, cmsCreateTransform(src, TYPE_RGBA_8 |
TYPE_RGBA_16_PREMUL, sRGB, same, INTENT_PERCEPTUAL, BLACKPOINTCOMPENSATION | NOWHITEONWHITEFIXUP | COPY_ALPHA), one cmsDoTransform per row.
The library: oracle/_ref/liblcms2_ref.so = the Little CMS 2.16 the reference vendors under cpp/icc, compiled from those sources where they
lie (oracle/ref_lcms/Makefile) — the reference's own CMM; the distribution's liblcms2 (2.12) only when that build is absent.  (The reference's
colorspace.cpp itself cannot be compiled here without a stand-in for android/log.h.)"""
import os
import ctypes as C
import numpy as np

TYPE_RGBA_8 = (4 << 16) | (1 << 7) | (3 << 3) | 1
TYPE_RGBA_16_PREMUL = (4 << 16) | (1 << 7) | (3 << 3) | 2 | (1 << 23)
FLAGS = 0x2000 | 0x0004 | 0x04000000
_L = None


def _lib():
    global _L
    if _L is None:
        ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "liblcms2_ref.so")
        L = C.CDLL(ref if os.path.exists(ref) and not os.environ.get("JXO_SYSTEM_LCMS") else "liblcms2.so.2")
        L.cmsOpenProfileFromMem.restype = C.c_void_p; L.cmsOpenProfileFromMem.argtypes = [C.c_char_p, C.c_uint32]
        L.cmsCreate_sRGBProfile.restype = C.c_void_p
        L.cmsCreateTransform.restype = C.c_void_p; L.cmsCreateTransform.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.cmsDoTransform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.cmsDeleteTransform.argtypes = [C.c_void_p]; L.cmsCloseProfile.argtypes = [C.c_void_p]
        L.cmsGetEncodedCMMversion.restype = C.c_int
        _L = L
    return _L


def cmm_version():
    return int(_lib().cmsGetEncodedCMMversion())


def available():
    try:
        _lib()
        return True
    except OSError:
        return False


def convert(px: np.ndarray, icc: bytes) -> np.ndarray:
    """px [h, w, 4] u8 or u16 -> the reference's result (same dtype)."""
    L = _lib()
    src = L.cmsOpenProfileFromMem(icc, len(icc))
    if not src:
        return px.copy()
    dst = L.cmsCreate_sRGBProfile()
    fmt = TYPE_RGBA_16_PREMUL if px.dtype == np.uint16 else TYPE_RGBA_8
    if px.dtype == np.uint16 and L.cmsGetEncodedCMMversion() < 2130:
        # the reference vendors Little CMS 2.16 (cpp/icc/lcms2.h:87); PREMUL formatters arrived in 2.13 and this box has 2.12, where
        # cmsCreateTransform rejects the format.  For OPAQUE pixels premultiplied == straight, so TYPE_RGBA_16 gives the reference's result;
        # callers of this oracle pass opaque 16-bit images only.
        assert (px[..., 3] == 65535).all(), "16-bit oracle on Little CMS < 2.13 is only valid for opaque alpha"
        fmt = (4 << 16) | (1 << 7) | (3 << 3) | 2
    xf = L.cmsCreateTransform(src, fmt, dst, fmt, 0, FLAGS)
    out = np.ascontiguousarray(px).copy()
    if xf:
        for y in range(out.shape[0]):
            row = out[y]
            L.cmsDoTransform(xf, row.ctypes.data, row.ctypes.data, out.shape[1])
        L.cmsDeleteTransform(xf)
    L.cmsCloseProfile(dst); L.cmsCloseProfile(src)
    return out
