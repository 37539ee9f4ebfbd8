"""oracle/post_oracle.py — TEST INFRASTRUCTURE ONLY: numpy restatement of the reference's first-party post-decode stages.

  A10  applyColorMatrix / applyColorMatrix16Bit   jxlcoder/src/main/cpp/colorspaces/ColorMatrix.cpp:35-219
       transfer curves                              colorspaces/Trc.cpp:31-329;  tone mapper  Rec2408ToneMapper.{h:36-45,cpp:80-100}
       matrix set-up at the call site               JniDecoding.cpp:138-228;  GamutRgbToXYZ  colorspaces/ColorSpaceProfile.h:131-143
  A11  ReformatColorConfig                          ReformatBitmap.cpp:46-263 and the imagebit/*.cpp kernels it calls

Pinned against the reference itself: oracle/_ref/libref_post.so is built from those sources where they lie
(oracle/ref_post/Makefile) and tests/test_post_stages.py compares every function here with it — integer stages bit-exact,
the LUT stage within one LUT step (libm powf vs numpy's float32 pow).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product never does.
"""
import numpy as np

F32 = np.float32

# PreferredColorConfig (cpp/Support.h:37-44)
DEFAULT, RGBA_8888, RGBA_F16, RGB_565, RGBA_1010102, HARDWARE = 1, 2, 3, 4, 5, 6


# ----------------------------------------------------------------------------------------------- A11 kernels
def associate8(px):                                   # imagebit/RGBAlpha.cpp:67-90: c*a/255, integer truncation
    o = px.copy()
    a = px[..., 3:4].astype(np.uint16)
    o[..., :3] = (px[..., :3].astype(np.uint16) * a // 255).astype(np.uint8)
    return o


def associate16(px, depth):                           # RGBAlpha.cpp:92-117: c*a/(2^depth-1)
    o = px.copy()
    mx = (1 << depth) - 1
    a = px[..., 3:4].astype(np.uint32)
    o[..., :3] = (px[..., :3].astype(np.uint32) * a // mx).astype(np.uint16)
    return o


def u16_to_f16(px, depth):                            # imagebit/RgbaU16toHF.cpp:42-144: half(float(v) * (1/(2^depth-1))), RNE
    scale = F32(1.0) / F32((1 << depth) - 1)
    return (px.astype(F32) * scale).astype(np.float16).view(np.uint16)


def rgba8_to_f16(px, attenuate):                      # imagebit/Rgba8ToF16.cpp:44-138
    p = associate8(px) if attenuate else px
    scale = F32(1.0) / F32(255)
    return (p.astype(F32) * scale).astype(np.float16).view(np.uint16)


def rgba16_to_8(px, depth):                           # imagebit/Rgba16.cpp:32-68: >> (depth - 8)
    return (px.astype(np.uint32) >> (depth - 8)).astype(np.uint8)


def rgba8_to_565(px, attenuate):                      # imagebit/Rgb565.cpp:99-128
    p = (associate8(px) if attenuate else px).astype(np.uint16)
    return (((p[..., 0] >> 3) << 11) | ((p[..., 1] >> 2) << 5) | (p[..., 2] >> 3)).astype(np.uint16)


def rgba16_to_565(px, depth):                         # Rgb565.cpp:130-160 (16-bit arithmetic: the shifted-out high bits of red are lost)
    p = px.astype(np.uint32)
    rb, gd = depth - 8 + 3, depth - 8 + 2
    r = ((p[..., 0] >> rb) << 11) & 0xFFFF
    g = ((p[..., 1] >> gd) << 5) & 0xFFFF
    return (r | g | (p[..., 2] >> rb)).astype(np.uint16)


def rgba8_to_1010102(px, attenuate):                  # imagebit/Rgb1010102.cpp:177-213
    p = (associate8(px) if attenuate else px).astype(np.uint32)
    return ((p[..., 3] >> 6) << 30) | ((p[..., 2] << 2) << 20) | ((p[..., 1] << 2) << 10) | (p[..., 0] << 2)


def rgba16_to_1010102(px, depth):                     # Rgb1010102.cpp:215-249
    p = px.astype(np.uint32)
    d, ad = depth - 10, depth - 2
    return (((p[..., 3] >> ad) & 3) << 30) | (((p[..., 2] >> d) & 0x3FF) << 20) | (((p[..., 1] >> d) & 0x3FF) << 10) | ((p[..., 0] >> d) & 0x3FF)


def _aligned_stride(line_bytes, alignment=64):        # ReformatBitmap.cpp:105-107 and siblings
    return line_bytes + (alignment - line_bytes % alignment) % alignment


def reformat(px, config, depth, use_floats, alpha_premultiplied, has_alpha_in_origin, api_level=34):
    """ReformatColorConfig (ReformatBitmap.cpp:46-263).  px: (h, w, 4) u8 or u16.  Returns (rows, stride_bytes, use_floats,
    config_name) with rows a (h, stride) u8 array exactly as the reference's imageData vector."""
    h, w = px.shape[:2]
    if config == DEFAULT:                             # :52-63
        if depth > 8 and api_level >= 26:
            config = RGBA_1010102 if (api_level >= 33 and not has_alpha_in_origin) else RGBA_F16
        else:
            config = RGBA_8888
    if not alpha_premultiplied and has_alpha_in_origin:   # :65-77
        px = associate16(px, depth) if use_floats else associate8(px)
    name = "RGBA_F16" if use_floats else "ARGB_8888"

    def rows_of(arr, stride):
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(h, -1)
        out = np.zeros((h, stride), np.uint8)
        out[:, :raw.shape[1]] = raw
        return out

    if config == RGBA_8888:
        if use_floats:
            return rows_of(rgba16_to_8(px, depth), w * 4), w * 4, False, "ARGB_8888"
        return rows_of(px, w * 4), w * 4, False, name
    if config == RGBA_F16:
        if use_floats:
            return rows_of(u16_to_f16(px, depth), w * 8), w * 8, True, name       # in place: stride unchanged, name unchanged
        s = _aligned_stride(w * 8)
        return rows_of(rgba8_to_f16(px, not alpha_premultiplied), s), s, True, "RGBA_F16"
    if config == RGB_565:
        s = _aligned_stride(w * 2)
        d = rgba16_to_565(px, depth) if use_floats else rgba8_to_565(px, not alpha_premultiplied)
        return rows_of(d, s), s, False, "RGB_565"
    if config == RGBA_1010102:
        s = _aligned_stride(w * 4)
        d = rgba16_to_1010102(px, depth) if use_floats else rgba8_to_1010102(px, not alpha_premultiplied)
        return rows_of(d.astype(np.uint32), s), s, False, "RGBA_1010102"
    if config == HARDWARE:                            # :193-258: RGBA8 copy or F16 conversion into the hardware buffer
        if use_floats:
            return rows_of(u16_to_f16(px, depth), w * 8), w * 8, True, "HARDWARE"
        return rows_of(px, w * 4), w * 4, False, "HARDWARE"
    return rows_of(px, w * 4 * (2 if use_floats else 1)), w * 4 * (2 if use_floats else 1), use_floats, name


# ----------------------------------------------------------------------------------------------- A10
SRGB, ITUR709, GAMMA2P2, SMPTE428, PQ, HLG = "srgb", "709", "gamma2.2", "smpte428", "pq", "hlg"


def _pow(a, e):
    return np.power(a.astype(F32), F32(e), dtype=F32)


def to_linear(v, fn):                                 # colorspaces/Trc.cpp (avifToLinear*)
    v = np.asarray(v, F32)
    if fn == SRGB:                                    # :169-179
        return np.where(v < 0, 0, np.where(v < F32(12.92) * F32(0.0030412825601275209), v / F32(12.92),
                        np.where(v < 1, _pow((v + F32(0.0550107189475866)) / F32(1.0550107189475866), 2.4), 1))).astype(F32)
    if fn == ITUR709:                                 # :31-41
        return np.where(v < 0, 0, np.where(v < F32(4.5) * F32(0.018053968510807), v / F32(4.5),
                        np.where(v < 1, _pow((v + F32(0.09929682680944)) / F32(1.09929682680944), F32(1.0) / F32(0.45)), 1))).astype(F32)
    if fn == GAMMA2P2:                                # :53-55
        return _pow(np.clip(v, 0, 1), 2.2)
    if fn == SMPTE428:                                # :223-225
        return (_pow(np.maximum(v, 0), 2.6) / F32(0.91655527974030934)).astype(F32)
    if fn == PQ:                                      # :197-208
        p = _pow(np.maximum(v, F32(1e-30)), F32(1.0) / F32(78.84375))
        num = np.maximum(p - F32(0.8359375), 0)
        den = np.maximum(F32(18.8515625) - F32(18.6875) * p, np.finfo(F32).tiny)
        lin = _pow(num / den, F32(1.0) / F32(0.1593017578125)) * F32(10000.0) / F32(203.0)
        return np.where(v > 0, lin, 0).astype(F32)
    if fn == HLG:                                     # :235-250
        lo = _pow((v * v) * (F32(1.0) / F32(3.0)), 1.2)
        hi = _pow((np.exp((v - F32(0.55991073)) / F32(0.17883277), dtype=F32) + F32(0.28466892)) / F32(12.0), 1.2)
        return np.where(v < 0, 0, np.where(v <= 0.5, lo, hi) * F32(1000.0) / F32(203.0)).astype(F32)
    raise ValueError(fn)


def to_gamma_srgb(v):                                 # Trc.cpp:180-191
    v = np.asarray(v, F32)
    return np.where(v < 0, 0, np.where(v < F32(0.0030412825601275209), v * F32(12.92),
                    np.where(v < 1, F32(1.0550107189475866) * _pow(np.maximum(v, 0), F32(1.0) / F32(2.4)) - F32(0.0550107189475866), 1))).astype(F32)


PRIMARIES_XY = {                                      # colorspaces/ColorSpaceProfile.h (getSRGBPrimaries, getDisplayP3Primaries, getRec2020Primaries)
    1: (0.640, 0.330, 0.300, 0.600, 0.150, 0.060),
    11: (0.680, 0.320, 0.265, 0.690, 0.150, 0.060),
    9: (0.708, 0.292, 0.170, 0.797, 0.131, 0.046),
}
D65 = (0.3127, 0.3290)


def gamut_rgb_to_xyz(prim, white):                    # ColorSpaceProfile.h:131-143 (float32 like Eigen::Matrix3f)
    rx, ry, gx, gy, bx, by = [F32(t) for t in prim]
    wx, wy = F32(white[0]), F32(white[1])
    m = np.array([[rx / ry, gx / gy, bx / by], [1, 1, 1], [(1 - rx - ry) / ry, (1 - gx - gy) / gy, (1 - bx - by) / by]], F32)
    wxyz = np.array([wx / wy, 1, (1 - wx - wy) / wy], F32)
    s = np.linalg.inv(m.astype(np.float64)).astype(F32) @ wxyz
    return (m * s[None, :]).astype(F32)


def conversion_matrix(primaries, xy=None):            # JniDecoding.cpp:166-199
    if primaries in PRIMARIES_XY:
        prim, white = PRIMARIES_XY[primaries], D65
    else:
        prim, white = tuple(xy[:6]), tuple(xy[6:8])
    src = gamut_rgb_to_xyz(prim, white)
    dst = gamut_rgb_to_xyz(PRIMARIES_XY[1], D65)      # Rec.709 primaries == sRGB primaries
    return (np.linalg.inv(dst.astype(np.float64)) @ src.astype(np.float64)).astype(F32)


TF_OF_JXL = {18: (HLG, True), 17: (SMPTE428, False), 16: (PQ, True), 65535: (GAMMA2P2, False), 1: (ITUR709, False), 13: (SRGB, False)}


def color_matrix(px, depth, primaries, tf, xy=None, intensity_target=255.0):
    """applyColorMatrix / applyColorMatrix16Bit with the call-site set-up of JniDecoding.cpp:138-228.  px (h, w, 4) u8/u16."""
    fn, tone = TF_OF_JXL[tf]
    m = conversion_matrix(primaries, xy)
    is16 = px.dtype == np.uint16
    if is16:                                          # ColorMatrix.cpp:141-157
        n = 1 << depth
        cut = F32(n - 1)
        lin_lut = to_linear(np.arange(n, dtype=F32) * (F32(1.0) / cut), fn)
        gam_lut = np.clip(np.round(to_gamma_srgb(np.arange(n, dtype=F32) * (F32(1.0) / cut)) * cut), 0, cut).astype(np.uint16)
        idx_scale, idx_max = cut, n - 1
        src = np.minimum(px[..., :3], n - 1)
    else:                                             # :49-60
        lin_lut = to_linear(np.arange(256, dtype=F32) * (F32(1.0) / F32(255.0)), fn)
        gam_lut = np.clip(np.round(to_gamma_srgb(np.arange(2049, dtype=F32) * (F32(1.0) / F32(2048.0))) * F32(255.0)), 0, 255).astype(np.uint8)
        idx_scale, idx_max = F32(2048.0), 2048
        src = px[..., :3]
    rgb = lin_lut[src]                                # (h, w, 3) f32
    if tone:                                          # Rec2408ToneMapper.h:36-45, .cpp:80-100 (display 250 nits, white 203)
        ld = F32(intensity_target) / F32(203.0)
        wa = (F32(250.0) / F32(203.0)) / (ld * ld)
        wb = F32(1.0) / (F32(250.0) / F32(203.0))
        y = F32(0.2627) * rgb[..., 0] + F32(0.6780) * rgb[..., 1] + F32(0.0593) * rgb[..., 2]
        scale = (F32(1.0) + wa * y) / (F32(1.0) + wb * y)
        mapped = np.minimum(rgb * scale[..., None], F32(1.0))
        # quirk (.cpp:91-93): a pixel with Y == 0 `continue`s WITHOUT advancing the row pointer, so the loop stays on that
        # pixel until the row ends: everything from the first zero-luma pixel of a row onwards is left un-mapped.
        zero = (y == 0)
        first_zero = np.where(zero.any(axis=1), zero.argmax(axis=1), px.shape[1])
        keep = np.arange(px.shape[1])[None, :] < first_zero[:, None]
        rgb = np.where(keep[..., None], mapped, rgb).astype(F32)
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    out = np.stack([r * m[0, 0] + g * m[0, 1] + b * m[0, 2], r * m[1, 0] + g * m[1, 1] + b * m[1, 2], r * m[2, 0] + g * m[2, 1] + b * m[2, 2]], -1).astype(F32)
    idx = np.minimum((np.clip(out, 0, 1) * idx_scale).astype(np.uint16), idx_max)
    res = px.copy()
    res[..., :3] = gam_lut[idx]
    return res
