"""The drop-in boundary to the letter: out-params of DecodeJpegXlOneShot incl. the ICC bytes and the chromaticities of
JxlColorEncoding, the INT32_MAX size guard and its message, checkDecodePreconditions' API-level gates, and the reference-side binding
of INTEGRATION.md compiled against the reference's own header (build container only)."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_case

REF_CPP = "/root/reference/jxlcoder/src/main/cpp"
STUB_SO = os.path.join(ROOT, "tests", "boundary", "libjxldecoding_amd.so")


def _bits(fields):
    """LSB-first bit packer: [(value, nbits)] -> bytes"""
    acc = n = 0
    for v, k in fields:
        acc |= (v & ((1 << k) - 1)) << n
        n += k
    return acc.to_bytes((n + 7) // 8 + 8, "little")


def huge_header(side=40000):
    """A bare codestream header for a side x side image: signature, SizeHeader (not small, 18-bit height, ratio 1:1), all-default metadata."""
    return _bits([(0x0AFF, 16), (0, 1), (2, 2), (side - 1, 18), (1, 3), (1, 1), (1, 1)])


def build_stub():
    import jxl_coder_amd as J
    src = os.path.join(ROOT, "tests", "boundary", "jxl_decoding_amd.cpp")
    if not os.path.isdir(REF_CPP):
        return os.path.exists(STUB_SO)
    # -include cstring: the reference's header calls strdup and relies on bionic's headers pulling <cstring> in
    if not os.path.exists(STUB_SO) or os.path.getmtime(STUB_SO) < max(os.path.getmtime(src), os.path.getmtime(J.library_path())):
        subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-include", "cstring", "-include", "cstdint", "-I", REF_CPP, "-I", os.path.join(REF_CPP, "jxl"), "-I", os.path.join(ROOT, "include"),
                        "-o", STUB_SO, src, J.library_path(), "-Wl,-rpath," + os.path.dirname(J.library_path())], check=True)
    return True


def stub():
    if not build_stub():
        pytest.skip("the compiled DecodeJpegXlOneShot binding is built in the build container (needs the reference's header)")
    from jxl_coder_amd import api
    api.lib()                  # loads libjxlamd.so the way the package does (torch's HIP runtime first when a GPU is present)
    L = C.CDLL(STUB_SO)
    L.boundary_basic_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64 * 2)]
    L.boundary_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64 * 12), C.POINTER(C.c_double * 8), C.c_char_p, C.c_size_t]
    return L


def test_size_guard_and_its_message():
    """interop/JxlDecoding.cpp:103-109 + JxlDecoding.h:38-52: w*h*4*bytes >= INT32_MAX throws InvalidImageSizeException."""
    import jxl_coder_amd as J
    from jxl_coder_amd import api
    data = huge_header()
    assert J.JxlCoder.getSize(data) == (40000, 40000)
    n = C.c_size_t()
    assert api.lib().jxlamd_output_size(data, len(data), 0, C.byref(n)) == -3          # JXLAMD_ERR_SIZE
    assert api.lib().jxlamd_last_error(None).decode() == "Invalid image size exceed allowance, current size w: 40000, h: 40000"
    assert api.lib().jxlamd_output_size(data, len(data), api.JXLAMD_NO_SIZE_GUARD, C.byref(n)) == 0 and n.value == 40000 * 40000 * 4
    ok = _bits([(0x0AFF, 16), (0, 1), (2, 2), (23169 - 1, 18), (1, 3), (1, 1), (1, 1)])   # 23169^2 * 4 = 2147210244 < INT32_MAX
    assert api.lib().jxlamd_output_size(ok, len(ok), 0, C.byref(n)) == 0
    with pytest.raises(J.InvalidImageSizeException, match="w: 40000, h: 40000"):
        J.JxlDecoder.__new__(J.JxlDecoder).decode_one_shot(data)                         # raised before any device work


def test_api_level_gates_of_check_decode_preconditions():
    """cpp/Support.cpp:35-92, same order and messages."""
    from jxl_coder_amd.api import _check_preconditions, PreferredColorConfig as P, ScaleMode as S
    _check_preconditions(P.RGBA_1010102, S.FIT, 6, 33)
    with pytest.raises(ValueError, match="Invalid Color Config: 0 was passed"):
        _check_preconditions(0, S.FIT)
    with pytest.raises(ValueError, match="RGBA_1010102 supported only 33\\+ OS version but current is: 32"):
        _check_preconditions(P.RGBA_1010102, S.FIT, 6, 32)
    with pytest.raises(ValueError, match="supported only 26\\+ OS version but current is: 25"):
        _check_preconditions(P.RGBA_F16, S.FIT, 6, 25)
    with pytest.raises(ValueError, match="HARDWARE supported only 29\\+ OS version but current is: 28"):
        _check_preconditions(P.HARDWARE, S.FIT, 6, 28)
    with pytest.raises(ValueError, match="Invalid Scale Mode was passed"):
        _check_preconditions(P.DEFAULT, 0)
    with pytest.raises(ValueError, match="Invalid Sampler: 0 was passed"):
        _check_preconditions(P.DEFAULT, S.FIT, 0)


def test_icc_bytes_and_colour_encoding_fields(golden_meta):
    """The embedded ICC of a non-XYB file comes back byte for byte (what the reference's libjxl returns for TARGET_DATA), with
    have_encoded_profile = prefer_encoding = 0; enum profiles carry the chromaticities libjxl reports."""
    from jxl_coder_amd import api
    L = api.lib()
    L.jxlamd_get_icc.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    data = open(os.path.join(GOLDEN, "licc96x64_e3.jxl"), "rb").read()
    info = api.Info()
    assert L.jxlamd_basic_info(data, len(data), C.byref(info)) == 0
    meta = golden_meta["licc96x64_e3"]
    assert (info.have_encoded_profile, info.prefer_encoding, info.uses_original_profile, info.icc_size) == (0, 0, 1, meta["icc_size"])
    buf = np.zeros(info.icc_size, np.uint8); n = C.c_size_t()
    assert L.jxlamd_get_icc(data, len(data), buf.ctypes.data, buf.size, C.byref(n)) == 0 and n.value == meta["icc_size"]
    assert hashlib.sha256(buf.tobytes()).hexdigest() == meta["icc_sha256"]
    assert L.jxlamd_get_icc(data, len(data), buf.ctypes.data, 16, C.byref(n)) == -5                     # JXLAMD_ERR_BUFFER
    # XYB image with an embedded profile: the reference's libjxl (no CMS) outputs sRGB and reports the sRGB enum profile
    asset = open(os.path.join(GOLDEN, "asset_jxl_icc_12bit.jxl"), "rb").read()
    assert L.jxlamd_basic_info(asset, len(asset), C.byref(info)) == 0
    ref = golden_meta["asset_jxl_icc_12bit"]["info"]
    for k in ("have_encoded_profile", "prefer_encoding", "color_space", "white_point", "primaries", "transfer_function", "bits_per_sample"):
        assert getattr(info, k) == ref[k], k
    assert info.icc_size == 0
    assert np.allclose(list(info.white_point_xy) + list(info.primaries_red_xy) + list(info.primaries_green_xy) + list(info.primaries_blue_xy), ref["xy"], atol=1e-9)
    pq, _ = load_case("v160x120_16bit_pq2100_epf3")
    assert L.jxlamd_basic_info(pq, len(pq), C.byref(info)) == 0
    assert np.allclose(list(info.primaries_red_xy) + list(info.primaries_green_xy) + list(info.primaries_blue_xy), [0.708, 0.292, 0.170, 0.797, 0.131, 0.046])


def test_compiled_reference_binding_on_the_host():
    """DecodeBasicInfo / the size guard of DecodeJpegXlOneShot through the compiled binding (no GPU needed for either)."""
    L = stub()
    data, _ = load_case("v264x520_e7")
    wh = (C.c_uint64 * 2)()
    assert L.boundary_basic_info(data, len(data), C.byref(wh)) == 1 and tuple(wh) == (264, 520)
    assert L.boundary_basic_info(b"nope", 4, C.byref(wh)) == 0
    big = huge_header()
    msg = C.create_string_buffer(256)
    meta = (C.c_uint64 * 12)(); xy = (C.c_double * 8)()
    rc = L.boundary_decode(big, len(big), 1, None, 0, C.byref(meta), C.byref(xy), msg, 256)
    assert rc == -3 and msg.value.decode() == "Invalid image size exceed allowance, current size w: 40000, h: 40000"


@pytest.mark.gpu
def test_compiled_reference_binding_decodes():
    import jxl_coder_amd as J
    L = stub()
    dec = J.JxlDecoder(0)
    for name, allowed in (("v264x520_e7", 1), ("v160x120_16bit_pq2100_epf3", 1), ("v160x120_16bit_pq2100_epf3", 0), ("va300x520_e7", 1)):
        data, exp = load_case(name)
        want, info = dec.decode_one_shot(data, allowed_floats=bool(allowed))
        out = np.zeros(want.nbytes, np.uint8); meta = (C.c_uint64 * 12)(); xy = (C.c_double * 8)(); msg = C.create_string_buffer(256)
        assert L.boundary_decode(data, len(data), allowed, out.ctypes.data, out.size, C.byref(meta), C.byref(xy), msg, 256) == 1
        assert np.array_equal(out.view(want.dtype).reshape(want.shape), want)
        assert (meta[0], meta[1]) == (want.shape[1], want.shape[0]) and meta[2] == int(want.dtype == np.uint16) and meta[3] == info["out_bits"]
        assert meta[5] == 1 and meta[6] == info["prefer_encoding"] and meta[7] == info["has_alpha_in_origin"] and meta[10] == 0
        assert (meta[8], meta[9]) == (info["primaries"], info["transfer_function"])
    dec.close()


@pytest.mark.gpu
def test_files_with_embedded_icc_decode(golden_meta):
    import jxl_coder_amd as J
    dec = J.JxlDecoder(0)
    data, exp = load_case("licc96x64_e3")                       # lossless, non-XYB: samples stay in the profile's space, bit-exact
    out, info = dec.decode_one_shot(data)
    assert np.array_equal(out, exp) and info["prefer_encoding"] == 0 and info["icc_size"] == golden_meta["licc96x64_e3"]["icc_size"]
    asset = open(os.path.join(GOLDEN, "asset_jxl_icc_12bit.jxl"), "rb").read()    # XYB + ICC (reference demo asset): decodes to sRGB
    out, info = dec.decode_one_shot(asset, allowed_floats=True)
    ref = golden_meta["asset_jxl_icc_12bit"]
    assert list(out.shape) == ref["shape"] and out.dtype == np.uint16
    rs = out.astype(np.int64).sum(axis=(1, 2))
    assert np.abs(rs - np.array(ref["row_sums"])).max() <= 16.0 * out.shape[1] * 4          # mean |diff| <= 16/65535 per sample (conftest U16_MEAN_ABS), row by row
    dec.close()


@pytest.mark.gpu
def test_icc_stage_matches_little_cms(golden_meta):
    """A8 convertUseDefinedColorSpace on the device (jxlamd_icc_transform: 129^3 lattice sampled from Little CMS with the reference's intent and
    flags, trilinear on the GPU) against Little CMS itself run per pixel with the reference's parameters (oracle/icc_oracle.py).
    Stated tolerance: u8 max |diff| <= 2, mean <= 0.25; u16 max <= 512/65535, mean <= 48/65535 (lattice interpolation vs the library's own
    16-bit pipeline)."""
    import torch
    import jxl_coder_amd as J
    from jxl_coder_amd import api
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import icc_oracle
    if not icc_oracle.available():
        pytest.skip("liblcms2.so.2 not present on this box")
    data, exp = load_case("licc96x64_e3")
    icc = np.zeros(golden_meta["licc96x64_e3"]["icc_size"], np.uint8); n = C.c_size_t()
    assert api.lib().jxlamd_get_icc(data, len(data), icc.ctypes.data, icc.size, C.byref(n)) == 0
    icc = icc.tobytes()
    dec = J.JxlDecoder(0)
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:64, 0:96]
    for is16 in (False, True):
        maxv = 65535 if is16 else 255
        img = np.stack([xx / 95 * maxv, yy / 63 * maxv, rng.uniform(0, maxv, (64, 96)), np.full((64, 96), maxv)], -1).astype(np.uint16 if is16 else np.uint8)
        img[:8, :8, :3] = 0; img[-8:, -8:, :3] = maxv
        want = icc_oracle.convert(img, icc)
        buf = torch.from_numpy(img.view(np.uint8).reshape(-1).copy()).cuda()
        dec.icc_transform_device(buf.data_ptr(), 96, 64, is16, icc)
        got = buf.cpu().numpy().view(img.dtype).reshape(img.shape)
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        assert np.array_equal(got[..., 3], img[..., 3])
        assert (want != img).any()                                                   # the profile (Rec.2100 PQ) really changes the pixels
        if is16:
            assert d.max() <= 512 and d.mean() <= 48, (d.max(), d.mean())
        else:
            assert d.max() <= 1 and d.mean() <= 0.01, (d.max(), d.mean())      # 8-bit images: the lattice holds the 8-bit transform itself (<= 1: the box's Little CMS 2.12 vs the checker's 2.16)
    # end to end: JxlCoder.decode of the lossless + ICC file runs the stage (the ICC vector is non-empty, preferEncoding false)
    px = J.JxlCoder.decode(data, J.PreferredColorConfig.RGBA_8888)
    ref = icc_oracle.convert(exp, icc)
    d = np.abs(px.astype(int) - ref.astype(int))
    assert d.max() <= 2 and d.mean() <= 0.25
    dec.close()


LINEAR_CASES = ["vlin96x64_e3", "vlin2100_96x64_e3", "vlinp3_96x64_e3", "llin96x64_e3", "vlingrey96x64_e3"]


@pytest.mark.parametrize("name", LINEAR_CASES)
def test_synthesised_profile_of_linear_enum_encodings_equals_libjxls(name, golden_meta):
    """The reference does not treat the LINEAR transfer function as a preferred encoding: DecodeJpegXlOneShot asks libjxl for the data profile
    (which libjxl synthesises for an enum encoding) and fails the decode without it (interop/JxlDecoding.cpp:126-141); the JNI layer then converts
    through Little CMS.  jxlamd_get_icc returns that profile (host_icc_synth.inc) — byte for byte what the reference's libjxl returned for the
    same file (ICC v4.4, 'para' curves, Bradford chad, cicp, MD5 profile ID) — and jxlamd_basic_info reports its size with prefer_encoding 0."""
    from jxl_coder_amd import api
    data = open(os.path.join(GOLDEN, name + ".jxl"), "rb").read()
    want = open(os.path.join(GOLDEN, name + ".icc"), "rb").read()
    info = api.Info()
    assert api.lib().jxlamd_basic_info(data, len(data), C.byref(info)) == 0
    assert info.prefer_encoding == 0 and info.have_encoded_profile == 1 and info.transfer_function == 8
    assert info.icc_size == len(want) == golden_meta[name]["icc_size"]
    buf = np.zeros(info.icc_size, np.uint8); n = C.c_size_t()
    assert api.lib().jxlamd_get_icc(data, len(data), buf.ctypes.data, buf.size, C.byref(n)) == 0
    assert buf[: n.value].tobytes() == want


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["vlin96x64_e3", "llin96x64_e3"])
def test_linear_enum_image_goes_through_a8_like_the_reference(name):
    """End to end for a linear-light enum encoding: the decoder proper returns the data (linear) pixels the reference's libjxl returns, and
    JxlCoder.decode then runs stage A8 with the synthesised profile — linear -> sRGB through the Little CMS lattice — as the reference's JNI layer
    does with libjxl's profile (JniDecoding.cpp:103-114); checked against Little CMS run per pixel on the reference's pixels and profile."""
    import jxl_coder_amd as J
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import icc_oracle
    if not icc_oracle.available():
        pytest.skip("Little CMS not present on this box")
    data, exp = load_case(name)
    icc = open(os.path.join(GOLDEN, name + ".icc"), "rb").read()
    dec = J.JxlDecoder(0)
    raw, info = dec.decode_one_shot(data)
    d = np.abs(raw.astype(int) - exp.astype(int))
    assert (d.max() == 0) if name.startswith("l") else (d.max() <= 1 and d.mean() <= 0.08), (d.max(), d.mean())      # effort-3 file with one EPF iteration, linear-light codes: the rcpps offset of conftest.py (measured 0.051)
    assert info["prefer_encoding"] == 0 and info["icc_size"] == len(icc)
    px = J.JxlCoder.decode(data, J.PreferredColorConfig.RGBA_8888)
    # Little CMS on the DECODED linear pixels: one 8-bit linear code near black spans up to 13 sRGB codes, so the decoder's +-1 must not enter
    # this comparison — it is the A8 stage (lattice vs the library per pixel) that is checked here, with its own tolerance
    want = icc_oracle.convert(raw, icc)
    assert (want != raw).any()                                     # linear -> sRGB really changes the pixels
    d = np.abs(px.astype(int) - want.astype(int))
    assert d.max() <= 1 and d.mean() <= 0.01, (d.max(), d.mean())      # 8-bit images: the lattice holds the 8-bit transform itself (<= 1: the box's Little CMS 2.12 vs the checker's 2.16)
    dec.close()
