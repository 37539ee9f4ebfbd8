"""Post-decode stages: the reference's applyColorMatrix (cpp/colorspaces/ColorMatrix.cpp) and
ReformatColorConfig (cpp/ReformatBitmap.cpp) with the imagebit kernels.

CPU part: the numpy oracle (oracle/post_oracle.py) against golden outputs of the reference's own sources
(tests/golden/post_golden.npz, made by tests/golden/make_post_golden.py through oracle/_ref/libref_post.so).
GPU part: the HIP kernels behind jxlamd_reformat / jxlamd_color_matrix against the oracle.

Tolerances: every integer stage is BIT-EXACT.  The colour-matrix stage goes through LUTs built with powf (256 + 2049 entries
for u8, 2 x 65 536 for u16); numpy's float32 pow, libm's powf and the float matrix inverse differ in the last ulp, which moves
a sample sitting on a LUT-index boundary by one LUT step.  u8: at most 0.2 % of samples differ, by <= 2; u16: at most 2 %
differ, by <= 16/65535 (one step of the sRGB LUT at its steepest)."""
import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import post_oracle as P   # noqa: E402  (test infrastructure)

G = np.load(os.path.join(ROOT, "tests/golden/post_golden.npz"))
CM_CASES = [(9, 16, 10000.0), (9, 18, 1000.0), (1, 13, 255.0), (11, 13, 255.0), (1, 1, 255.0), (2, 65535, 255.0), (11, 17, 255.0)]
U16_LUT_MAX, U16_LUT_FRAC = 16, 0.02
U8_LUT_MAX, U8_LUT_FRAC = 2, 0.002
GPU_U16_LUT_MAX = 64       # GPU vs oracle: three independent pow implementations (numpy, host libm, reference) meet in the PQ case; <= 0.1 % of full scale


def test_oracle_integer_stages_match_the_reference_bit_for_bit():
    p8, p16 = G["p8"], G["p16"]
    assert np.array_equal(P.associate8(p8), G["associate8"])
    assert np.array_equal(P.associate16(p16, 16), G["associate16"])
    assert np.array_equal(P.u16_to_f16(p16, 16), G["u16_to_f16"])
    assert np.array_equal(P.rgba16_to_8(p16, 16), G["rgba16_to_8"])
    assert np.array_equal(P.rgba16_to_565(p16, 16), G["rgba16_to_565"])
    assert np.array_equal(P.rgba16_to_1010102(p16, 16), G["rgba16_to_1010102"])
    for att in (0, 1):
        assert np.array_equal(P.rgba8_to_f16(p8, att), G[f"rgba8_to_f16_{att}"])
        assert np.array_equal(P.rgba8_to_565(p8, att), G[f"rgba8_to_565_{att}"])
        assert np.array_equal(P.rgba8_to_1010102(p8, att), G[f"rgba8_to_1010102_{att}"])


@pytest.mark.parametrize("prim,tf,target", CM_CASES)
def test_oracle_colour_matrix_matches_the_reference(prim, tf, target):
    xy = list(G["custom_xy"])
    assert np.abs(P.conversion_matrix(prim, xy).ravel() - G[f"cm_matrix_{prim}"]).max() < 2e-6
    d8 = np.abs(P.color_matrix(G["p8"], 8, prim, tf, xy, target).astype(int) - G[f"cm_{prim}_{tf}_8"].astype(int))   # rows with zero-luma pixels included
    assert d8.max() <= U8_LUT_MAX and (d8 > 0).mean() <= U8_LUT_FRAC
    d = np.abs(P.color_matrix(G["p16"], 16, prim, tf, xy, target).astype(int) - G[f"cm_{prim}_{tf}_16"].astype(int))
    assert d.max() <= U16_LUT_MAX and (d > 0).mean() <= U16_LUT_FRAC


def test_oracle_against_live_reference_library_when_present():
    import ctypes as C
    path = os.path.join(ROOT, "oracle/_ref/libref_post.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_post.so not built")
    L = C.CDLL(path)
    rng = np.random.default_rng(7)
    for (h, w) in ((1, 1), (3, 17), (5, 64)):
        p8 = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        a = p8.copy(); L.refpost_associate8(C.c_void_p(a.ctypes.data), w * 4, w, h)
        assert np.array_equal(a, P.associate8(p8))
        d = np.zeros((h, w), np.uint16); L.refpost_rgba8_to_565(C.c_void_p(p8.ctypes.data), w * 4, C.c_void_p(d.ctypes.data), w * 2, w, h, 1)
        assert np.array_equal(d, P.rgba8_to_565(p8, True))


def test_reformat_flow_mirrors_reformatcolorconfig():
    """cpp/ReformatBitmap.cpp:46-263: DEFAULT resolution, premultiply rule, 64-byte row alignment, config names."""
    rng = np.random.default_rng(3)
    p8 = rng.integers(0, 256, (4, 37, 4), dtype=np.uint8); p16 = rng.integers(0, 65536, (4, 37, 4), dtype=np.uint16)
    rows, stride, fl, name = P.reformat(p8, P.DEFAULT, 8, False, False, False, 34)
    assert (stride, fl, name) == (37 * 4, False, "ARGB_8888") and np.array_equal(rows.reshape(4, 37, 4), p8)
    rows, stride, fl, name = P.reformat(p16, P.DEFAULT, 16, True, False, False, 34)             # >8 bit, no alpha, API >= 33
    assert (stride, name) == (192, "RGBA_1010102") and stride % 64 == 0
    rows, stride, fl, name = P.reformat(p16, P.DEFAULT, 16, True, False, True, 34)              # alpha in origin -> F16, premultiplied first
    assert (stride, fl, name) == (37 * 8, True, "RGBA_F16")
    assert np.array_equal(rows.view(np.uint16).reshape(4, 37, 4), P.u16_to_f16(P.associate16(p16, 16), 16))
    rows, stride, fl, name = P.reformat(p16, P.DEFAULT, 16, True, False, False, 25)             # API < 26
    assert name == "ARGB_8888" and np.array_equal(rows.reshape(4, 37, 4), P.rgba16_to_8(p16, 16))
    rows, stride, fl, name = P.reformat(p8, P.RGBA_F16, 8, False, False, True, 34)              # the reference attenuates twice here
    assert stride == 320 and np.array_equal(rows[:, :37 * 8].copy().view(np.uint16).reshape(4, 37, 4), P.rgba8_to_f16(P.associate8(p8), True))
    rows, stride, fl, name = P.reformat(p8, P.RGB_565, 8, False, True, True, 34)
    assert (stride, name) == (128, "RGB_565") and np.array_equal(rows[:, :74].copy().view(np.uint16).reshape(4, 37), P.rgba8_to_565(p8, False))


# ------------------------------------------------------------------------------------------------- GPU
def _dev(arr):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()).cuda()
    torch.cuda.synchronize()
    return t


@pytest.fixture(scope="module")
def dec():
    import torch
    torch.cuda.init()                      # torch first: it owns the device allocations of these tests
    import jxl_coder_amd as J
    d = J.JxlDecoder(0)
    yield d
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("is16", [False, True])
@pytest.mark.parametrize("config", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("premult,has_alpha", [(False, False), (False, True), (True, True)])
def test_gpu_reformat_is_bit_exact(dec, is16, config, premult, has_alpha):
    import torch
    rng = np.random.default_rng(11 + config)
    h, w = 19, 203                                                      # rows not a multiple of 64 bytes in any format
    px = rng.integers(0, 65536 if is16 else 256, (h, w, 4), dtype=np.uint16 if is16 else np.uint8)
    px[3, :, 3] = 0
    depth = 16 if is16 else 8
    for api in (34, 29):
        exp_rows, exp_stride, exp_fl, exp_name = P.reformat(px, config, depth, is16, premult, has_alpha, api)
        src = _dev(px)
        ri = dec.reformat_query(w, h, is16, config, has_alpha, api)
        assert ri.stride == exp_stride and ri.bytes == exp_stride * h and bool(ri.use_floats) == exp_fl
        dst = torch.full((int(ri.bytes),), 0xAB, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()             # torch's fill runs on torch's stream, the decoder context has its own
        dec.reformat_device(src.data_ptr(), w, h, is16, depth, config, premult, has_alpha, api, dst.data_ptr(), dst.numel())
        assert np.array_equal(dst.cpu().numpy().reshape(h, exp_stride), exp_rows)


@pytest.mark.gpu
@pytest.mark.parametrize("prim,tf,target", CM_CASES)
def test_gpu_colour_matrix(dec, prim, tf, target):
    rng = np.random.default_rng(5)
    h, w = 33, 517
    xy = list(G["custom_xy"])
    for is16 in (False, True):
        px = rng.integers(0, 65536 if is16 else 256, (h, w, 4), dtype=np.uint16 if is16 else np.uint8)
        px[2, 100, :3] = 0; px[5, 0, :3] = 0; px[7, w - 1, :3] = 0        # zero-luma pixels: the rest of those rows stays un-mapped
        exp = P.color_matrix(px, 16 if is16 else 8, prim, tf, xy, target)
        buf = _dev(px)
        dec.color_matrix_device(buf.data_ptr(), w, h, is16, 16 if is16 else 8, prim, tf, target, xy)
        got = buf.cpu().numpy().view(np.uint16 if is16 else np.uint8).reshape(h, w, 4)
        d = np.abs(got.astype(int) - exp.astype(int))
        if is16:
            assert d.max() <= GPU_U16_LUT_MAX and (d > 0).mean() <= U16_LUT_FRAC
        else:
            assert d.max() <= U8_LUT_MAX and (d > 0).mean() <= U8_LUT_FRAC
        assert np.array_equal(got[..., 3], px[..., 3])


@pytest.mark.gpu
def test_gpu_colour_matrix_skips_what_the_reference_skips(dec):
    px = np.random.default_rng(1).integers(0, 256, (4, 9, 4), dtype=np.uint8)
    buf = _dev(px)
    dec.color_matrix_device(buf.data_ptr(), 9, 4, False, 8, 1, 8, 255.0)          # linear transfer: stage not run (JniDecoding.cpp:131-137)
    assert np.array_equal(buf.cpu().numpy().reshape(4, 9, 4), px)


@pytest.mark.gpu
def test_gpu_decode_bitmap_pipeline_on_pq_image():
    """C5-shaped path end to end on the device: 16-bit PQ / Rec.2100 decode -> tone map + gamut + sRGB (API < 34) -> F16 / 1010102."""
    import torch
    import jxl_coder_amd as J
    from conftest import load_case
    data, _ = load_case("v160x120_16bit_pq2100_epf3")
    raw, info = J.JxlCoder._decoder().decode_one_shot(data, allowed_floats=True)
    assert raw.dtype == np.uint16 and info["transfer_function"] == 16 and info["primaries"] == 9
    for api, cfg in ((34, J.PreferredColorConfig.DEFAULT), (29, J.PreferredColorConfig.DEFAULT), (29, J.PreferredColorConfig.RGBA_F16),
                     (29, J.PreferredColorConfig.RGBA_8888), (33, J.PreferredColorConfig.RGB_565)):
        bmp = J.JxlCoder.decodeBitmap(data, cfg, api_level=api)
        px = raw
        if api < 34:
            px = P.color_matrix(raw, 16, info["primaries"], 16, None, info["intensity_target"])
        rows, stride, fl, name = P.reformat(px, int(cfg), 16, True, bool(info["alpha_premultiplied"]), bool(info["has_alpha_in_origin"]), api)
        assert (bmp.stride, bmp.use_floats, bmp.config) == (stride, fl, name)
        if api >= 34:
            assert np.array_equal(bmp.rows, rows)
        else:                                                            # one LUT step of the u16 colour matrix, then an exact reformat
            assert (bmp.rows != rows).mean() <= 0.03


@pytest.mark.gpu
def test_gpu_decode_bitmap_premultiplies_real_alpha():
    """RGBA VarDCT file (alpha in origin, not premultiplied): ReformatColorConfig associates alpha before the writers
    (cpp/ReformatBitmap.cpp:65-77) — on decoded pixels, bit for bit against the oracle's reformat of the same decode."""
    import jxl_coder_amd as J
    from conftest import load_case
    for name, is16 in (("va300x520_e7", False), ("va530x270_16bit_e7", True)):
        data, _ = load_case(name)
        raw, info = J.JxlCoder._decoder().decode_one_shot(data, allowed_floats=True)
        assert info["has_alpha_in_origin"] == 1 and info["alpha_premultiplied"] == 0 and raw[..., 3].min() == 0
        for cfg in (J.PreferredColorConfig.DEFAULT, J.PreferredColorConfig.RGBA_8888, J.PreferredColorConfig.RGBA_F16, J.PreferredColorConfig.RGBA_1010102):
            bmp = J.JxlCoder.decodeBitmap(data, cfg, api_level=34)
            rows, stride, fl, cfgname = P.reformat(raw, int(cfg), 16 if is16 else 8, is16, False, True, 34)
            assert (bmp.stride, bmp.use_floats, bmp.config) == (stride, fl, cfgname)
            assert np.array_equal(bmp.rows, rows)


@pytest.mark.gpu
def test_fused_post_stage_equals_the_two_stage_form():
    """jxlamd_post_fused (A10 + A11 in one pass) against jxlamd_color_matrix followed by jxlamd_reformat on the same
    device buffers: bit for bit, for every colour-matrix case (tone-mapped rows with zero-luma "stuck" pixels included: G["p8"] / G["p16"]
    carry them), every target format, straight and premultiplied alpha, with and without the matrix stage."""
    import torch
    import jxl_coder_amd as J
    dec = J.JxlDecoder(0)
    rng = np.random.default_rng(11)
    h, w = 37, 301
    imgs = {False: rng.integers(0, 256, (h, w, 4), dtype=np.uint8), True: rng.integers(0, 65536, (h, w, 4), dtype=np.uint16)}
    for v in imgs.values():
        v[5, 40:44, :3] = 0; v[9, 0, :3] = 0; v[20, w - 1, :3] = 0      # zero-luma pixels: the tone mapper's loop sticks there for the rest of the row
    checked = 0
    for is16, img in imgs.items():
        depth = 16 if is16 else 8
        for cfg in (J.PreferredColorConfig.RGBA_8888, J.PreferredColorConfig.RGBA_F16, J.PreferredColorConfig.RGB_565, J.PreferredColorConfig.RGBA_1010102,
                    J.PreferredColorConfig.HARDWARE, J.PreferredColorConfig.DEFAULT):
            for prim, tf, target in ((9, 16, 10000.0), (11, 13, 255.0), (1, 13, 255.0), (None, None, None)):
                for premult, has_alpha in ((False, True), (True, True), (False, False)):
                    src = torch.from_numpy(img.copy()).cuda()
                    ri = dec.reformat_query(w, h, is16, cfg, has_alpha, 33)
                    two = torch.zeros(int(ri.bytes), dtype=torch.uint8, device="cuda")
                    one = torch.zeros(int(ri.bytes), dtype=torch.uint8, device="cuda")
                    src2 = torch.from_numpy(img.copy()).cuda()
                    torch.cuda.synchronize()          # torch's fill kernels run on torch's stream, the stages on the decoder's own: order them
                    if prim is not None:
                        dec.color_matrix_device(src.data_ptr(), w, h, is16, depth, prim, tf, target)
                    dec.reformat_device(src.data_ptr(), w, h, is16, depth, cfg, premult, has_alpha, 33, two.data_ptr(), two.numel())
                    ri2 = dec.post_fused_device(src2.data_ptr(), w, h, is16, depth, prim is not None, prim or 1, tf if prim else 13, target if prim else 255.0, cfg,
                                                premult, has_alpha, 33, one.data_ptr(), one.numel())
                    torch.cuda.synchronize()
                    assert (ri2.stride, ri2.format, ri2.resolved_config) == (ri.stride, ri.format, ri.resolved_config)
                    assert torch.equal(one, two), (is16, int(cfg), prim, tf, premult, has_alpha)
                    assert np.array_equal(src2.cpu().numpy(), img)                     # the fused form leaves its source alone
                    checked += 1
    assert checked == 2 * 6 * 4 * 3
    dec.close()


@pytest.mark.gpu
def test_decode_pipeline_fused_equals_staged():
    """JxlCoder.decode's post stages fused (default) vs as the reference's two stages: same Bitmap bytes (PQ 16-bit with tone map -> F16 /
    1010102, RGBA with alpha -> 565 / F16, plain sRGB -> 8888)."""
    import jxl_coder_amd as J
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_case
    for name, cfgs in (("v160x120_16bit_pq2100_epf3", (J.PreferredColorConfig.RGBA_F16, J.PreferredColorConfig.RGBA_1010102, J.PreferredColorConfig.RGBA_8888)),
                       ("va300x520_e7", (J.PreferredColorConfig.RGB_565, J.PreferredColorConfig.RGBA_F16, J.PreferredColorConfig.DEFAULT)),
                       ("v256_e7", (J.PreferredColorConfig.RGBA_8888, J.PreferredColorConfig.HARDWARE))):
        data = load_case(name)[0]
        for cfg in cfgs:
            a = J.JxlCoder._decode_pipeline(data, cfg, 33, None, fused_post=True)
            b = J.JxlCoder._decode_pipeline(data, cfg, 33, None, fused_post=False)
            assert (a.stride, a.config, a.use_floats) == (b.stride, b.config, b.use_floats)
            assert np.array_equal(a.rows, b.rows), (name, int(cfg))


def test_reference_cmm_and_system_cmm_agree_on_a8():
    """Stage A8's checker (oracle/icc_oracle.py) runs the Little CMS 2.16 the reference vendors (cpp/icc/*.c compiled where they lie: oracle/_ref/
    liblcms2_ref.so); the product's lattice builder (host_icc_lut.cpp) samples the distribution's liblcms2 (2.12 in this image).  On the embedded
    profile of the ICC fixture the two libraries give the same 8-bit pixels and 16-bit values within 1 code — so the product's lattice needs no
    second source, and the A8 GPU test (tests/test_boundary.py) is a check against the reference's own CMM."""
    import ctypes as C
    import subprocess, sys, textwrap
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import icc_oracle
    if not icc_oracle.available() or not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "liblcms2_ref.so")):
        pytest.skip("needs both Little CMS builds")
    assert icc_oracle.cmm_version() == 2160                          # the reference's: cpp/icc/lcms2.h LCMS_VERSION
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        import icc_oracle
        icc = open(sys.argv[1], "rb").read()
        rng = np.random.default_rng(11)
        px8 = rng.integers(0, 256, (64, 257, 4)).astype(np.uint8)
        px16 = rng.integers(0, 65536, (64, 257, 4)).astype(np.uint16); px16[..., 3] = 65535
        np.save(sys.argv[2], icc_oracle.convert(px8, icc)); np.save(sys.argv[3], icc_oracle.convert(px16, icc))
        print(icc_oracle.cmm_version())
    """) % os.path.join(ROOT, "oracle")
    import tempfile
    import jxl_coder_amd as J
    from jxl_coder_amd import api
    data = open(os.path.join(ROOT, "tests", "golden", "licc96x64_e3.jxl"), "rb").read()
    n = C.c_size_t(); buf = np.zeros(1 << 16, np.uint8)
    assert api.lib().jxlamd_get_icc(data, len(data), buf.ctypes.data, buf.size, C.byref(n)) == 0 and n.value > 0
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "p.icc"), "wb").write(buf[: n.value].tobytes())
        outs = {}
        for tag, env in (("ref", {}), ("sys", {"JXO_SYSTEM_LCMS": "1"})):
            r = subprocess.run([sys.executable, "-c", code, os.path.join(td, "p.icc"), os.path.join(td, tag + "8.npy"), os.path.join(td, tag + "16.npy")],
                               env=dict(os.environ, **env), capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr[-800:]
            outs[tag] = (int(r.stdout.strip()), np.load(os.path.join(td, tag + "8.npy")), np.load(os.path.join(td, tag + "16.npy")))
    assert outs["ref"][0] == 2160 and outs["sys"][0] != 2160
    d8 = np.abs(outs["ref"][1].astype(int) - outs["sys"][1].astype(int)); d16 = np.abs(outs["ref"][2].astype(int) - outs["sys"][2].astype(int))
    assert d8.max() <= 1 and d16.max() <= 2, (d8.max(), d16.max())


@pytest.mark.gpu
def test_writer_post_equals_decode_then_fused_post_stage():
    """jxlamd_decoder_set_writer_post (A10 + A11 WITH the decode): the Bitmap bytes equal jxlamd_decode followed by jxlamd_post_fused, bit for
    bit — for frames whose last filter stage emits them itself (three EPF iterations: the PQ 16-bit fixture, and a 16-bit PQ frame with black bars, whose
    zero-luma pixels stop the reference's tone mapper for the rest of their rows: the second pass of the fused writer), for column-sweep, alpha and Modular
    frames (one pass behind the writer), single decodes and a mixed batch."""
    import torch
    import jxl_coder_amd as J
    from conftest import load_case
    sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
    files = {name: load_case(name)[0] for name in ("v160x120_16bit_pq2100_epf3", "v264x520_e7", "va300x520_e7", "l512_e7", "va530x270_16bit_e7")}
    try:
        import jxl_ref, synth
        if jxl_ref.available():
            img = synth.photo_like(520, 300, seed=5, bits=16)
            img[:40] = 0; img[120:180, 100:400] = 0; img[:, 500:] = 0                      # letterbox bars and a black rectangle: rows with pixels of zero luma
            files["live_pq16_epf3_black_bars"] = jxl_ref.encode(img, effort=7, distance=1.0, epf=3, primaries=9, transfer=16, intensity_target=4000.0)
            # the same picture with the encoder's own filter choice (Gaborish + one EPF iteration: the column sweep's post instantiations, round 5) and an 8-bit HLG one
            files["live_pq16_sweep_black_bars"] = jxl_ref.encode(img, effort=7, distance=1.0, primaries=9, transfer=16, intensity_target=4000.0)
            files["live_hlg8_sweep_black_bars"] = jxl_ref.encode((img >> 8).astype(np.uint8), effort=7, distance=2.0, primaries=9, transfer=18, intensity_target=1000.0)
    except Exception:
        pass
    dec = J.JxlDecoder(0)
    plain = J.JxlDecoder(0)
    combos = ((29, J.PreferredColorConfig.RGBA_F16), (29, J.PreferredColorConfig.RGBA_8888), (33, J.PreferredColorConfig.RGBA_1010102), (29, J.PreferredColorConfig.RGB_565),
              (34, J.PreferredColorConfig.DEFAULT))
    def expected(data, cfg, api):
        raw, info = plain.decode_one_shot(data, allowed_floats=True)
        h, w = raw.shape[:2]; is16 = raw.dtype == np.uint16
        src = torch.from_numpy(raw.view(np.uint8).reshape(-1).copy()).cuda()
        tf = info["transfer_function"]
        matrix = bool(info["prefer_encoding"] and tf in (16, 18, 17, 1, 65535, 13) and info["color_space"] == 0 and api < 34)
        ri = plain.reformat_query(w, h, is16, cfg, info["has_alpha_in_origin"], api)
        dst = torch.zeros(int(ri.bytes), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        plain.post_fused_device(src.data_ptr(), w, h, is16, 16 if is16 else 8, matrix, info["primaries"], tf, info["intensity_target"], cfg,
                                bool(info["alpha_premultiplied"]), bool(info["has_alpha_in_origin"]), api, dst.data_ptr(), dst.numel())
        torch.cuda.synchronize()
        return dst.cpu().numpy(), raw
    zero_rows_seen = False
    for name, data in files.items():
        for api, cfg in combos:
            want, raw = expected(data, cfg, api)
            if "black" in name:
                zero_rows_seen = zero_rows_seen or bool(((raw[..., :3] == 0).all(axis=2)).any())
            dec.set_writer_post(True, cfg, api)
            out = torch.zeros(want.size, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()                                                        # (the decoder writes on its own stream: torch's fill must be done)
            dec.decode_to_device(data, out.data_ptr(), out.numel(), allowed_floats=True)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want), (name, api, int(cfg))
    if "live_pq16_epf3_black_bars" in files:
        assert zero_rows_seen                                                               # the stuck-row behaviour was exercised
    # a batch: frames that emit the Bitmap from their last stage next to frames that take the pass behind the writer
    api, cfg = 29, J.PreferredColorConfig.RGBA_F16
    names = ["v160x120_16bit_pq2100_epf3", "v160x120_16bit_pq2100_epf3", "v160x120_16bit_pq2100_epf3"]
    wants = [expected(files[n], cfg, api)[0] for n in names]
    dec.set_writer_post(True, cfg, api)
    outs = [torch.zeros(wv.size, dtype=torch.uint8, device="cuda") for wv in wants]
    torch.cuda.synchronize()
    dec.decode_batch_to_device([files[n] for n in names], [o.data_ptr() for o in outs], [o.numel() for o in outs])
    torch.cuda.synchronize()
    for o, wv, n in zip(outs, wants, names):
        assert np.array_equal(o.cpu().numpy(), wv), n
    # ... and a batch of column-sweep frames (RGB and RGBA) whose sweep emits the Bitmap format
    api, cfg = 29, J.PreferredColorConfig.RGB_565
    names = ["v264x520_e7", "va300x520_e7", "v264x520_e7"]
    wants = [expected(files[n], cfg, api)[0] for n in names]
    dec.set_writer_post(True, cfg, api)
    outs = [torch.zeros(wv.size, dtype=torch.uint8, device="cuda") for wv in wants]
    torch.cuda.synchronize()
    dec.decode_batch_to_device([files[n] for n in names], [o.data_ptr() for o in outs], [o.numel() for o in outs])
    torch.cuda.synchronize()
    for o, wv, n in zip(outs, wants, names):
        assert np.array_equal(o.cpu().numpy(), wv), n
    # and back: RGBA output again
    dec.set_writer_post(False)
    raw2, _ = dec.decode_one_shot(files["v264x520_e7"])
    assert np.array_equal(raw2, plain.decode_one_shot(files["v264x520_e7"])[0])
    with pytest.raises(ValueError):
        dec.set_writer_post(True, 9, 29)
    dec.close(); plain.close()
