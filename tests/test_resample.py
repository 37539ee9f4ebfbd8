"""decodeSampled resampler: geometry is integer-exact vs the restatement of weaver/src/scale.rs; the HIP kernels
match the numpy oracle within +-1 LSB for every filter / mode (own stated tolerance: pic-scale is un-vendored, parity unpinned)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_case

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import resample_oracle as R  # noqa: E402


def test_geometry_matches_the_reference_rules():
    import jxl_coder_amd as J
    from jxl_coder_amd import api
    for (w, h, nw, nh, mode) in [(3840, 2160, 1280, 720, 1), (3840, 2160, 1000, 1000, 1), (3840, 2160, 1000, 1000, 2), (3840, 2160, 1000, 1000, 3),
                                 (768, 992, 300, -1, 1), (768, 992, -1, 301, 2), (768, 992, 301, -2, 3), (768, 992, -2, 333, 1), (5, 7, 64, 64, 2),
                                 (2048, 858, 1, 1, 1), (100, 100, 101, 99, 2)]:
        ri = api.RescaleInfo()
        assert api.lib().jxlamd_rescale_query(w, h, nw, nh, mode, api.C.byref(ri)) == 0
        assert (ri.scaled_w, ri.scaled_h, ri.crop_x, ri.crop_y, ri.out_w, ri.out_h) == R.geometry(w, h, nw, nh, mode), (w, h, nw, nh, mode)
    # scale.rs:202-234 by hand: Fit 3840x2160 -> 1000x1000 scales by min(0.26, 0.46) -> 1000x563 (562.5 rounds away from zero), no crop
    assert R.geometry(3840, 2160, 1000, 1000, 1) == (1000, 563, 0, 0, 1000, 563)
    # Fill scales by the max -> 1778x1000, centre-cropped to 1000x1000 at x = 389
    assert R.geometry(3840, 2160, 1000, 1000, 2) == (1778, 1000, 389, 0, 1000, 1000)
    with pytest.raises(ValueError):
        J.JxlCoder.decodeSampled(b"\xff\x0a", 10, 10, J.PreferredColorConfig.RGBA_8888, J.ScaleMode.FIT, jxlResizeFilter=11)     # Invalid Sampler


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_resample_filters_match_the_oracle(sampler):
    import torch
    import jxl_coder_amd as J
    dec = J.JxlDecoder(0)
    rng = np.random.default_rng(sampler)
    for (h, w, nw, nh, mode, is16, depth, premul) in [(97, 131, 50, 40, 3, False, 8, False), (97, 131, 200, 260, 2, False, 8, True),
                                                      (64, 80, 33, -1, 1, True, 16, True), (40, 40, 7, 9, 1, True, 12, False)]:
        maxv = (1 << depth) - 1
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(np.sin(xx / 7.0) * 0.5 + 0.5) * maxv, (np.cos(yy / 5.0) * 0.5 + 0.5) * maxv, rng.uniform(0, maxv, (h, w)),
                        np.clip((xx + yy) / (w + h) * maxv * 1.3, 0, maxv)], -1).astype(np.uint16 if is16 else np.uint8)
        exp = R.rescale(img, depth, nw, nh, mode, sampler, premul)
        src = torch.from_numpy(img.view(np.uint8).reshape(-1).copy()).cuda()
        q = dec.rescale_query(w, h, nw, nh, mode)
        dst = torch.zeros(q.out_w * q.out_h * (8 if is16 else 4), dtype=torch.uint8, device="cuda")
        dec.rescale_device(src.data_ptr(), w, h, is16, depth, nw, nh, mode, sampler, premul, dst.data_ptr(), dst.numel())
        got = dst.cpu().numpy().view(np.uint16 if is16 else np.uint8).reshape(q.out_h, q.out_w, 4)
        assert got.shape == exp.shape
        d = np.abs(got.astype(np.int64) - exp.astype(np.int64))
        tol = 1 if not premul else max(1, maxv // 128)        # un-premultiplying amplifies f32 rounding where alpha is small
        assert d.max() <= tol and (d > 0).mean() < 0.05, (sampler, mode, d.max(), (d > 0).mean())
    dec.close()


@pytest.mark.gpu
def test_decode_sampled_end_to_end():
    """JxlCoder.decodeSampled (kt/JxlCoder.kt:65-105): decode -> RescaleImage -> reformat, the reference's stage order."""
    import jxl_coder_amd as J
    data, exp = load_case("v264x520_e7")
    out = J.JxlCoder.decodeSampled(data, 100, 100, J.PreferredColorConfig.RGBA_8888, J.ScaleMode.FIT, jxlResizeFilter=6)
    assert out.shape == (100, 51, 4)                               # 264x520 fitted into 100x100: scale 100/520 -> 51x100
    ref = R.rescale(exp, 8, 100, 100, 1, 6, False)
    assert np.abs(out.astype(int) - ref.astype(int)).max() <= 2    # +-1 from the decode, +-1 from the f32 filter
    fill = J.JxlCoder.decodeSampled(data, 100, 100, J.PreferredColorConfig.RGBA_8888, J.ScaleMode.FILL)
    assert fill.shape == (100, 100, 4)
