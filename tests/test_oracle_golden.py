"""CPU tests: the oracle (oracle/libjxo.so) against the golden vectors produced by the reference's own libjxl."""
import os

import numpy as np
import pytest

from conftest import (LOSSLESS_CASES, VARDCT_CASES, VARDCT_MAX_ABS, VARDCT_MEAN_ABS, vardct_mean_tol, U16_CASES, U16_PQ_CASES, U16_MAX_ABS,
                      U16_MEAN_ABS, ROOT, load_case)


@pytest.mark.parametrize("name", LOSSLESS_CASES)
def test_oracle_lossless_bit_exact(oracle, name):
    data, exp = load_case(name)
    out, info = oracle.decode(data, 8)
    assert out.shape == exp.shape
    assert np.array_equal(out, exp)           # integer path: bit-exact


@pytest.mark.parametrize("name", VARDCT_CASES)
def test_oracle_vardct_within_tolerance(oracle, name):
    data, exp = load_case(name)
    out, info = oracle.decode(data, 8)
    d = np.abs(out.astype(int) - exp.astype(int))
    assert d.max() <= VARDCT_MAX_ABS and d.mean() <= vardct_mean_tol(name)
    assert np.array_equal(out[..., 3], exp[..., 3])


def test_oracle_basic_info_matches_reference(oracle, golden_meta):
    for name in VARDCT_CASES + LOSSLESS_CASES:
        data, exp = load_case(name)
        info = oracle.basic_info(data)
        ref = golden_meta[name]["info"]
        assert (info["xsize"], info["ysize"]) == (ref["xsize"], ref["ysize"])
        assert info["bits_per_sample"] == ref["bits_per_sample"]
        assert info["num_extra_channels"] == ref["num_extra_channels"]
        assert info["transfer_function"] == ref["transfer_function"] and info["primaries"] == ref["primaries"]


def test_oracle_rejects_truncated(oracle):
    data, _ = load_case("v256_e7")
    with pytest.raises(ValueError):
        oracle.decode(data[: len(data) // 2], 8)
    with pytest.raises(ValueError):
        oracle.decode(b"", 8)
    with pytest.raises(ValueError):
        oracle.decode(b"\x00\x01\x02\x03not a jxl", 8)


def test_oracle_against_live_reference_when_present(oracle):
    """In the build container oracle/_ref (the reference's libjxl) is available: cross-check fresh seeds."""
    jxl_ref = pytest.importorskip("jxl_ref")
    if not jxl_ref.available():
        pytest.skip("oracle/_ref not built")
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth
    for seed, (w, h), kw in [(11, (96, 72), dict(effort=7)), (12, (130, 70), dict(effort=5, distance=2.0)), (13, (64, 48), dict(lossless=True, effort=7))]:
        img = synth.photo_like(w, h, seed=seed)
        data = jxl_ref.encode(img, **kw)
        ref, _, _ = jxl_ref.decode(data)
        out, _ = oracle.decode(data, 8)
        d = np.abs(out.astype(int) - ref.astype(int))
        if kw.get("lossless"):
            assert d.max() == 0
        else:
            assert d.max() <= VARDCT_MAX_ABS and d.mean() <= VARDCT_MEAN_ABS


@pytest.mark.parametrize("name", U16_CASES + U16_PQ_CASES)
def test_oracle_16bit_output(oracle, name):
    data, exp = load_case(name)
    assert exp.dtype == np.uint16
    out, info = oracle.decode(data, 16)
    d = np.abs(out.astype(int) - exp.astype(int))
    assert d.mean() <= U16_MEAN_ABS
    assert np.array_equal(out[..., 3], exp[..., 3])                     # opaque 65535 or the Modular-coded alpha, bit for bit
    if name in U16_CASES:
        assert d.max() <= U16_MAX_ABS
    else:
        assert (d > U16_MAX_ABS).mean() < 2e-3


def test_reference_binary_is_the_pinned_one():
    """oracle/_ref holds the reference's own prebuilt codec (libjxl 0.12.0, cpp/lib/x86_64/libjxl.so) as a binary; every golden vector was
    generated through it.  Pin the file so a different build cannot silently become 'the reference'."""
    import hashlib
    p = os.path.join(ROOT, "oracle", "_ref", "libjxl.so")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    assert hashlib.sha256(open(p, "rb").read()).hexdigest() == "25bd94ff22ae13a62027e266e96fa040c05d116544a75816156d4c540b6c4abe"
