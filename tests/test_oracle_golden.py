"""CPU tests: the oracle (oracle/libjxo.so) against the golden vectors produced by the reference's own libjxl."""
import os

import numpy as np
import pytest

from conftest import (LOSSLESS_CASES, VARDCT_CASES, VARDCT_MAX_ABS, VARDCT_MEAN_ABS, vardct_mean_tol, U16_CASES, U16_PQ_CASES, U16_MAX_ABS,
                      U16_MEAN_ABS, ROOT, JPEG_CASES, load_case)


@pytest.mark.parametrize("name", LOSSLESS_CASES)
def test_oracle_lossless_bit_exact(oracle, name):
    data, exp = load_case(name)
    out, info = oracle.decode(data, 16 if exp.dtype == np.uint16 else 8)      # (16-bit output: the floating-point images)
    assert out.shape == exp.shape
    assert np.array_equal(out, exp)           # integer path: bit-exact


@pytest.mark.parametrize("name", VARDCT_CASES)
def test_oracle_vardct_within_tolerance(oracle, name):
    data, exp = load_case(name)
    out, info = oracle.decode(data, 8)
    d = np.abs(out.astype(int) - exp.astype(int))
    assert d.max() <= VARDCT_MAX_ABS and d.mean() <= vardct_mean_tol(name)
    assert np.array_equal(out[..., 3], exp[..., 3])


@pytest.mark.parametrize("name", JPEG_CASES)
def test_oracle_jpeg_transcodes(oracle, name):
    """YCbCr frames with chroma subsampling and RAW dequant matrices (recompressed JPEGs) against the reference binary's pixels."""
    data, exp = load_case(name)
    out, info = oracle.decode(data, 8)
    d = np.abs(out.astype(int) - exp.astype(int))
    assert d.max() <= VARDCT_MAX_ABS and d.mean() <= 1e-3, (d.max(), d.mean())          # measured: 1 - 27 samples of a file differ, by one


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


def test_epf_offset_is_the_reference_builds_rcpps(oracle):
    """The fixtures with EPF forced to 1 / 2 / 3 iterations sit 0.037 / 0.068 / 0.10 (mean, always the same sign) from the reference.  The reference's
    libjxl is an SSE2-only build: its ApproximateReciprocal in the EPF's normalisation is the CPU's 12-bit rcpps.  With the golden host's instruction in the C
    oracle's normalisation (epf_x86=True: the table oracle/tools/extract_rcp12.py wrote, so any host reproduces it) the three fixtures agree with the goldens
    like any other file (<= 0.012), which pins the cause; without it they show the offset.  rcpps differs between CPU vendors, which is why the product divides
    exactly by default and offers that table as an option (jxlamd_decoder_set_epf_reciprocal)."""
    res = {False: {}, True: {}}
    for x86 in (False, True):
        for name in ["v256_e3_gab0_epf1", "v256_e3_gab0_epf2", "v256_e3_gab0_epf3"]:
            data, exp = load_case(name)
            out, _ = oracle.decode(data, 8, epf_x86=x86)
            d = out.astype(int)[..., :3] - exp.astype(int)[..., :3]
            res[x86][name] = (abs(d).max(), float(abs(d).mean()), float(d.mean()))
        data, exp = load_case("v160x120_16bit_pq2100_epf3")            # config 5's arithmetic: three EPF iterations, PQ 16-bit
        out, _ = oracle.decode(data, 16, epf_x86=x86)
        d = abs(out.astype(int)[..., :3] - exp.astype(int)[..., :3])
        res[x86]["pq16"] = (d.max(), float(d.mean()), int((d > 256).sum()))
    pq_exact, pq_approx = res[False].pop("pq16"), res[True].pop("pq16")
    exact, approx = res[False], res[True]
    # exact division: the offset grows by ~0.03 per iteration and is one-sided (mean |d| == mean d within rounding)
    assert exact["v256_e3_gab0_epf3"][1] > exact["v256_e3_gab0_epf2"][1] > exact["v256_e3_gab0_epf1"][1] > 0.02
    assert all(abs(v[1] - v[2]) < 0.004 for v in exact.values())
    # the golden host's rcpps
    assert all(v[0] <= 1 and v[1] <= 0.012 for v in approx.values()), approx
    # ... and the same instruction accounts for the PQ 16-bit outliers (samples near zero in out-of-gamut pixels, where the inverse opsin matrix
    # cancels terms of order 1 and a 3e-4 relative offset of the filtered XYB becomes thousands of PQ codes): max 11 262 -> 815, 23 -> 5 samples
    # beyond 256 codes, mean 1.11 -> 0.16
    assert pq_exact[0] > 4 * pq_approx[0] and pq_approx[2] < pq_exact[2] and pq_approx[1] < 0.4 * pq_exact[1], (pq_exact, pq_approx)


def test_rcp12_table_is_this_hosts_rcpps_where_the_goldens_were_made(golden_meta):
    """oracle/jxo_rcp12.h and csrc/rcp12_lut.h hold the golden host's rcpps (oracle/tools/extract_rcp12.py).  On a host of the same CPU model the instruction and
    the table agree on every entry; elsewhere (AMD's table differs) this only records that it does not."""
    import platform, re, subprocess, tempfile
    if platform.machine() not in ("x86_64", "AMD64"):
        pytest.skip("rcpps is an x86 instruction")
    a = [int(x) for x in re.findall(r"\b\d+\b", open(os.path.join(ROOT, "oracle", "jxo_rcp12.h")).read().split("{", 1)[1])]
    b = [int(x) for x in re.findall(r"\b\d+\b", open(os.path.join(ROOT, "jxl_coder_amd", "csrc", "rcp12_lut.h")).read().split("{", 1)[1])]
    assert a == b and len(a) == 2048 and max(a) <= 4096 and min(a) >= 0          # identical in product and checker
    src = ("#include <xmmintrin.h>\n#include <stdio.h>\n#include <string.h>\n#include <stdint.h>\n"
           "int main(void){for(uint32_t i=0;i<2048;i++){uint32_t u=0x3f800000u|(i<<12),r;float f;memcpy(&f,&u,4);"
           "f=_mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(f)));memcpy(&r,&f,4);printf(\"%u\\n\",(r-0x3f000000u)>>11);}return 0;}\n")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "r.c"), "w").write(src)
        subprocess.run(["gcc", "-O1", "-o", os.path.join(d, "r"), os.path.join(d, "r.c")], check=True)
        here = [int(x) for x in subprocess.run([os.path.join(d, "r")], check=True, capture_output=True, text=True).stdout.split()]
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    if cpu == golden_meta["_generated_on"]["cpu_model"]:
        assert here == a
    elif here != a:
        import warnings
        warnings.warn("this host's rcpps (%s) differs from the golden host's table in %d of 2048 entries" % (cpu, sum(x != y for x, y in zip(here, a))))
