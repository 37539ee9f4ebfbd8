#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run in the build container, where /root/reference exists).

Inputs : seeded synthetic images (tools/synth.py) encoded by the reference's own libjxl encoder through
         oracle/_ref with the reference's call sequence (interop/JxlEncoding.cpp:54-192); "q90" = distance 1.0
         (JXLGetDistance, interop/JxlEncoding.cpp:38-46).
Expected: RGBA output + the DecodeJpegXlOneShot out-params of the reference's libjxl decoder (oracle/_ref),
         stored as compressed .npz next to each .jxl.  Data only; the reference itself does not travel.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import jxl_ref  # noqa: E402
import synth  # noqa: E402

CASES = {
    # name: (w, h, synth kwargs, encode kwargs)
    "v64_e3_gab0_epf0": (64, 64, dict(seed=1), dict(effort=3, gaborish=0, epf=0)),
    "v256_e3_gab0_epf0": (256, 256, dict(seed=1), dict(effort=3, gaborish=0, epf=0)),
    "v256_e3_gab1_epf0": (256, 256, dict(seed=1), dict(effort=3, gaborish=1, epf=0)),
    "v256_e3_gab0_epf1": (256, 256, dict(seed=1), dict(effort=3, gaborish=0, epf=1)),
    "v256_e3_gab0_epf2": (256, 256, dict(seed=1), dict(effort=3, gaborish=0, epf=2)),
    "v256_e3_gab0_epf3": (256, 256, dict(seed=1), dict(effort=3, gaborish=0, epf=3)),
    "v256_e7": (256, 256, dict(seed=1), dict(effort=7)),
    "v264x520_e7": (264, 520, dict(seed=1), dict(effort=7)),
    "v267x131_e7": (267, 131, dict(seed=2), dict(effort=7)),
    "v300x300_e7_d3": (300, 300, dict(seed=3), dict(effort=7, distance=3.0)),
    "v64_hard_e7": (64, 64, dict(seed=4, hard=True), dict(effort=7)),
    "v160x120_16bit_e7": (160, 120, dict(seed=21, bits=16), dict(effort=7)),
    "v160x120_16bit_pq2100_epf3": (160, 120, dict(seed=21, bits=16), dict(effort=7, epf=3, primaries=9, transfer=16, intensity_target=10000.0)),
    "v160x120_16bit_hlg2100": (160, 120, dict(seed=22, bits=16), dict(effort=7, primaries=9, transfer=18, intensity_target=1000.0)),     # HLG: inverse OOTF + OETF in the decoder
    "v160x120_16bit_dci_p3": (160, 120, dict(seed=23, bits=16), dict(effort=7, primaries=11, transfer=17)),                        # DCI gamma 2.6, P3 primaries
    # photographs whose dark pixels clamp to 0 under gamma 2.6 / PQ — code-value differences of hundreds on a handful of samples, nothing in linear light
    "v300x200_16bit_dci_p3_s10": (300, 200, dict(seed=10, bits=16), dict(effort=7, distance=1.0, primaries=11, transfer=17)),
    "v300x200_16bit_dci_p3_s12": (300, 200, dict(seed=12, bits=16), dict(effort=7, distance=1.0, primaries=11, transfer=17)),
    "v300x200_16bit_pq2100_s10": (300, 200, dict(seed=10, bits=16), dict(effort=7, distance=1.0, primaries=9, transfer=16, intensity_target=10000.0)),
    # RGBA through the reference's encoder call sequence: VarDCT colour + Modular-coded (lossless, no squeeze) alpha
    "va300x520_e7": (300, 520, dict(seed=5, alpha=True), dict(effort=7)),
    "va530x270_16bit_e7": (530, 270, dict(seed=8, bits=16, alpha=True), dict(effort=7)),
    "va200x150_e7": (200, 150, dict(seed=9, alpha=True), dict(effort=7)),         # single-section frame: alpha lives in GlobalModular, LfGroup 0 starts where it ends
    "l64_e1": (64, 64, dict(seed=1), dict(lossless=True, effort=1)),
    "l64_e3": (64, 64, dict(seed=1), dict(lossless=True, effort=3)),
    "l64_e7": (64, 64, dict(seed=1), dict(lossless=True, effort=7)),
    "l200x120_e7": (200, 120, dict(seed=5), dict(lossless=True, effort=7)),
    "l512_e7": (512, 512, dict(seed=1), dict(lossless=True, effort=7)),          # BASELINE config 1
    "l300x260_e5": (300, 260, dict(seed=6), dict(lossless=True, effort=5)),        # single 512-px group
    "l700x500_e7": (700, 500, dict(seed=7), dict(lossless=True, effort=7)),        # 3x2 groups with ragged edges
    "l530x300_e1": (530, 300, dict(seed=31), dict(lossless=True, effort=1)),       # libjxl's fast lossless path: prefix codes + LZ77 in every group stream (3x2 groups)
    "l300x280_e2": (300, 280, dict(seed=32), dict(lossless=True, effort=2)),
    "la280x300_e1": (280, 300, dict(seed=33, alpha=True), dict(lossless=True, effort=1)),   # RGBA
    # Modular group sizes other than 256 (JXL_ENC_FRAME_SETTING_MODULAR_GROUP_SIZE = 26; libjxl itself picks 512 for images that fit one such group):
    # 128-px groups (3x2), 512-px groups (2x1, channels wider than the device's LDS rows + weighted predictor), one 1024-px group row
    # Squeeze (JXL_ENC_FRAME_SETTING_RESPONSIVE = 16): default squeeze parameters, local-tree GlobalModular, group streams with channels of mixed shifts
    "lr130x300_e7": (130, 300, dict(seed=74), dict(lossless=True, effort=7, extra=((16, 1),))),            # tall: the sequence starts with a vertical step
    "lrg300x200_e7": (300, 200, dict(seed=76, grey=True), dict(lossless=True, effort=7, extra=((16, 1),))),  # one colour channel
    "lra200x150_e5": (200, 150, dict(seed=72, alpha=True), dict(lossless=True, effort=5, extra=((16, 1),))), # RGBA: 60 stream channels
    "lr2100x40_e3": (2100, 40, dict(seed=92), dict(lossless=True, effort=3, extra=((16, 1),))),              # beyond 2048 px: squeeze residuals in the ModularLfGroup streams
    "va400x300_e7_d2": (400, 300, dict(seed=75, alpha=True), dict(effort=7, distance=2.0)),                 # VarDCT colour + lossy (squeezed, quantised) alpha over 2x2 groups
    "l300x200_g128_e7": (300, 200, dict(seed=61), dict(lossless=True, effort=7, extra=((26, 0),))),
    "la300x200_g128_e5": (300, 200, dict(seed=65, alpha=True), dict(lossless=True, effort=5, extra=((26, 0),))),
    "l516x300_g512_e5": (516, 300, dict(seed=62), dict(lossless=True, effort=5, extra=((26, 2),))),
    "l1030x130_g1024_e3": (1030, 130, dict(seed=63), dict(lossless=True, effort=3, extra=((26, 3),))),
    # ImageMetadata.orientation 2..8: the decoder re-orients (interop/JxlDecoding.cpp never switches that off), the writer transposes / mirrors
    **{f"vo72x40_e3_o{o}": (72, 40, dict(seed=41), dict(effort=3, orientation=o)) for o in range(2, 9)},
    "vo264x300_e7_o6": (264, 300, dict(seed=42), dict(effort=7, orientation=6)),              # several groups, ragged edges, rotated
    "lo40x24_e7_o5": (40, 24, dict(seed=43), dict(lossless=True, effort=7, orientation=5)),   # the Modular writer's re-orientation
    "lo200x120_e7_o8": (200, 120, dict(seed=44), dict(lossless=True, effort=7, orientation=8)),
    # ---- non-photographic content (synth.screenshot / flat / gradient / two_colour), encoder at the reference's defaults (interop/JxlEncoding.cpp:145-160:
    # distance + effort only).  Lossless: multi-channel palettes (e1 / e3), group-level palettes (700x500 without patches), e7 = a kReferenceOnly
    # Modular frame with the glyph patches + a main frame that adds them back.  Lossy: VarDCT main frame + patches from an XYB Modular reference frame.
    "ls400x300_e1": (400, 300, dict(gen="screenshot", seed=1), dict(lossless=True, effort=1)),
    "ls400x300_e3": (400, 300, dict(gen="screenshot", seed=1), dict(lossless=True, effort=3)),
    "lsa400x300_e3": (400, 300, dict(gen="screenshot", seed=3, alpha=True), dict(lossless=True, effort=3)),
    "ls700x500_e7_nopatch": (700, 500, dict(gen="screenshot", seed=2), dict(lossless=True, effort=7, extra=((8, 0),))),
    "lgrad400x300_e7": (400, 300, dict(gen="gradient1d"), dict(lossless=True, effort=7)),
    "lgrad2d200x150_e3": (200, 150, dict(gen="gradient"), dict(lossless=True, effort=3)),
    "lflat400x300_e7": (400, 300, dict(gen="flat"), dict(lossless=True, effort=7)),
    "l2c400x300_e7": (400, 300, dict(gen="two_colour", seed=1), dict(lossless=True, effort=7)),
    "lmany128x96_e3": (128, 96, dict(gen="many_colours", seed=5), dict(lossless=True, effort=3)),          # palette of ~1000 colours (beyond 256 entries)
    # lossy palette (JXL_ENC_FRAME_SETTING_LOSSY_PALETTE = 23): delta palettes — explicit delta entries in front of the colours (the palette channel is
    # nb_colours + nb_deltas wide), indices below zero = the 143 implicit deltas, every delta added to the Average4 prediction of the pixel (raster order)
    "lpl400x300_e7_nopatch": (400, 300, dict(gen="screenshot", seed=2), dict(lossless=True, effort=7, extra=((23, 1), (8, 0)))),   # 7 colours + 12 explicit deltas
    "lpl200x136_e7_photo": (200, 136, dict(seed=5), dict(lossless=True, effort=7, extra=((23, 1), (8, 0)))),                        # 1024 colours, implicit deltas only
    # MA trees that look at previous channels (JXL_ENC_FRAME_SETTING_MODULAR_NB_PREV_CHANNELS = 29; cjxl -E): properties 16.. = |v|, v, |v - gradient|, v - gradient
    # of the nearest earlier channels of the same size and shift
    "lpc200x136_e7_prev3": (200, 136, dict(seed=5), dict(lossless=True, effort=7, extra=((29, 3),))),
    "lpca300x200_e9_prev11": (300, 200, dict(seed=9, alpha=True), dict(lossless=True, effort=9, extra=((29, 11),))),       # RGBA, eleven references asked for, three there
    "lpcr200x136_e7_prev3": (200, 136, dict(seed=5), dict(lossless=True, effort=7, extra=((29, 3), (16, 1)))),             # with squeeze: channels of many sizes, few share one
    # grey + alpha (two-channel PNGs), and images with an extra channel that is not the alpha (depth / spot colour / selection mask: decoded, not part of the RGBA output)
    "lpm400x300_e7_premultiplied": (400, 300, dict(seed=6, alpha=True, premul=True), dict(lossless=True, effort=7, premultiplied=True)),      # premultiplied alpha: delivered as stored
    # floating-point images (cjxl from EXR / PFM): the Modular planes hold the floats' bit patterns; float16 with an HDR range (below 0, above 1), float32 in 0..1
    "lf16_300x200_e7_hdr": (300, 200, dict(seed=5, float=16, frange=(-0.2, 1.5)), dict(lossless=True, effort=7)),
    "lf16a300x200_e3": (300, 200, dict(seed=6, alpha=True, float=16), dict(lossless=True, effort=3)),
    "lf32_200x136_e7": (200, 136, dict(seed=5, float=32), dict(lossless=True, effort=7)),
    # integer samples of more than 16 bits (what cjxl writes from 24-bit PNM sources; libjxl's encoder stops at 24): int32 planes, 64-bit neighbourhoods, and from 23 bits
    # on libjxl's double-precision conversion to [0, 1]; the reference returns RGBA16 for them (interop/JxlDecoding.cpp:92-101)
    "l24_200x136_e7": (200, 136, dict(seed=5, int_bits=24), dict(lossless=True, effort=7, int_bits=24)),
    "l20g_200x136_e3": (200, 136, dict(seed=6, int_bits=20, grey=True), dict(lossless=True, effort=3, int_bits=20)),
    "l24_300x200_e1": (300, 200, dict(seed=7, int_bits=24), dict(lossless=True, effort=1, int_bits=24)),
    "lga300x200_e7": (300, 200, dict(seed=9, grey=True, alpha=True), dict(lossless=True, effort=7)),
    "lga300x200_e1": (300, 200, dict(seed=9, grey=True, alpha=True), dict(lossless=True, effort=1)),
    "lxd400x300_e7_depth": (400, 300, dict(seed=5, extra_type=1), dict(lossless=True, effort=7)),
    "lxs400x300_e3_rgba_selection": (400, 300, dict(seed=6, alpha=True, extra_type=3), dict(lossless=True, effort=3)),
    "lra2100x130_e3": (2100, 130, dict(seed=3, alpha=True), dict(lossless=True, effort=3, extra=((16, 1),))),      # RGBA with squeeze beyond 2048 px: 28 channels in a group's stream, residuals in the ModularLfGroup streams
    "lra400x300_e7": (400, 300, dict(seed=6, alpha=True), dict(lossless=True, effort=7, extra=((16, 1),))),  # squeezed RGBA at effort 7: group streams whose own leaf codes have more than 64 clusters
    "ls400x300_e7": (400, 300, dict(gen="screenshot", seed=1), dict(lossless=True, effort=7)),              # patches
    "lpl400x300_e7": (400, 300, dict(gen="screenshot", seed=2), dict(lossless=True, effort=7, extra=((23, 1),))),                   # lossy palette in the patch frame and in the main frame
    "ls700x500_e5": (700, 500, dict(gen="screenshot", seed=2), dict(lossless=True, effort=5)),              # patches over several groups
    "lsa400x300_e7": (400, 300, dict(gen="screenshot", seed=3, alpha=True), dict(lossless=True, effort=7)), # patches + alpha
    "vs400x300_e7_d1": (400, 300, dict(gen="screenshot", seed=1), dict(effort=7, distance=1.0)),
    "vs400x300_e7_d3": (400, 300, dict(gen="screenshot", seed=1), dict(effort=7, distance=3.0)),
    "vs400x300_e9_d1": (400, 300, dict(gen="screenshot", seed=1), dict(effort=9, distance=1.0)),
    # upsampled frames: the reference's quality <= 12 is distance >= 10 (interop/JxlEncoding.cpp:38-46), where libjxl codes the frame at half size;
    # 4x / 8x through JXL_ENC_FRAME_SETTING_RESAMPLING = 2
    "vu400x300_e7_d10": (400, 300, dict(seed=3), dict(effort=7, distance=10.0)),
    "vu523x267_e7_up4": (523, 267, dict(seed=4), dict(effort=7, distance=2.0, extra=((2, 4),))),
    "vu523x267_e7_up8": (523, 267, dict(seed=4), dict(effort=7, distance=1.0, extra=((2, 8),))),
    "vus400x300_e7_d12": (400, 300, dict(gen="screenshot", seed=1), dict(effort=7, distance=12.0)),       # upsampling + patches
    # ... and RGBA at low quality: the alpha channel is coded at half size too (extra-channel upsampling), or alone (RESAMPLING of the extra channels = 3)
    "vua400x300_e7_d12": (400, 300, dict(seed=3, alpha=True), dict(effort=7, distance=12.0)),
    "vusa400x300_e7_d12": (400, 300, dict(gen="screenshot", seed=3, alpha=True), dict(effort=7, distance=12.0)),
    "va400x300_e7_ecup2": (400, 300, dict(seed=3, alpha=True), dict(effort=7, distance=1.0, extra=((3, 2),))),
    # ... and the same on frames that are NOT XYB (round 6; `cjxl --resampling=2 -d 0`): a Modular frame of the image's own samples coded at half size and enlarged, and a
    # lossless RGBA frame whose alpha alone is coded at half size (its enlarged, fractional values get the 8-bit writer's dither)
    "lu400x300_e3_up2": (400, 300, dict(seed=3), dict(lossless=True, effort=3, extra=((2, 2),))),
    "lua400x300_e3_ecup2": (400, 300, dict(seed=3, alpha=True), dict(lossless=True, effort=3, extra=((3, 2),))),
    "lu523x267_e5_up4": (523, 267, dict(seed=4, alpha=True), dict(lossless=True, effort=5, extra=((2, 4),))),
    # progressive DC (JXL_ENC_FRAME_SETTING_PROGRESSIVE_DC = 19): the LF image travels as an LF frame of its own (a Modular XYB frame at an eighth of the size),
    # the main frame's LfGroup sections carry no LF coefficients; one and two LF groups
    "vlf600x410_e7": (600, 410, dict(seed=11), dict(effort=7, distance=1.0, extra=((19, 1),))),
    "vlf2100x100_e7_d2": (2100, 100, dict(seed=11), dict(effort=7, distance=2.0, extra=((19, 1),))),
    "vlfa520x300_e7_d15": (520, 300, dict(seed=11, alpha=True), dict(effort=7, distance=1.5, extra=((19, 1),))),      # RGBA: the LF frame carries an alpha channel at an eighth too (decoded, not used)
    # noise synthesis (JXL_ENC_FRAME_SETTING_NOISE = 6: the encoder models the image's grain as 8 points of a strength curve, the decoder regenerates it)
    "vn300x200_e7": (300, 200, dict(seed=4, grain=6), dict(effort=7, distance=1.0, extra=((6, 1),))),
    "vn600x410_e7_d15": (600, 410, dict(seed=4, grain=5), dict(effort=7, distance=1.5, extra=((6, 1),))),        # 3 x 2 groups with ragged edges: each group seeds its own generator
    "vna333x277_e7_d15": (333, 277, dict(seed=4, grain=5, alpha=True), dict(effort=7, distance=1.5, extra=((6, 1),))),
    # noise on upsampled frames: the random planes are drawn and added at the full resolution, after the upsampling (tiles of 256 x 256 seeded by their position there)
    "vnu600x410_e7_up2": (600, 410, dict(seed=4, grain=6), dict(effort=7, distance=2.0, extra=((6, 1), (2, 2)))),
    "vnu523x267_e7_d12": (523, 267, dict(seed=4, grain=6), dict(effort=7, distance=12.0, extra=((6, 1),))),
    # custom chromaticities in an enum colour encoding (what encoders write for Adobe RGB / ProPhoto sources): primaries by xy, and a D50 white point that
    # libjxl's output stage adapts with Bradford (white xy, red, green, blue xy)
    "vcadobe200x136_e7": (200, 136, dict(seed=9), dict(effort=7, custom_xy=(0.3127, 0.3290, 0.64, 0.33, 0.21, 0.71, 0.15, 0.06))),
    "vcprophoto200x136_e7": (200, 136, dict(seed=9), dict(effort=7, custom_xy=(0.3457, 0.3585, 0.7347, 0.2653, 0.1596, 0.8404, 0.0366, 0.0001))),
    "vapac520x300_e7": (520, 300, dict(seed=12, alpha=True), dict(effort=7, distance=1.0, extra=((17, 1),))),      # RGBA + progressive AC: the alpha's group streams follow the AC data of the LAST pass
    # what `cjxl -p` writes for RGBA: progressive AC + a squeezed (responsive) alpha — the alpha's channels are spread over the passes by their shift
    # (Passes::GetDownsamplingBracket: every pass has a ModularGroup stream of its own behind the AC data), single group and 3 x 2 groups
    "vpm400x300_e7_premultiplied": (400, 300, dict(seed=6, alpha=True, premul=True), dict(effort=7, distance=1.0, premultiplied=True)),
    "vf16a300x200_e7": (300, 200, dict(seed=6, alpha=True, float=16), dict(effort=7, distance=1.0)),      # float16 image, VarDCT colour + float16 alpha (Modular: bit patterns)
    "vf32a300x200_e7": (300, 200, dict(seed=6, alpha=True, float=32), dict(effort=7, distance=1.0)),
    "vga300x200_e7": (300, 200, dict(seed=9, grey=True, alpha=True), dict(effort=7, distance=1.0)),                  # grey + alpha, VarDCT
    "vga300x200_e7_d12": (300, 200, dict(seed=9, grey=True, alpha=True), dict(effort=7, distance=12.0)),            # ... upsampled
    "vxd400x300_e7_depth": (400, 300, dict(seed=5, extra_type=1), dict(effort=7, distance=1.0)),                     # RGB + a depth channel
    "vxs400x300_e7_rgba_spot": (400, 300, dict(seed=6, alpha=True, extra_type=2), dict(effort=7, distance=1.0)),     # RGBA + a spot-colour channel
    "vapr400x300_e7": (400, 300, dict(seed=6, alpha=True), dict(effort=7, distance=1.0, extra=((17, 1), (16, 1)))),
    "vaqr520x300_e7": (520, 300, dict(seed=13, alpha=True), dict(effort=7, distance=1.0, extra=((18, 1), (16, 1)))),
    # ... and for a photograph with progressive DC: the LF frame is a Modular frame of several passes itself
    "vlfq600x410_e7": (600, 410, dict(seed=14), dict(effort=7, distance=1.0, extra=((19, 1), (18, 1)))),
    # two levels of LF frames (progressive DC = 2): a Modular LF frame of level 2 (1 / 64 of the size) serves a VarDCT LF frame of level 1, which serves the image
    "vlf2_600x410_e7_d2": (600, 410, dict(seed=24), dict(effort=7, distance=2.0, extra=((19, 2),))),
    "vlf2a520x300_e7": (520, 300, dict(seed=25, alpha=True), dict(effort=7, distance=1.5, extra=((19, 2),))),
    # hard-edged saturated content (synth.hard_edged): grey images (the reference returns R = G = B for them), RGB at distances where the encoder switches
    # two and three EPF iterations on, RGBA.  These are the files on which the EPF's reciprocal (the reference build's rcpps) shows in the MAX difference
    "vhg800x600_e7_d1": (800, 600, dict(gen="hard_edged", seed=1, grey=True), dict(effort=7, distance=1.0)),
    "vhg800x600_e7_d2": (800, 600, dict(gen="hard_edged", seed=2, grey=True), dict(effort=7, distance=2.0)),
    "vh1000x700_e7_d2": (1000, 700, dict(gen="hard_edged", seed=3), dict(effort=7, distance=2.0)),
    "vh800x600_e7_d4": (800, 600, dict(gen="hard_edged", seed=4), dict(effort=7, distance=4.0)),
    "vha640x480_e7_d1": (640, 480, dict(gen="hard_edged", seed=5, alpha=True), dict(effort=7, distance=1.0)),
    "vflat400x300_e7": (400, 300, dict(gen="flat"), dict(effort=7)),
    "vgrad200x150_e7": (200, 150, dict(gen="gradient"), dict(effort=7)),
    "v2c400x300_e7": (400, 300, dict(gen="two_colour", seed=1), dict(effort=7)),
}


# JPEG transcodes: what the reference's JxlCoder.construct / JXLJpegInterop (cpp/JXLJpegInterop.cpp:40 -> interop/JxlConstruction.hpp:46-90,
# JxlEncoderAddJPEGFrame) writes and its decode() reads back: a VarDCT frame that is not XYB — YCbCr, the JPEG's quant tables as RAW dequant
# matrices, DCT8 only, chroma coded at the JPEG's subsampling.  The JPEG itself is made here by Pillow (libjpeg) from a seeded synthetic image.
JPEG_CASES = {
    # name: (w, h, synth kwargs, Pillow save kwargs)
    "j444_200x136": (200, 136, dict(seed=3), dict(quality=85, subsampling=0)),
    "j420_200x136": (200, 136, dict(seed=3), dict(quality=85, subsampling=2)),          # ragged in both directions: 12.5 x 8.5 MCUs
    "j422_200x136": (200, 136, dict(seed=3), dict(quality=85, subsampling=1)),
    "j420_600x410": (600, 410, dict(seed=5), dict(quality=75, subsampling=2)),          # 3 x 2 groups, multi-section
    "j420_prog_333x277": (333, 277, dict(seed=6), dict(quality=90, subsampling=2, progressive=True)),
    "jgrey_160x120": (160, 120, dict(seed=7, grey=True), dict(quality=85)),
    "j420s_400x300": (400, 300, dict(gen="screenshot", seed=2), dict(quality=92, subsampling=2)),
}


def add_jpeg_cases(meta, only):
    import io
    from PIL import Image
    for name, (w, h, sk, jk) in JPEG_CASES.items():
        if only and name not in only:
            continue
        img = make_image(w, h, sk)
        buf = io.BytesIO()
        (Image.fromarray(img[..., 0], "L") if img.shape[2] == 1 else Image.fromarray(img[..., :3])).save(buf, "JPEG", **jk)
        data = jxl_ref.encode_jpeg(buf.getvalue())
        out, info, _ = jxl_ref.decode(data, allow16=True)
        open(os.path.join(HERE, name + ".jxl"), "wb").write(data)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rgba=out)
        info = {k: (v if isinstance(v, list) else float(v) if isinstance(v, float) else int(v)) for k, v in info.items()}
        meta[name] = dict(bytes=len(data), jpeg_bytes=len(buf.getvalue()), shape=list(out.shape), dtype=str(out.dtype), info=info, jpeg=jk, synth=sk)
        print(name, len(buf.getvalue()), len(data), out.shape)


# Animations with layers: cropped frames blended over reference slots (what cjxl writes from GIF / APNG sources; the reference's JxlAnimatedEncoder
# itself writes full replacing frames).  Built through libjxl's encoder API (oracle/ref_shim: ref_encode_anim = JxlEncoderSetFrameHeader with layer_info);
# expected = every coalesced frame as the reference's JxlAnimatedDecoder::getFrame returns it (interop/JxlAnimatedDecoder.cpp:28-144) + its frame list.
def anim_scene(kind):
    W, H = 160, 120
    def rgba(img, a):
        return np.dstack([img[..., :3], np.full(img.shape[:2], a, np.uint8) if np.isscalar(a) else a.astype(np.uint8)])
    bg = synth.photo_like(W, H, seed=1)
    yy, xx = np.mgrid[0:40, 0:60]
    al = 255 * np.clip(1 - np.hypot(xx - 30, yy - 20) / 25, 0, 1)
    spr = rgba(synth.photo_like(60, 40, seed=2), al)
    if kind == "blend":            # alpha-blended sprite moving over a kept background; edges not multiples of four, one layer partly outside the canvas
        return W, H, [dict(rgba=rgba(bg, 255), duration=5, save=1), dict(rgba=spr, x0=20, y0=30, blend=2, source=1, duration=5, save=1),
                      dict(rgba=spr, x0=70, y0=50, blend=2, source=1, duration=5, save=1), dict(rgba=spr[:, ::-1].copy(), x0=-10, y0=90, blend=2, source=1, duration=7)]
    if kind == "blend_premul":     # the same layers with premultiplied alpha (declared so in the file): kBlend is then fg + bg (1 - alpha)
        W, H, fr = anim_scene("blend")
        for f in fr:
            px = f["rgba"].copy(); px[..., :3] = (px[..., :3].astype(int) * px[..., 3:4] // 255).astype(np.uint8); f["rgba"] = px
        return W, H, fr
    if kind == "modes":            # kAdd / kMulAdd / kMul layers, zero-duration layers (merged into the next shown frame), two slots, a translucent background
        soft = rgba(synth.photo_like(W, H, seed=3), 200)
        dim = rgba(np.full((30, 50, 3), 40, np.uint8), 128)
        mul = rgba(np.full((50, 70, 3), 200, np.uint8), 255)
        return W, H, [dict(rgba=soft, duration=0, save=2), dict(rgba=dim, x0=12, y0=8, blend=1, source=2, duration=3, save=2),
                      dict(rgba=spr, x0=64, y0=40, blend=3, source=2, duration=0, save=1), dict(rgba=mul, x0=30, y0=60, blend=4, source=1, duration=4, save=1),
                      dict(rgba=spr, x0=100, y0=10, blend=2, source=2, duration=2, save=0), dict(rgba=spr, x0=0, y0=0, blend=0, source=1, duration=6)]
    if kind == "split_modes":      # colour and alpha channel with DIFFERENT blend modes: colour kBlend over alpha kReplace / kAdd / kMul, colour kAdd over alpha kBlend
        soft = rgba(synth.photo_like(W, H, seed=3), 180)
        return W, H, [dict(rgba=soft, duration=2, save=1), dict(rgba=spr, x0=20, y0=30, blend=2, alpha_blend=0, source=1, duration=3, save=1),
                      dict(rgba=spr, x0=70, y0=50, blend=2, alpha_blend=1, source=1, duration=3, save=1),
                      dict(rgba=spr[:, ::-1].copy(), x0=90, y0=10, blend=2, alpha_blend=4, source=1, duration=3, save=1),
                      dict(rgba=spr, x0=8, y0=70, blend=1, alpha_blend=2, source=1, duration=4)]
    raise ValueError(kind)


ANIM_CASES = {
    "an_split_modes_lossless": ("split_modes", dict(lossless=True, effort=3)),
    # name: (scene, encode kwargs)
    "an_blend_lossless": ("blend", dict(lossless=True, effort=3)),
    "an_blend_d1_e7": ("blend", dict(lossless=False, distance=1.0, effort=7)),
    "an_modes_lossless": ("modes", dict(lossless=True, effort=3)),
    "an_modes_d2_e5": ("modes", dict(lossless=False, distance=2.0, effort=5)),
    # the reference's quality <= 12 (distance >= 10): every layer coded at half size, upsampled, then blended at the full resolution
    "an_blend_premul_lossless": ("blend_premul", dict(lossless=True, effort=3, premultiplied=True)),
    "an_blend_premul_d1_e7": ("blend_premul", dict(lossless=False, distance=1.0, effort=7, premultiplied=True)),
    "an_blend_d12_e7": ("blend", dict(lossless=False, distance=12.0, effort=7)),
    "an_modes_d15_e7": ("modes", dict(lossless=False, distance=15.0, effort=7)),
    # layered STILLS (have_animation = 0: every frame is a layer of the one image — what layered exports write): the same layers, composited into a single picture
    "ly_modes_lossless": ("modes", dict(lossless=True, effort=3, still=True)),
    "ly_blend_d1_e7": ("blend", dict(lossless=False, distance=1.0, effort=7, still=True)),
}


def add_anim_cases(meta, only):
    for name, (scene, ek) in ANIM_CASES.items():
        if only and name not in only:
            continue
        W, H, frames = anim_scene(scene)
        ek2 = dict(ek); still = ek2.pop("still", False)
        data = jxl_ref.encode_anim(frames, W, H, tps=(0, 1) if still else (100, 1), loops=3, **ek2)
        durations, loops = jxl_ref.anim_info(data)
        coalesced = 1 if still else sum(1 for i, f in enumerate(frames) if f.get("duration", 1) > 0 or i == len(frames) - 1)
        out = np.stack([jxl_ref.decode_frame(data, i) for i in range(coalesced)])
        last, info, _ = jxl_ref.decode(data)
        assert np.array_equal(last, out[-1])
        open(os.path.join(HERE, name + ".jxl"), "wb").write(data)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rgba=last, frames=out)
        info = {k: (v if isinstance(v, list) else float(v) if isinstance(v, float) else int(v)) for k, v in info.items()}
        meta[name] = dict(bytes=len(data), shape=list(last.shape), dtype=str(last.dtype), info=info, encode=ek, scene=scene, durations_ms=durations, loops=loops,
                          coalesced_frames=coalesced)
        print(name, len(data), out.shape, durations, loops)


# ---- hand-written codestreams (tools/jxl_write.py): what libjxl's encoder never emits and its decoder takes — the DCT128 / DCT256 varblock families
# (AcStrategy 21 .. 26).  One 256 x 256 group (or a 128 x 128 image), seeded coefficients / LF samples / chroma-from-luma factors / quant field / sharpness;
# expected pixels = the reference's decode.
def metadata_case(name, W):
    """What only the image metadata can ask for (no option of libjxl's encoder writes them): custom upsampling weights for factors 2 / 4 / 8 — the defaults scaled and
    shifted by seeded noise, far enough that the default kernels miss the reference's pixels by 18 - 21 codes — and a preview frame in front of the image's frame
    (DecodeJpegXlOneShot never subscribes to it: the decoder walks over it)."""
    def img(size, seed, **kw):
        nb = size // 8
        rng = np.random.default_rng(seed)
        yy, xx = np.mgrid[0:nb, 0:nb]
        lf = np.stack([np.round(30 * np.sin(xx / 5.0 + seed)).astype(np.int64), 5000 + 60 * xx + 45 * yy, np.round(40 * np.cos(yy / 4.0 + seed)).astype(np.int64)])
        blocks = [dict(bx=x, by=y, strategy=0, qf=8, coef={1: {int(k): int(v) for k, v in zip(rng.choice(np.arange(1, 64), 5, replace=False), rng.integers(-20, 21, 5)) if v}})
                  for y in range(nb) for x in range(nb)]
        return W.write_vardct(size, size, blocks, lf, **kw)
    if name == "w_preview":
        return img(64, 1, preview=img(32, 7, as_frame=True))
    fac = {"w_up2_custom": 2, "w_up4_custom": 4, "w_up8_custom": 8}[name]
    txt = open(os.path.join(ROOT, "jxl_coder_amd", "csrc", "upsampling_weights.h")).read()
    body = txt[txt.index("kUpsampling%d[" % fac):]
    body = body[body.index("{") + 1:body.index("}")]
    w = [float(x.rstrip("f")) for x in body.replace("\n", " ").split(",") if x.strip()]
    rng = np.random.default_rng(50 + fac)
    return img(64, fac, upsampling=fac, up_weights={fac: [x * (1 + 0.5 * rng.standard_normal()) + 0.02 * rng.standard_normal() for x in w]})


def dequant_case(name, W):
    """DequantMatrices encodings 1 - 5 (the special 8 x 8 tables from their own parameters: IDENTITY, DCT2X2, DCT4X4 with multipliers, DCT4X8 / DCT8X4 with a
    multiplier, AFV) and 6 (distance bands) for DCT8, on an image whose varblocks use every one of those transforms; with libjxl's library tables instead the
    reference's pixels are up to 46 codes away.  w_dequant_b: other parameters, 17 distance bands for DCT16, library tables for the rest."""
    seed = {"w_dequant_a": 9, "w_dequant_b": 10, "w_dequant_c": 11}[name]
    rng = np.random.default_rng(seed)
    size = 128 if name == "w_dequant_b" else 64
    nb = size // 8
    yy, xx = np.mgrid[0:nb, 0:nb]
    lf = np.stack([np.round(30 * np.sin(xx / 5.0 + seed)).astype(np.int64), 5000 + 60 * xx + 45 * yy, np.round(40 * np.cos(yy / 4.0 + seed)).astype(np.int64)])

    def co(st, n, amp):
        total, covered = W.natural_order_len(st), W.COVERED_X[st] * W.COVERED_Y[st]
        return {int(k): int(v) for k, v in zip(rng.choice(np.arange(covered, total), n, replace=False), rng.integers(-amp, amp + 1, n)) if v}
    f16 = lambda v: float(np.float16(v))
    r = lambda lo, hi: f16(rng.uniform(lo, hi))
    if name != "w_dequant_b":
        strategies = [1, 2, 3, 12, 13, 14, 15, 16, 17, 0]
        blocks = [dict(bx=x, by=y, strategy=strategies[(y * nb + x) % len(strategies)], qf=int(rng.integers(4, 12)), coef={1: co(0, 12, 25), 0: co(0, 4, 5), 2: co(0, 5, 8)})
                  for y in range(nb) for x in range(nb)]
    else:
        blocks = []
        for by in range(0, nb, 2):
            for bx in range(0, nb, 2):
                if (bx // 2 + by // 2) % 3 == 0:
                    blocks.append(dict(bx=bx, by=by, strategy=4, qf=int(rng.integers(4, 12)), coef={1: co(4, 30, 25), 0: co(4, 6, 5), 2: co(4, 8, 8)}))
                else:
                    for (dx, dy) in ((0, 0), (1, 0), (0, 1), (1, 1)):
                        st = [3, 14, 12, 2, 1, 17][(bx + by + dx + 2 * dy) % 6]
                        blocks.append(dict(bx=bx + dx, by=by + dy, strategy=st, qf=int(rng.integers(4, 12)), coef={1: co(0, 12, 25), 0: co(0, 4, 5), 2: co(0, 5, 8)}))
    b48 = [[r(28, 40), r(-1, -0.5), r(-0.9, -0.5), r(-0.7, -0.5)], [r(10, 14), r(-1, -0.6), r(-1, -0.2), r(-0.4, -0.2)], [r(7, 10), r(-1.5, -1), r(-1.5, -1), r(-1.6, -1.2)]]
    b4 = [[r(30, 40), r(-0.1, 0.1), r(-0.1, 0.1), r(-0.1, 0.1)], [r(5, 7), r(-0.1, 0.1), r(-0.1, 0.1), r(-0.1, 0.1)], [r(1.5, 2), r(-0.3, -0.2), r(-0.3, -0.2), r(-0.6, -0.4)]]
    deq = {
        1: (1, [[r(2, 6), r(30, 60), r(30, 60)], [r(0.5, 1.5), r(8, 16), r(8, 16)], [r(0.2, 0.5), r(2, 4), r(2, 4)]]),
        2: (2, [[r(40, 70), r(30, 50), r(15, 25), r(8, 12), r(6, 9), r(4, 6)], [r(10, 18), r(8, 12), r(4, 6), r(2, 3.5), r(2, 2.5), r(1.5, 2)],
                [r(8, 12), r(4, 6), r(1.5, 2.5), r(0.8, 1.2), r(0.4, 0.6), r(0.2, 0.3)]]),
        3: (3, ([[r(0.7, 1.4), r(0.7, 1.4)] for _ in range(3)], [[r(25, 40), r(-0.3, 0.1), r(-0.3, 0.1), r(-0.5, 0)], [r(5, 8), r(-0.3, 0.1), r(-0.2, 0), r(-0.2, 0)],
                                                                   [r(1.5, 2.5), r(-0.4, -0.1), r(-0.4, -0.1), r(-0.6, -0.2)]])),
        9: (4, ([r(0.7, 1.4) for _ in range(3)], [row[:3] for row in b48])),
        10: (5, ([[r(40, 55), r(40, 55), r(3, 5), r(3, 5), r(3, 5), r(5, 8), r(-0.2, 0.1), r(-0.2, 0.1), r(-0.2, 0.1)],
                  [r(14, 18), r(14, 18), r(0.6, 1), r(0.6, 1), r(0.6, 1), r(0.7, 1.1), r(-0.2, 0.1), r(-0.2, 0.1), r(-0.2, 0.1)],
                  [r(5, 7), r(5, 7), r(0.15, 0.25), r(0.15, 0.25), r(0.15, 0.25), r(0.3, 0.4), r(-0.3, -0.2), r(-0.3, -0.2), r(-0.3, -0.2)]], b48, b4)),
    }
    if name == "w_dequant_c":
        # the five special forms belong to the MODE, not to a table — any table of one 8 x 8 block may carry any of them (libjxl checks the table's size only).
        # Here they are rotated: DCT8 <- IDENTITY form, IDENTITY <- DCT2X2 form, DCT2X2 <- DCT4X4 form, DCT4X4 <- DCT4X8 form, DCT4X8 <- AFV form, AFV <- IDENTITY form
        deq = {0: deq[1], 1: deq[2], 2: deq[3], 3: deq[9], 9: deq[10], 10: deq[1]}
    elif name == "w_dequant_a":
        deq[0] = (6, [[r(40, 60), r(-0.2, 0), r(-0.5, -0.3), r(-0.5, -0.3)], [r(7, 10), r(-0.1, 0), r(-0.4, -0.2), r(-0.4, -0.2)], [r(6, 9), r(-2, -1), r(-1, -0.5), r(-0.5, 0)]])
    else:
        del deq[9]
        deq[4] = (6, [[r(120, 150)] + [r(-0.5, -0.1) for _ in range(15)], [r(40, 60)] + [r(-0.4, -0.1) for _ in range(15)], [r(15, 20)] + [r(-0.8, -0.2) for _ in range(15)]])
    return W.write_vardct(size, size, blocks, lf, dequant=deq)


def passes_case(name, W):
    """Frames of 6 and 11 passes (the format's maximum; libjxl's encoder writes at most 4): every pass adds its coefficients << its shift; mixed varblock sizes
    (DCT8, DCT16, DCT32, DCT2X2, DCT4X4), sections apart (TOC of 3 + passes entries)."""
    shifts = {"w_passes6": [3, 2, 1, 0, 0], "w_passes11": [3, 3, 2, 2, 1, 1, 0, 0, 0, 0]}[name]
    size, seed, N = 128, len(shifts), len(shifts) + 1
    nb = size // 8
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:nb, 0:nb]
    lf = np.stack([np.round(30 * np.sin(xx / 5.0 + seed)).astype(np.int64), 5000 + 60 * xx + 45 * yy, np.round(40 * np.cos(yy / 4.0 + seed)).astype(np.int64)])
    sts = [0, 0, 4, 0, 5, 2, 3]
    blocks = []
    occ = np.zeros((nb, nb), bool)
    for y in range(nb):
        for x in range(nb):
            if occ[y, x]:
                continue
            st = sts[int(rng.integers(0, len(sts)))]
            cx, cy = W.COVERED_X[st], W.COVERED_Y[st]
            if x + cx > nb or y + cy > nb or occ[y:y + cy, x:x + cx].any():
                st = 0; cx = cy = 1
            occ[y:y + cy, x:x + cx] = True
            total, cov = W.natural_order_len(st), cx * cy

            def co(n, amp):
                return {int(k): int(v) for k, v in zip(rng.choice(np.arange(cov, total), min(n, total - cov), replace=False), rng.integers(-amp, amp + 1, n)) if v}
            blocks.append(dict(bx=x, by=y, strategy=st, qf=int(rng.integers(4, 12)), coef_passes=[{1: co(6, 6), 0: co(2, 2), 2: co(3, 3)} for _ in range(N)]))
    return W.write_vardct(size, size, blocks, lf, pass_shifts=shifts)


def writer_case(name):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import jxl_write as W
    if name.startswith("w_passes"):
        return passes_case(name, W)
    if name.startswith("w_dequant"):
        return dequant_case(name, W)
    if name.startswith("w_spline"):
        return spline_case(name, W)
    if name.startswith("w_up") or name == "w_preview":
        return metadata_case(name, W)
    seed = {"w_dct256": 1, "w_dct128": 2, "w_dct_mix_a": 3, "w_dct_mix_b": 4, "w_dct128_small": 5, "w_dct256_nofilter": 6}[name]
    rng = np.random.default_rng(seed)
    size = 128 if name == "w_dct128_small" else 256
    nb = size // 8
    yy, xx = np.mgrid[0:nb, 0:nb]
    lf = np.stack([np.round(30 * np.sin(xx / 5.0 + seed)).astype(np.int64), 5000 + 60 * xx + 45 * yy + np.round(200 * np.sin(yy / 3.0 + seed)).astype(np.int64),
                   np.round(40 * np.cos(yy / 4.0 + seed)).astype(np.int64)])

    def coefs(st, n, amp):
        total, covered = W.natural_order_len(st), W.COVERED_X[st] * W.COVERED_Y[st]
        ks = rng.choice(np.arange(covered, total), n, replace=False)
        return {int(k): int(v) for k, v in zip(ks, rng.integers(-amp, amp + 1, n)) if v}

    def blk(bx, by, st, qf=None):
        return dict(bx=bx, by=by, strategy=st, qf=int(rng.integers(3, 12)) if qf is None else qf, coef={1: coefs(st, 40, 30), 0: coefs(st, 10, 6), 2: coefs(st, 12, 10)})
    layouts = {
        "w_dct256": [(0, 0, 24)], "w_dct256_nofilter": [(0, 0, 24)], "w_dct128_small": [(0, 0, 21)],
        "w_dct128": [(0, 0, 21), (16, 0, 21), (0, 16, 21), (16, 16, 21)],
        # 25: 128 wide x 256 tall; 22: 64 wide x 128 tall; 23: 128 wide x 64 tall
        "w_dct_mix_a": [(0, 0, 25), (16, 0, 22), (24, 0, 22), (16, 16, 23), (16, 24, 23)],
        # 26: 256 wide x 128 tall; then a DCT128x128 and four DCT64x64
        "w_dct_mix_b": [(0, 0, 26), (0, 16, 21), (16, 16, 18), (24, 16, 18), (16, 24, 18), (24, 24, 18)],
    }
    blocks = [blk(*b) for b in layouts[name]]
    nt = (nb + 7) // 8
    kw = dict(xfromy=rng.integers(-20, 20, (nt, nt)), bfromy=rng.integers(-10, 30, (nt, nt)), sharpness=rng.integers(0, 8, (nb, nb)))
    if name == "w_dct256_nofilter":
        kw.update(gab=False, epf_iters=0)
    return W.write_vardct(size, size, blocks, lf, **kw)


def spline_case(name, W):
    """Splines (K.4) — libjxl's encoder API cannot place them, jxl-art files do: curves of one to seven control points, colour and thickness varying along the
    arc (higher DCT coefficients), negative colours, a dot (one control point), a curve that leaves the image, thin and thick lines, both signs of the
    quantisation adjustment; over a flat image without loop filters (w_spline_a) and over texture with Gaborish + EPF (w_spline_b / _c)."""
    rng = np.random.default_rng({"w_spline_a": 11, "w_spline_b": 12, "w_spline_c": 13}[name])
    size = 128 if name == "w_spline_a" else 256
    nb = size // 8
    yy, xx = np.mgrid[0:nb, 0:nb]

    def dct(v0, *rest):
        v = [0] * 32; v[0] = v0
        for i, r in enumerate(rest):
            v[1 + i] = r
        return v
    if name == "w_spline_a":
        lf = np.stack([np.zeros((nb, nb), np.int64), 5000 + 0 * xx, np.zeros((nb, nb), np.int64)])
        blocks = [dict(bx=x, by=y, strategy=0, qf=8) for y in range(nb) for x in range(nb)]
        spl = [dict(points=[(20, 20), (60, 40), (100, 30), (110, 100)], color=[dct(8), dct(40, 10), dct(-6)], sigma=dct(12))]
        return W.write_vardct(size, size, blocks, lf, gab=False, epf_iters=0, splines=spl)
    lf = np.stack([np.round(30 * np.sin(xx / 5.0)).astype(np.int64), 5000 + 60 * xx + 45 * yy + np.round(200 * np.sin(yy / 3.0)).astype(np.int64),
                   np.round(40 * np.cos(yy / 4.0)).astype(np.int64)])

    def coefs(st, n, amp):
        total, covered = W.natural_order_len(st), W.COVERED_X[st] * W.COVERED_Y[st]
        ks = rng.choice(np.arange(covered, min(total, covered + 600)), n, replace=False)
        return {int(k): int(v) for k, v in zip(ks, rng.integers(-amp, amp + 1, n)) if v}
    blocks = [dict(bx=x, by=y, strategy=5, qf=int(rng.integers(4, 10)), coef={1: coefs(5, 12, 12), 0: coefs(5, 3, 3), 2: coefs(5, 4, 4)}) for y in range(0, nb, 4) for x in range(0, nb, 4)]
    spl = [dict(points=[(10, 200), (50, 120), (90, 180), (130, 60), (180, 140), (220, 40), (250, 100)], color=[dct(10, -8, 3), dct(60, 25, -12, 6), dct(-10, 5)], sigma=dct(10, 6, -3)),
           dict(points=[(128, 128)], color=[dct(0), dct(90), dct(20)], sigma=dct(25)),                                             # a dot
           dict(points=[(200, 230), (270, 250), (300, 180)], color=[dct(-12), dct(-50, 10), dct(8)], sigma=dct(4, 1)),                 # darker than the image, leaves it on the right
           dict(points=[(30, 30), (31, 90)], color=[dct(4), dct(30), dct(0)], sigma=dct(2))]                                          # thin
    if name == "w_spline_c":
        # (the long curve's thickness crosses zero at 5/6 of its arc — kept in w_spline_b; with this file's quantisation adjustment the reference's own
        # cosine approximation decides two pixels there by 8 levels, so here it stays positive)
        spl[0] = dict(spl[0], sigma=dct(14, 5, -3))
        spl = spl[:1] + [dict(points=[(40, 20), (120, 20), (120, 100), (40, 100), (40, 21)], color=[dct(0, 6), dct(45, -20, 10, -5, 3), dct(12, 4)], sigma=dct(30, -10))]
    return W.write_vardct(size, size, blocks, lf, splines=spl, spline_quant_adjust=-3 if name == "w_spline_b" else 5,
                          xfromy=rng.integers(-20, 20, (4, 4)), bfromy=rng.integers(-10, 30, (4, 4)), sharpness=rng.integers(0, 8, (nb, nb)))


WRITER_CASES = ["w_spline_a", "w_spline_b", "w_spline_c", "w_dct256", "w_dct128", "w_dct_mix_a", "w_dct_mix_b", "w_dct128_small", "w_dct256_nofilter",
                "w_up2_custom", "w_up4_custom", "w_up8_custom", "w_preview", "w_dequant_a", "w_dequant_b", "w_dequant_c", "w_passes6", "w_passes11"]


def add_writer_cases(meta, only):
    for name in WRITER_CASES:
        if only and name not in only:
            continue
        data = writer_case(name)
        out, info, _ = jxl_ref.decode(data, allow16=True)
        open(os.path.join(HERE, name + ".jxl"), "wb").write(data)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rgba=out)
        info = {k: (v if isinstance(v, list) else float(v) if isinstance(v, float) else int(v)) for k, v in info.items()}
        meta[name] = dict(bytes=len(data), shape=list(out.shape), dtype=str(out.dtype), info=info, source="tools/jxl_write.py (hand-written codestream), decoded by the reference")
        print(name, len(data), out.shape, "mean", out[..., :3].mean(), "std", out[..., 1].std())


def make_image(w, h, sk):
    """the synthetic source image of a case (sk: the case's synth kwargs; popped keys are put back by the caller)"""
    sk = dict(sk)
    sk.pop("extra_type", None)          # (an additional extra channel: main() hands its plane to the encoder)
    premul = sk.pop("premul", False)
    fl = sk.pop("float", 0); frange = sk.pop("frange", (0.0, 1.0))
    gen = sk.pop("gen", "photo")
    alpha = sk.pop("alpha", False)
    grey = sk.pop("grey", False)
    grain = sk.pop("grain", 0)
    int_bits = sk.pop("int_bits", 0)
    if int_bits:                # an integer image of 17 .. 24 bits as float32 k / (2^bits - 1): a 16-bit photograph with seeded low bits, every bit of the sample in use
        base = synth.photo_like(w, h, seed=sk.get("seed", 0), bits=16).astype(np.uint64)
        if grey:
            base = base[..., :1]
        iv = (base << (int_bits - 16)) | np.random.default_rng(3000 + sk.get("seed", 0)).integers(0, 1 << (int_bits - 16), base.shape, dtype=np.uint64)
        return np.ascontiguousarray((iv.astype(np.float64) / ((1 << int_bits) - 1)).astype(np.float32))
    if gen == "photo":
        img = synth.photo_like(w, h, **sk)
        if grain:               # sensor-like grain (seeded): what makes the encoder's noise estimation find something to model
            img = np.clip(img.astype(int) + np.random.default_rng(1000 + sk.get("seed", 0)).normal(0, grain, img.shape), 0, 255).astype(np.uint8)
        if grey:
            img = np.ascontiguousarray(img[..., :1])
        if alpha:
            img = with_alpha(img)
        if premul:
            img = img.copy(); img[..., :3] = (img[..., :3].astype(int) * img[..., 3:4] // 255).astype(img.dtype)
        if fl:                  # floating-point samples (float32 / float16) over frange; the alpha stays in 0..1
            f = img.astype(np.float32) / 255.0
            f[..., :3] = f[..., :3] * (frange[1] - frange[0]) + frange[0] if f.shape[2] >= 3 else f[..., :3]
            img = f.astype(np.float32 if fl == 32 else np.float16)
        return img
    if gen == "screenshot":
        return synth.screenshot(w, h, sk.get("seed", 0), channels=4 if alpha else 3)
    if gen == "hard_edged":
        return synth.hard_edged(w, h, sk.get("seed", 0), channels=1 if grey else 4 if alpha else 3)
    if gen == "flat":
        return synth.flat(w, h)
    if gen == "gradient":
        return synth.gradient(w, h)
    if gen == "gradient1d":
        return np.ascontiguousarray(np.repeat(synth.gradient(w, 1), h, axis=0))
    if gen == "two_colour":
        return synth.two_colour(w, h, sk.get("seed", 0))
    if gen == "many_colours":
        return synth.many_colours(w, h, sk.get("seed", 0))
    raise ValueError(gen)


# Tall multi-LF-group frames for the band-sharded decode (BASELINE config 4 in miniature): 17 group rows = 3 LF-group rows.
# Expected pixels are too large to commit: the .jxl + per-row sums of the reference's output (the band test's main assertion is
# bit-identity with the whole-frame decode, which these sums pin to the reference).
ROWSUM_CASES = {
    "vb264x4200_e7_epf3": (264, 4200, dict(seed=11), dict(effort=7, epf=3)),      # Gaborish + 3 EPF iterations: halo H = 7 rows
    "vb520x4400_e7": (520, 4400, dict(seed=12), dict(effort=7)),                  # encoder defaults at d = 1: Gaborish + 1 iteration, H = 3
    # VarDCT + lossy (squeezed) alpha beyond 2048 px: the alpha's shift-3 residuals exceed a group and travel in the ModularLfGroup streams
    "va2300x700_e7_d3": (2300, 700, dict(seed=91, alpha=True), dict(effort=7, distance=3.0)),
}


def add_rowsum_cases(meta, only):
    for name, (w, h, sk, ek) in ROWSUM_CASES.items():
        if only and name not in only:
            continue
        sk2 = dict(sk); al = sk2.pop("alpha", False)
        img = synth.photo_like(w, h, **sk2)
        if al:
            img = with_alpha(img)
        data = jxl_ref.encode(img, **ek)
        out, info, _ = jxl_ref.decode(data, allow16=True)
        open(os.path.join(HERE, name + ".jxl"), "wb").write(data)
        meta[name] = dict(bytes=len(data), shape=list(out.shape), dtype=str(out.dtype), encode=ek, synth=sk,
                          row_sums=[int(x) for x in out.astype(np.int64).sum(axis=(1, 2))], alpha_row_sums=[int(x) for x in out[..., 3].astype(np.int64).sum(axis=1)])
        print(name, len(data), out.shape)


def with_alpha(img):
    """deterministic alpha plane: smooth waves plus fully transparent / fully opaque rectangles"""
    h, w = img.shape[:2]
    yy, xx = np.mgrid[0:h, 0:w]
    mx = 65535 if img.dtype == np.uint16 else 255
    a = (0.5 + 0.47 * np.sin(xx / 37.0) * np.cos(yy / 23.0)) * mx
    a[: h // 8, : w // 5] = 0
    a[h // 3: h // 3 + h // 8, w // 2: w // 2 + w // 5] = mx
    return np.dstack([img[..., :3], a.astype(img.dtype)])


ASSETS = {"asset_first_jxl": "first_jxl.jxl", "asset_wide_gamut": "wide_gamut.jxl", "asset_animated": "animated_jxl.jxl"}     # data files of the reference (app/src/main/assets)


def add_unsupported_exemplar(only):
    """A VALID file the device path refuses (the tests' "unsupported, not corrupt, and never a CPU route" case): a flat 8200 x 8200 RGBA image, lossless with
    squeeze — 20 squeeze steps on four channels = 84 stream channels, four more than the frame tables hold.  81 KB; decoding it is not part of any test."""
    name = "u8200x8200_squeeze_84_channels"
    if only and name not in only:
        return
    img = np.zeros((8200, 8200, 4), np.uint8); img[..., 0] = 37; img[..., 1] = 150; img[..., 2] = 190; img[..., 3] = 255
    data = jxl_ref.encode(img, lossless=True, effort=3, extra=((16, 1),))
    open(os.path.join(HERE, name + ".jxl"), "wb").write(data)
    print(name, len(data))
    # ... and a float32 image whose samples change sign (bit patterns of either sign: W + N - NW leaves 32 bits), lossless at effort 2 — DESIGN.md section 8
    name = "u48x32_float32_mixed_sign"
    img = (synth.photo_like(48, 32, seed=5).astype(np.float32) / 255.0 - 0.5).astype(np.float32)
    data = jxl_ref.encode(img, lossless=True, effort=2)
    open(os.path.join(HERE, name + ".jxl"), "wb").write(data)
    print(name, len(data))


def add_assets(meta, asset_dir="/root/reference/app/src/main/assets"):
    """Real photographs: the reference's own demo assets (inputs) + what the reference's libjxl decodes them to."""
    for name, src in ASSETS.items():
        data = open(os.path.join(asset_dir, src), "rb").read()
        out, info, _ = jxl_ref.decode(data, allow16=True)
        open(os.path.join(HERE, name + ".jxl"), "wb").write(data)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rgba=out)
        info = {k: (v if isinstance(v, list) else float(v) if isinstance(v, float) else int(v)) for k, v in info.items()}
        meta[name] = dict(bytes=len(data), shape=list(out.shape), dtype=str(out.dtype), info=info, source="reference demo asset app/src/main/assets/" + src)


# The other demo assets of the reference that decode on the device: the file (data fixture), per-row sums of the reference's output and
# its 32x32 block means (compressed npz) — the full pixels are up to 92 MB per file.
BIG_ASSETS = {"asset_dark_street": "dark_street.jxl", "asset_large_jxl": "large_jxl.jxl", "asset_pexels": "pexels-thibaut-tattevin-18273081.jxl",
              "asset_second_jxl": "second_jxl.jxl", "asset_summer_nature": "summer_nature.jxl",
              "asset_art": "art.jxl",
              # VarDCT colour + squeeze-coded alpha (the alpha row sums are exact: Modular path)
              "asset_alpha_jxl": "alpha_jxl.jxl", "asset_alpha_png": "alpha_png_freepik.jxl", "asset_hdr_cosmos": "hdr_cosmos.jxl"}            # 73 bytes of MA tree: a 1024x1024 Modular frame in one 1024-px group (lossless: the row sums are exact)


def block_means(out, n=32):
    h, w = out.shape[:2]
    hh, ww = (h + n - 1) // n, (w + n - 1) // n
    pad = np.zeros((hh * n, ww * n, 4), np.float64); cnt = np.zeros((hh * n, ww * n, 1), np.float64)
    pad[:h, :w] = out; cnt[:h, :w] = 1
    sm = pad.reshape(hh, n, ww, n, 4).sum(axis=(1, 3)); c = cnt.reshape(hh, n, ww, n, 1).sum(axis=(1, 3))
    return (sm / c).astype(np.float32)


def add_big_assets(meta, asset_dir="/root/reference/app/src/main/assets"):
    for name, src in BIG_ASSETS.items():
        data = open(os.path.join(asset_dir, src), "rb").read()
        out, info, _ = jxl_ref.decode(data, allow16=True)
        open(os.path.join(HERE, name + ".jxl"), "wb").write(data)
        np.savez_compressed(os.path.join(HERE, name + ".blocks.npz"), means=block_means(out))
        info = {k: (v if isinstance(v, list) else float(v) if isinstance(v, float) else int(v)) for k, v in info.items()}
        meta[name] = dict(bytes=len(data), shape=list(out.shape), dtype=str(out.dtype), info=info, source="reference demo asset app/src/main/assets/" + src,
                          row_sums=[int(x) for x in out.astype(np.int64).sum(axis=(1, 2))], alpha_row_sums=[int(x) for x in out[..., 3].astype(np.int64).sum(axis=1)], fnv1a64="%016x" % jxl_ref.fnv1a64(out.tobytes()))
        print(name, len(data), out.shape, out.dtype)


def main():
    only = set(sys.argv[1:])
    meta = json.load(open(os.path.join(HERE, "golden.json"))) if only else {}
    for name, (w, h, sk, ek) in CASES.items():
        if only and name not in only:
            continue
        img = make_image(w, h, sk)
        more = {}
        if "extra_type" in sk:          # one more extra channel behind the alpha (JxlExtraChannelType: 1 depth, 2 spot colour, 3 selection mask, 4 black): a ramp; the reference's RGBA output ignores it
            more["extra_channel"] = ((np.arange(w)[None, :] // 2 + np.arange(h)[:, None] // 3).astype(np.uint8), sk["extra_type"])
        data = jxl_ref.encode(img, **ek, **more)
        out, info, _ = jxl_ref.decode(data, allow16=True)
        open(os.path.join(HERE, name + ".jxl"), "wb").write(data)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rgba=out)
        info = {k: (v if isinstance(v, list) else float(v) if isinstance(v, float) else int(v)) for k, v in info.items()}
        meta[name] = dict(bytes=len(data), shape=list(out.shape), dtype=str(out.dtype), info=info, encode=ek, synth=sk)
        print(name, len(data), out.shape)
    if not only or "assets" in only:
        add_assets(meta)
    add_rowsum_cases(meta, only)
    add_jpeg_cases(meta, only)
    add_anim_cases(meta, only)
    add_writer_cases(meta, only)
    if only and "u8200x8200_squeeze_84_channels" in only:
        add_unsupported_exemplar(only)          # (9 s and 1.5 GB of encoder memory: on request only)
    if not only or "big_assets" in only:
        add_big_assets(meta)
    # The VarDCT goldens depend on the CPU that made them: the reference's SSE2-only libjxl normalises the EPF sums with the host's 12-bit rcpps, which
    # differs between CPU vendors / generations (tests/test_oracle_golden.py::test_epf_offset_is_the_reference_builds_rcpps) — recorded with the vectors
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:  # noqa: BLE001
        cpu = "unknown"
    meta["_generated_on"] = dict(cpu_model=cpu, note="VarDCT expected pixels come from the reference's prebuilt SSE2-only libjxl 0.12 run on this CPU: its EPF normalisation uses "
                                 "the host's rcpps (12-bit reciprocal estimate), so the vectors are reproducible on this host class only; where oracle/_ref travels the GPU "
                                 "tests also compare against the reference run live")
    if only:
        json.dump(meta, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
        return
    # the bench frame: 4K q90 (BASELINE.json configs[1]); expected pixels are too large to commit -> hash only
    img = synth.photo_like(3840, 2160, seed=0)
    data = jxl_ref.encode(img, effort=7, distance=1.0)
    os.makedirs(os.path.join(ROOT, "bench_data"), exist_ok=True)
    open(os.path.join(ROOT, "bench_data", "syn4k_q90_seed0.jxl"), "wb").write(data)
    out, info, _ = jxl_ref.decode(data)
    import hashlib
    meta["syn4k_q90_seed0"] = dict(bytes=len(data), shape=list(out.shape), dtype=str(out.dtype),
                                   sha256=hashlib.sha256(out.tobytes()).hexdigest(),
                                   row_sums=[int(x) for x in out[::240].astype(np.int64).sum(axis=(1, 2))])
    json.dump(meta, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
