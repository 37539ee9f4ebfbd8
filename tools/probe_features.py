"""Which files of the reference's encoder does the product's host parser + device code (CPU harness, tests/emul) decode, and how far from the
reference's decoder?  A sweep over encoder settings: prints one line per case.  Needs /root/reference (oracle/_ref): a development tool.

    python tools/probe_features.py [substring] [--gpu]      (--gpu: the same sweep on the MI355X through the C-ABI; the GPU box carries oracle/_ref)
"""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import jxl_ref        # noqa: E402
import synth          # noqa: E402
import jxl_coder_amd as J        # noqa: E402

# libjxl frame-setting ids (encode.h)
RESAMPLING, EC_RESAMPLING, PHOTON, NOISE, DOTS, PATCHES, EPF, GAB, MODULAR, KEEP_INVIS, GROUP_ORDER = 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13
RESPONSIVE, PROG_AC, QPROG_AC, PROG_DC, PALETTE_COLORS, LOSSY_PALETTE, COLOR_TRANSFORM, MOD_COLORSPACE, MOD_GROUP, MOD_PRED, NB_PREV = 16, 17, 18, 19, 22, 23, 24, 25, 26, 27, 29


def emul_lib():
    so = os.path.join(ROOT, "tests", "emul", "libjxlemul.so")
    lib = C.CDLL(so)
    lib.emul_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.emul_last_error.restype = C.c_char_p
    return lib


def emul_decode(lib, data):
    w, h = J.JxlCoder.getSize(data)
    buf = np.zeros(w * h * 8, np.uint8)
    cw, ch, cb = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = lib.emul_decode(data, len(data), 1, buf.ctypes.data, buf.nbytes, C.byref(cw), C.byref(ch), C.byref(cb))
    if rc:
        return None, lib.emul_last_error().decode()
    n = cw.value * ch.value * 4 * (cb.value // 8)
    return buf[:n].view(np.uint16 if cb.value == 16 else np.uint8).reshape(ch.value, cw.value, 4), ""


def cases():
    ph = synth.photo_like(400, 300, seed=5)
    pha = synth.photo_like(400, 300, seed=6, channels=4)
    big = synth.photo_like(700, 520, seed=7)
    biga = synth.photo_like(700, 520, seed=8, channels=4)
    shot = synth.screenshot(400, 300, seed=2)
    grey = synth.photo_like(300, 200, seed=9, channels=1)
    ph16 = synth.photo_like(300, 200, seed=10, bits=16)
    yield "rgba d1 qprog_ac + responsive", pha, dict(distance=1.0, extra=((QPROG_AC, 1), (RESPONSIVE, 1)))
    yield "rgba d1 prog_ac + responsive", pha, dict(distance=1.0, extra=((PROG_AC, 1), (RESPONSIVE, 1)))
    yield "rgb d1 prog_ac + qprog_ac", ph, dict(distance=1.0, extra=((PROG_AC, 1), (QPROG_AC, 1)))
    yield "rgb d1 prog_ac", ph, dict(distance=1.0, extra=((PROG_AC, 1),))
    yield "rgb d2 prog_dc 2", big, dict(distance=2.0, extra=((PROG_DC, 2),))
    yield "rgb d1 prog_dc 1 + qprog_ac", big, dict(distance=1.0, extra=((PROG_DC, 1), (QPROG_AC, 1)))
    yield "rgba d12 (auto resampling?)", pha, dict(distance=12.0)
    yield "rgba d20", pha, dict(distance=20.0)
    yield "rgb d25", ph, dict(distance=25.0)
    yield "rgba d2 resampling 2", pha, dict(distance=2.0, extra=((RESAMPLING, 2),))
    yield "rgba d2 ec_resampling 2", pha, dict(distance=2.0, extra=((EC_RESAMPLING, 2),))
    yield "rgb d2 noise + resampling 2", ph, dict(distance=2.0, extra=((NOISE, 1), (RESAMPLING, 2)))
    # round 6: the same options on frames that are NOT XYB (lossless), hard-edged content in grey and colour, responsive + resampling, modular lossy grey
    yield "rgb lossless resampling 2", ph, dict(lossless=True, effort=3, extra=((RESAMPLING, 2),))
    yield "rgba lossless resampling 4", pha, dict(lossless=True, effort=5, extra=((RESAMPLING, 4),))
    yield "rgba lossless ec_resampling 2", pha, dict(lossless=True, effort=3, extra=((EC_RESAMPLING, 2),))
    yield "rgba lossless ec_resampling 4 + resampling 2", pha, dict(lossless=True, effort=3, extra=((EC_RESAMPLING, 4), (RESAMPLING, 2)))
    yield "shot lossless e7 resampling 2", shot, dict(lossless=True, effort=7, extra=((RESAMPLING, 2),))
    yield "rgb lossless responsive resampling 2", ph, dict(lossless=True, effort=5, extra=((RESPONSIVE, 1), (RESAMPLING, 2)))
    yield "grey hard-edged d2", synth.hard_edged(400, 300, 9, 1), dict(distance=2.0)
    yield "grey hard-edged lossy modular d1", synth.hard_edged(400, 300, 9, 1), dict(distance=1.0, modular=1)
    yield "rgb hard-edged d3", synth.hard_edged(400, 300, 9, 3), dict(distance=3.0)
    yield "rgba hard-edged d1", synth.hard_edged(400, 300, 9, 4), dict(distance=1.0)
    yield "rgb d5 (auto prog_dc)", big, dict(distance=5.0)
    yield "rgba d5", biga, dict(distance=5.0)
    yield "rgb lossy modular d1", ph, dict(distance=1.0, modular=1)
    yield "rgba lossy modular d1", pha, dict(distance=1.0, modular=1)
    yield "rgb lossy modular d3 responsive 0", ph, dict(distance=3.0, modular=1, extra=((RESPONSIVE, 0),))
    yield "shot lossy palette", shot, dict(lossless=True, extra=((LOSSY_PALETTE, 1),))
    yield "rgb group order centre", big, dict(distance=1.0, extra=((GROUP_ORDER, 1),))
    yield "rgb lossless group order centre", big, dict(lossless=True, effort=3, extra=((GROUP_ORDER, 1),))
    yield "rgb d1 dots", ph, dict(distance=1.0, extra=((DOTS, 1),))
    yield "grey d1", grey, dict(distance=1.0)
    yield "grey lossless", grey, dict(lossless=True)
    yield "rgb16 d1", ph16, dict(distance=1.0)
    yield "rgb16 lossless e7", ph16, dict(lossless=True)
    yield "rgb lossless e9 shot", shot, dict(lossless=True, effort=9)
    yield "rgba lossless responsive", pha, dict(lossless=True, extra=((RESPONSIVE, 1),))
    yield "rgb lossless nb_prev 3", ph, dict(lossless=True, effort=7, extra=((NB_PREV, 3),))
    yield "rgb lossless e9 nb_prev", ph, dict(lossless=True, effort=9)
    for ds in (1, 2, 3, 4):
        yield f"rgb d1 decoding_speed {ds}", ph, dict(distance=1.0, decoding_speed=ds)
    for e in (1, 2, 4, 5, 6, 8, 9):
        yield f"rgb d1 effort {e}", ph, dict(distance=1.0, effort=e)
    for e in (1, 2, 4, 5, 6, 8):
        yield f"rgba lossless effort {e}", pha, dict(lossless=True, effort=e)
    for wh in ((1, 1), (1, 17), (17, 1), (8, 8), (9, 7), (257, 3)):
        im = synth.photo_like(wh[0], wh[1], seed=11)
        yield f"rgb d1 {wh[0]}x{wh[1]}", im, dict(distance=1.0)
        yield f"rgb lossless {wh[0]}x{wh[1]}", im, dict(lossless=True)
    yield "rgb d0.1", ph, dict(distance=0.1)
    yield "rgb d0.05 e3", ph, dict(distance=0.05, effort=3)
    yield "rgb hard d0.1", synth.photo_like(256, 256, seed=3, hard=True), dict(distance=0.1)
    yield "rgb keep invisible lossless", pha, dict(lossless=True, extra=((KEEP_INVIS, 1),))
    yield "rgb lossless colour transform none", ph, dict(lossless=True, extra=((COLOR_TRANSFORM, 1),))
    yield "rgb d1 colour transform none (non-XYB VarDCT)", ph, dict(distance=1.0, extra=((COLOR_TRANSFORM, 1),))
    yield "rgb d1 colour transform ycbcr", ph, dict(distance=1.0, extra=((COLOR_TRANSFORM, 2),))
    yield "rgb lossless modular group 0 (128)", big, dict(lossless=True, effort=3, extra=((MOD_GROUP, 0),))
    yield "rgb lossless modular group 3 (1024)", big, dict(lossless=True, effort=3, extra=((MOD_GROUP, 3),))
    for p in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15):
        yield f"rgb lossless predictor {p}", ph[:120, :160], dict(lossless=True, effort=4, extra=((MOD_PRED, p),))
    # progressive forms (cjxl -p), LF frames, previous-channel properties (cjxl -E), streaming encodes
    yield "rgba d1 qprog_ac (plain alpha)", pha, dict(distance=1.0, extra=((QPROG_AC, 1),))
    yield "big rgba d1 qprog_ac + responsive", biga, dict(distance=1.0, extra=((QPROG_AC, 1), (RESPONSIVE, 1)))
    yield "big rgb d3 prog_dc 1 + prog_ac", big, dict(distance=3.0, extra=((PROG_DC, 1), (PROG_AC, 1)))
    yield "big rgb d1 prog_dc 2 + qprog_ac", big, dict(distance=1.0, extra=((PROG_DC, 2), (QPROG_AC, 1)))
    yield "big rgba d1.5 prog_dc 2", biga, dict(distance=1.5, extra=((PROG_DC, 2),))
    yield "lossless rgb qprog_ac + responsive", ph, dict(lossless=True, extra=((QPROG_AC, 1), (RESPONSIVE, 1)))
    yield "lossy modular rgba prog_ac", pha, dict(distance=2.0, modular=1, extra=((PROG_AC, 1),))
    yield "big rgba lossless e7 nb_prev 5", biga, dict(lossless=True, effort=7, extra=((NB_PREV, 5),))
    yield "rgba lossless e9 nb_prev 11", pha[:200, :300], dict(lossless=True, effort=9, extra=((NB_PREV, 11),))
    yield "rgb lossless responsive nb_prev 3", ph, dict(lossless=True, effort=7, extra=((NB_PREV, 3), (RESPONSIVE, 1)))
    yield "rgba d1 nb_prev 3", pha, dict(distance=1.0, extra=((NB_PREV, 3),))
    yield "rgb lossy modular nb_prev 3", ph, dict(distance=1.0, modular=1, extra=((NB_PREV, 3),))
    yield "shot lossless e7 nb_prev 3", shot, dict(lossless=True, effort=7, extra=((NB_PREV, 3),))
    yield "rgb16 lossless nb_prev 2", ph16, dict(lossless=True, effort=7, extra=((NB_PREV, 2),))
    yield "shot lossy palette, no patches", shot, dict(lossless=True, extra=((LOSSY_PALETTE, 1), (PATCHES, 0)))
    yield "rgb lossy palette (implicit deltas)", ph[:136, :200], dict(lossless=True, extra=((LOSSY_PALETTE, 1), (PATCHES, 0)))
    yield "shot d1 + noise", shot, dict(distance=1.0, extra=((NOISE, 1),))
    yield "rgba d12 + noise", pha, dict(distance=12.0, extra=((NOISE, 1),))
    # channel layouts and sample types: grey + alpha, an extra channel that is not the alpha, premultiplied alpha, floating-point samples
    ga = np.dstack([grey[..., 0], (128 + 100 * np.sin(np.arange(300)[None, :] * 0.05) * np.cos(np.arange(200)[:, None] * 0.04)).astype(np.uint8)])
    depth = (np.arange(400)[None, :] // 2 + np.arange(300)[:, None] // 3).astype(np.uint8)
    for nm, kw in (("lossless e7", dict(lossless=True, effort=7)), ("lossless e1", dict(lossless=True, effort=1)), ("d1", dict(distance=1.0)), ("d12", dict(distance=12.0))):
        yield f"grey+alpha {nm}", ga, kw
    yield "rgb + depth lossless", ph, dict(lossless=True, effort=7, extra_channel=(depth, 1))
    yield "rgb + depth d1", ph, dict(distance=1.0, extra_channel=(depth, 1))
    yield "rgba + spot colour d1", pha, dict(distance=1.0, extra_channel=(depth, 2))
    yield "rgba + selection mask lossless e3", pha, dict(lossless=True, effort=3, extra_channel=(depth, 3))
    yield "rgba + depth d12", pha, dict(distance=12.0, extra_channel=(depth, 1))
    pm = pha.copy(); pm[..., :3] = (pm[..., :3].astype(int) * pm[..., 3:4] // 255).astype(np.uint8)
    for nm, kw in (("lossless", dict(lossless=True, effort=7)), ("d1", dict(distance=1.0)), ("d12", dict(distance=12.0))):
        yield f"premultiplied rgba {nm}", pm, dict(kw, premultiplied=True)
    f32, f32a = ph[:200, :300].astype(np.float32) / 255, pha[:200, :300].astype(np.float32) / 255
    hdr16 = (f32 * 1.7 - 0.2).astype(np.float16)
    yield "float32 rgb d1", f32, dict(distance=1.0)
    yield "float32 rgba d1", f32a, dict(distance=1.0)
    yield "float32 rgb lossless", f32, dict(lossless=True)
    yield "float32 rgba lossless e3", f32a, dict(lossless=True, effort=3)
    yield "float16 rgba d1", f32a.astype(np.float16), dict(distance=1.0)
    yield "float16 rgba lossless e3", f32a.astype(np.float16), dict(lossless=True, effort=3)
    yield "float16 hdr range lossless e7", hdr16, dict(lossless=True, effort=7)
    yield "float16 hdr range d1", hdr16, dict(distance=1.0)
    for b in (2, 3):
        yield f"shot lossless e7 buffering {b}", synth.screenshot(600, 400, seed=2), dict(lossless=True, effort=7, extra=((34, b),))
        yield f"rgb d1 buffering {b}", synth.photo_like(600, 400, seed=3), dict(distance=1.0, extra=((34, b),))


def animations():
    """layered animations (libjxl's encoder API through oracle/ref_shim): every coalesced frame against the reference's getFrame sequence"""
    def rgba(img, a=255):
        return np.dstack([img[..., :3], np.full(img.shape[:2], a, np.uint8)])
    s1, s2, s3 = rgba(synth.screenshot(200, 136, seed=1)), rgba(synth.screenshot(120, 80, seed=2), 200), rgba(synth.screenshot(200, 136, seed=3))
    p1, p2 = rgba(synth.photo_like(200, 136, seed=4)), rgba(synth.photo_like(90, 70, seed=5), 160)
    scenes = {"screenshots, replace": [dict(rgba=s1, duration=3, save=1), dict(rgba=s3, duration=3)],
              "screenshots, blend": [dict(rgba=s1, duration=3, save=1), dict(rgba=s2, x0=30, y0=20, blend=2, source=1, duration=3, save=1), dict(rgba=s2, x0=60, y0=40, blend=2, source=1, duration=3)],
              "photos, blend": [dict(rgba=p1, duration=3, save=1), dict(rgba=p2, x0=33, y0=21, blend=2, source=1, duration=3, save=1), dict(rgba=p2, x0=-10, y0=90, blend=2, source=1, duration=3)]}
    for nm, frames in scenes.items():
        for kw in (dict(lossless=True, effort=7), dict(lossless=False, distance=1.0, effort=7), dict(lossless=False, distance=12.0, effort=7)):
            yield f"anim {nm} {'lossless' if kw['lossless'] else 'd%g' % kw['distance']}", frames, kw


def main():
    args = [a for a in sys.argv[1:] if a != "--gpu"]
    gpu = "--gpu" in sys.argv           # decode on the MI355X through the C-ABI (jxlamd_decode / jxlamd_decode_frame) instead of on the CPU harness
    sub = args[0] if args else ""
    lib = None if gpu else emul_lib()
    dec = J.JxlDecoder(0) if gpu else None

    def emul_decode(lib_, data, frame=-1):       # (shadows the module's function: same result shape for both routes)
        if gpu:
            try:
                return (dec.decode_one_shot(data)[0] if frame < 0 else dec.decode_frame(data, frame)[0]), ""
            except Exception as e:      # noqa: BLE001
                return None, f"{type(e).__name__}: {e}"
        if frame >= 0:
            lib_.emul_set_target_frame(frame)
        try:
            return globals()["emul_decode"](lib_, data)
        finally:
            if frame >= 0:
                lib_.emul_set_target_frame(-1)
    for name, img, kw in cases():
        if sub not in name:
            continue
        try:
            data = jxl_ref.encode(img, **kw)
        except Exception as e:      # noqa: BLE001
            print(f"{name:48s} ENCODE FAILED {e}")
            continue
        ref = jxl_ref.decode(data)
        got, err = emul_decode(lib, data)
        if got is None:
            print(f"{name:48s} {len(data):8d} B  REJECT {err}")
            continue
        r = np.asarray(ref[0])
        if r.shape != got.shape or r.dtype != got.dtype:
            print(f"{name:48s} {len(data):8d} B  SHAPE {r.shape} {r.dtype} vs {got.shape} {got.dtype}")
            continue
        d = np.abs(r.astype(np.int64) - got.astype(np.int64))
        print(f"{name:48s} {len(data):8d} B  ok max {int(d.max())} mean {float(d.mean()):.4f}")
    if lib is not None:
        lib.emul_set_target_frame.argtypes = [C.c_int]
    for name, frames, kw in animations():
        if sub not in name:
            continue
        data = jxl_ref.encode_anim(frames, 200, 136, **kw)
        durs, _ = jxl_ref.anim_info(data)
        worst, res = 0, "ok"
        for i in range(sum(1 for j, f in enumerate(frames) if f.get("duration", 1) > 0 or j == len(frames) - 1)):
            got, err = emul_decode(lib, data, i)
            if got is None:
                res = f"REJECT frame {i}: {err}"
                break
            worst = max(worst, int(np.abs(got.astype(np.int64) - np.asarray(jxl_ref.decode_frame(data, i)).astype(np.int64)).max()))
        print(f"{name:48s} {len(data):8d} B  {res} max {worst}")


if __name__ == "__main__":
    main()
