#!/usr/bin/env python3
"""BASELINE config 4 at full size on ONE GPU: a W x H VarDCT frame (default 32768 x 32768 = 1.07 GP, 128 group rows, 16 384 groups,
4 GiB of RGBA8) from the reference's encoder, decoded as NB bands in sequence (default 8: what 8 GPUs would each take) with the halo
exchange going through device buffers, compared per band with the reference's libjxl run live below its size guard (row sums and
max / mean |diff|), and — the bit-exact check — with the same frame decoded as 2 * NB + 1 bands (different band borders, incl. bands
that split LF groups).  The synthetic image is generated in row tiles by a process pool (tools/synth.py's recipe per tile).
usage: c4_full.py [W H NB [tail]]
tail: only the LAST band of the NB (for 32768 x 32768 and NB = 8: group rows 112 - 128, output byte offset 3.76 GB of the 4 GiB frame) is checked — decoded
next to its upper neighbour (which supplies the halo), once as one band and once as three, compared bit for bit and with the reference's rows: the
driver-run form of the full-size check (tests/test_band_sharded.py), ~2 minutes of which 1.5 are the reference's encoder."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def tile(args):
    w, h, y0, y1, seed = args
    rng = np.random.Generator(np.random.PCG64(0x9E3779B97F4A7C15 ^ seed))
    y, x = np.mgrid[y0:y1, 0:w].astype(np.float32)
    out = np.zeros((y1 - y0, w, 3), np.float32)
    nrng = np.random.Generator(np.random.PCG64((0x9E3779B97F4A7C15 ^ seed) + 977 * y0 + 1))      # per-tile noise stream
    for c in range(3):
        acc = np.full((y1 - y0, w), 0.5, np.float32)
        for _ in range(4):
            fx, fy = rng.uniform(-2.5, 2.5, 2) / max(w, h) * 2 * np.pi
            acc += np.float32(rng.uniform(0.05, 0.18)) * np.sin(np.float32(fx) * x + np.float32(fy) * y + np.float32(rng.uniform(0, 6.28)))
        for _ in range(6):
            fx, fy = rng.uniform(-0.9, 0.9, 2)
            acc += np.float32(rng.uniform(0.01, 0.05)) * np.sin(np.float32(fx) * x + np.float32(fy) * y + np.float32(rng.uniform(0, 6.28)))
        for _ in range(3):
            x0, yy0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            x1, yy1 = x0 + int(rng.integers(8, max(9, w // 3))), yy0 + int(rng.integers(8, max(9, h // 3)))
            a, b = max(yy0, y0), min(yy1, y1)
            v = np.float32(rng.uniform(-0.2, 0.2))
            if a < b:
                acc[a - y0:b - y0, x0:x1] += v
        acc += nrng.uniform(-0.02, 0.02, size=acc.shape).astype(np.float32)
        out[..., c] = acc
    return np.round(np.clip(out, 0, 1) * 255).astype(np.uint8)


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32768, 32768)
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    tail = len(sys.argv) > 4 and sys.argv[4] == "tail"
    free_gb = int(open("/proc/meminfo").read().split("MemAvailable:")[1].split()[0]) / 1e6
    need_gb = w * h * 48 / 1e9            # image + encoder working set (f32 planes) + reference output, with margin
    print("host RAM available %.0f GB, estimated need %.0f GB, cores %d" % (free_gb, need_gb, os.cpu_count()), flush=True)
    if free_gb < need_gb:
        raise SystemExit("not enough host RAM for a %d x %d encode here" % (w, h))
    import multiprocessing as mp
    t = time.time()
    rows = 512
    with mp.get_context("fork").Pool(min(64, os.cpu_count() or 1)) as pool:
        tiles = pool.map(tile, [(w, h, y0, min(h, y0 + rows), 41) for y0 in range(0, h, rows)])
    img = np.concatenate(tiles); del tiles
    print("generated %s in %.1f s" % (img.shape, time.time() - t), flush=True)
    import jxl_ref
    t = time.time(); data = jxl_ref.encode(img, effort=7, distance=1.0); del img
    print("encoded %d bytes (%.3f bpp) in %.1f s" % (len(data), len(data) * 8 / (w * h), time.time() - t), flush=True)
    t = time.time(); ref = jxl_ref.decode(data, threads=0)[0]; t_ref = time.time() - t
    print("reference libjxl decode (runner-suggested threads): %.2f s = %.0f MP/s" % (t_ref, w * h / 1e6 / t_ref), flush=True)
    import torch
    import jxl_coder_amd as J
    from jxl_coder_amd.shard import DeviceBand, band_rows, HALO_LF, HALO_PIXELS
    ygroups = (h + 255) // 256

    def decode_bands(n, rws=None):
        rws = rws if rws is not None else band_rows(ygroups, n)
        decs = [J.JxlDecoder(0) for _ in range(n)]
        stage = np.zeros((n, 3)); bands = {}
        for b in range(n):
            t = time.time(); bands[b] = DeviceBand(decs[b], data, rws[b], w, h, 4, "cuda:0"); stage[b, 0] = time.time() - t
        halo = (decs[0].band_halo_bytes(HALO_LF), decs[0].band_halo_bytes(HALO_PIXELS))
        for b in range(n - 1):
            bands[b + 1].import_(HALO_LF, 0, bands[b].export(HALO_LF, 1)); bands[b].import_(HALO_LF, 1, bands[b + 1].export(HALO_LF, 0))
        for b in range(n):
            t = time.time(); decs[b].band_reconstruct(); stage[b, 1] = time.time() - t
        for b in range(n - 1):
            bands[b + 1].import_(HALO_PIXELS, 0, bands[b].export(HALO_PIXELS, 1)); bands[b].import_(HALO_PIXELS, 1, bands[b + 1].export(HALO_PIXELS, 0))
        for b in range(n):
            t = time.time(); decs[b].band_finish(); stage[b, 2] = time.time() - t
        torch.cuda.synchronize()
        outs = [bands[b].out for b in range(n)]
        del bands, decs
        return rws, outs, stage, halo

    if tail:
        # the last band and the one above it (whose lower edge is the last band's halo; its own upper rows go unchecked: nothing is imported there)
        allr = band_rows(ygroups, nb)
        up, last = allr[-2], allr[-1]
        rws, outs, stage, halo = decode_bands(2, [up, last])
        px = outs[1].cpu().numpy().reshape(-1, w, 4)
        r = ref[last[0] * 256: last[0] * 256 + px.shape[0]]
        mx = 0; sm = 0
        for y0 in range(0, px.shape[0], 256):
            d = np.abs(px[y0:y0 + 256].astype(np.int16) - r[y0:y0 + 256].astype(np.int16)); mx = max(mx, int(d.max())); sm += int(d.sum(dtype=np.int64))
        print("last band group rows %s (%d groups, frame output byte offset %d): begin / reconstruct / finish %.0f / %.0f / %.0f ms | vs reference: max |diff| %d mean %.4f"
              % (last, (last[1] - last[0]) * ((w + 255) // 256), last[0] * 256 * w * 4, *(stage[1] * 1e3), mx, sm / px.size), flush=True)
        third = max(1, (last[1] - last[0]) // 3)
        parts = [(last[0], last[0] + third), (last[0] + third, last[0] + 2 * third), (last[0] + 2 * third, last[1])]
        parts = [p for p in parts if p[1] > p[0]]
        rws2, outs2, _, _ = decode_bands(1 + len(parts), [up] + parts)
        same = bool(torch.equal(torch.cat(outs2[1:]), outs[1]))
        print("last band as 1 band == as %d bands (borders inside LF groups) bit for bit over %d bytes: %s" % (len(parts), outs[1].numel(), same))
        if not same or mx > 1 or sm / px.size > 0.05:
            raise SystemExit(1)
        return
    rws, outs, stage, halo = decode_bands(nb)
    print("%d bands, per band (ms) begin[parse + LF stage] / reconstruct[smoothing + PassGroup + IDCT] / finish[filters + writer]:" % nb)
    worst, tot = 0, 0.0
    for b in range(nb):
        px = outs[b].cpu().numpy().reshape(-1, w, 4)
        r = ref[rws[b][0] * 256: rws[b][0] * 256 + px.shape[0]]
        mx = 0; sm = 0
        for y0 in range(0, px.shape[0], 256):
            d = np.abs(px[y0:y0 + 256].astype(np.int16) - r[y0:y0 + 256].astype(np.int16)); mx = max(mx, int(d.max())); sm += int(d.sum(dtype=np.int64))
        worst = max(worst, mx); tot += sm
        print("  band %d group rows %s (%d groups, output byte offset %d): %.0f / %.0f / %.0f ms | vs reference: max |diff| %d mean %.4f"
              % (b, rws[b], (rws[b][1] - rws[b][0]) * ((w + 255) // 256), rws[b][0] * 256 * w * 4, *(stage[b] * 1e3), mx, sm / px.size), flush=True)
    print("all bands vs reference: max |diff| %d, mean %.4f  (tolerance: max <= 1, mean <= 0.05)" % (worst, tot / (w * h * 4)))
    print("one band per GPU: max-over-bands %.0f ms, sum over bands (this GPU, sequential) %.0f ms = %.0f MP/s; reference CPU %.0f ms; halo per border and direction: LF %d B, pixels %d B"
          % (stage.sum(axis=1).max() * 1e3, stage.sum() * 1e3, w * h / 1e6 / stage.sum(), t_ref * 1e3, halo[0], halo[1]))
    whole_rows = torch.cat(outs); del outs
    n2 = min(ygroups, 2 * nb + 1)
    rws2, outs2, stage2, _ = decode_bands(n2)
    same = bool(torch.equal(torch.cat(outs2), whole_rows))
    print("%d bands == %d bands (other borders, LF groups split) bit for bit over %d bytes: %s" % (nb, n2, whole_rows.numel(), same))
    if not same or worst > 1 or tot / (w * h * 4) > 0.05:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
