#!/usr/bin/env python3
"""Bench inputs: distinct seeded 3840x2160 photo-like frames (tools/synth.py) encoded by the reference's own encoder
(oracle/_ref: the libjxl the reference ships) at q90 = distance 1.0, effort 7.
  make_bench_frames.py [N]                 bench_data/syn4k_q90_seed{1..N-1}.jxl + their row sums in tests/golden/golden.json (committed
                                           fixtures; seed 0 comes from tests/golden/make_golden.py) — run in the build container
  make_bench_frames.py --out DIR --count N [--kind c5]  seeds 0..N-1 into DIR (c5: 4K Rec.2100 PQ 16-bit EPF=3 frames) (not committed: bench.py generates its 256 distinct frames on the box;
                                           seeds that exist under bench_data/ are copied, the rest encoded by a process pool)
Input preparation only: nothing here is part of the decode path."""
import json, os, shutil, sys
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def encode(seed, threads, kind="c3"):
    import jxl_ref, synth
    if kind == "c5":     # BASELINE configs[4]: 4K Rec.2100 PQ 16-bit, distance 1.0, EPF forced to 3 iterations
        return jxl_ref.encode(synth.photo_like(3840, 2160, seed=1000 + seed, bits=16), effort=7, distance=1.0, epf=3, primaries=9, transfer=16,
                              intensity_target=10000.0, threads=threads)
    if kind == "mixed":  # content that is not a q90 photograph, at the reference encoder's defaults (interop/JxlEncoding.cpp:145-160): by seed % 3 a 1920x1080 screenshot at
        # distance 1 effort 7 (patch dictionary: a reference frame + the main frame), a 1920x1080 lossy RGBA photograph (VarDCT colour + squeezed lossy alpha), a
        # 3840x2160 photograph at quality <= 12 (distance 12: coded at half size, 2x upsampled)
        k = seed % 3
        if k == 0:
            return jxl_ref.encode(synth.screenshot(1920, 1080, seed=2000 + seed), effort=7, distance=1.0, threads=threads)
        if k == 1:
            return jxl_ref.encode(synth.photo_like(1920, 1080, seed=2000 + seed, channels=4), effort=7, distance=1.0, threads=threads)
        return jxl_ref.encode(synth.photo_like(3840, 2160, seed=2000 + seed), effort=7, distance=12.0, threads=threads)
    return jxl_ref.encode(synth.photo_like(3840, 2160, seed=seed), effort=7, distance=1.0, threads=threads)


def one(seed):
    import numpy as np
    import jxl_ref
    data = encode(seed, 2)
    open(os.path.join(ROOT, "bench_data", f"syn4k_q90_seed{seed}.jxl"), "wb").write(data)
    out = jxl_ref.decode(data, threads=2)[0]
    return seed, len(data), [int(x) for x in out[::240].astype(np.int64).sum(axis=(1, 2))]


def one_to(args):
    seed, out, kind = args
    dst = os.path.join(out, f"syn4k_q90_seed{seed}.jxl" if kind == "c3" else f"mixed_seed{seed}.jxl" if kind == "mixed" else f"syn4k_pq16_epf3_seed{seed}.jxl")
    if os.path.exists(dst):
        return seed
    src = os.path.join(ROOT, "bench_data", f"syn4k_q90_seed{seed}.jxl")
    if kind == "c3" and os.path.exists(src):
        shutil.copy(src, dst)
        return seed
    data = encode(seed, 1, kind)
    with open(dst + ".tmp", "wb") as f:
        f.write(data)
    os.replace(dst + ".tmp", dst)
    return seed


def big_frame(w, h, dst, seed=41):
    """BASELINE configs[3]: ONE w x h VarDCT q90 frame (default use: 32768 x 32768, 16 384 groups, ~157 MB) written to `dst`.  The image is
    generated in row tiles by a process pool (tools/gpu/c4_full.py: tools/synth.py's recipe per tile) and encoded by the reference's encoder."""
    import time
    import numpy as np
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tools", "gpu"))
    import c4_full
    import jxl_ref
    free_gb = int(open("/proc/meminfo").read().split("MemAvailable:")[1].split()[0]) / 1e6
    if free_gb < w * h * 48 / 1e9:
        raise SystemExit("not enough host RAM for a %d x %d encode here" % (w, h))
    t = time.time()
    with mp.get_context("fork").Pool(min(64, os.cpu_count() or 1)) as pool:
        tiles = pool.map(c4_full.tile, [(w, h, y0, min(h, y0 + 512), seed) for y0 in range(0, h, 512)])
    img = np.concatenate(tiles); del tiles
    data = jxl_ref.encode(img, effort=7, distance=1.0); del img
    with open(dst + ".tmp", "wb") as f:
        f.write(data)
    os.replace(dst + ".tmp", dst)
    return {"file": dst, "bytes": len(data), "seconds": round(time.time() - t, 1)}


if __name__ == "__main__":
    if "--big" in sys.argv:
        i = sys.argv.index("--big")
        print(json.dumps(big_frame(int(sys.argv[i + 1]), int(sys.argv[i + 2]), sys.argv[sys.argv.index("--out") + 1])))
        raise SystemExit(0)
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
        n = int(sys.argv[sys.argv.index("--count") + 1]) if "--count" in sys.argv else 256
        kind = sys.argv[sys.argv.index("--kind") + 1] if "--kind" in sys.argv else "c3"
        os.makedirs(out, exist_ok=True)
        with ProcessPoolExecutor(max(1, min(96, (os.cpu_count() or 2) // 2))) as ex:
            done = list(ex.map(one_to, [(s, out, kind) for s in range(n)], chunksize=1))
        print(json.dumps({"dir": out, "frames": len(done)}))
        raise SystemExit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    with ProcessPoolExecutor(4) as ex:
        res = list(ex.map(one, range(1, n)))
    p = os.path.join(ROOT, "tests", "golden", "golden.json")
    meta = json.load(open(p))
    for seed, nbytes, rs in res:
        meta[f"syn4k_q90_seed{seed}"] = dict(bytes=nbytes, shape=[2160, 3840, 4], dtype="uint8", row_sums=rs)
        print(seed, nbytes)
    json.dump(meta, open(p, "w"), indent=1, sort_keys=True)
