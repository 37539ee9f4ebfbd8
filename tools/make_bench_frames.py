#!/usr/bin/env python3
"""Generate bench_data/syn4k_q90_seed{1..N-1}.jxl (seed 0 comes from tests/golden/make_golden.py): distinct seeded 3840x2160 photo-like
frames (tools/synth.py) encoded by the reference's own encoder at q90 = distance 1.0, effort 7 (distinct seeds).
Run in the build container (needs oracle/_ref); the files are data fixtures, committed."""
import json, os, sys
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def one(seed):
    import numpy as np
    import jxl_ref, synth
    data = jxl_ref.encode(synth.photo_like(3840, 2160, seed=seed), effort=7, distance=1.0, threads=2)
    open(os.path.join(ROOT, "bench_data", f"syn4k_q90_seed{seed}.jxl"), "wb").write(data)
    out = jxl_ref.decode(data, threads=2)[0]
    return seed, len(data), [int(x) for x in out[::240].astype(np.int64).sum(axis=(1, 2))]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    with ProcessPoolExecutor(4) as ex:
        res = list(ex.map(one, range(1, n)))
    p = os.path.join(ROOT, "tests", "golden", "golden.json")
    meta = json.load(open(p))
    for seed, nbytes, rs in res:
        meta[f"syn4k_q90_seed{seed}"] = dict(bytes=nbytes, shape=[2160, 3840, 4], dtype="uint8", row_sums=rs)
        print(seed, nbytes)
    json.dump(meta, open(p, "w"), indent=1, sort_keys=True)
