#!/usr/bin/env python3
"""CPU baseline for BATCHES of frames (variant iii): a pool of single-threaded reference decoders, one per
worker process, decoding the same 4K frame for a bounded time.  Uses oracle/_ref (the reference's own libjxl) — a reported
baseline for bench.py's cpu_baseline leg, never part of the product.  Prints one JSON line."""
import json, multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def worker(path, seconds, q):
    import jxl_ref
    data = open(path, "rb").read()
    jxl_ref.decode(data, threads=1)
    n = 0; t0 = time.time()
    while time.time() - t0 < seconds:
        jxl_ref.decode(data, threads=1); n += 1
    q.put((n, time.time() - t0))


def main():
    path = sys.argv[1]
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(path, seconds, q)) for _ in range(procs)]
    t0 = time.time()
    for p in ps: p.start()
    res = [q.get() for _ in ps]
    for p in ps: p.join()
    wall = time.time() - t0
    frames = sum(r[0] for r in res)
    rate = sum(r[0] / r[1] for r in res)            # frames/s summed over workers (each over its own measured window)
    print(json.dumps({"procs": procs, "frames": frames, "wall_s": round(wall, 2), "frames_per_s": round(rate, 2), "MPps": round(rate * 3840 * 2160 / 1e6, 1)}))


if __name__ == "__main__":
    main()
