#!/usr/bin/env python3
"""bench.py — decoded MP/s of the JPEG XL decode hot path on MI355X (BASELINE.json metric).

Workload at N=1: BASELINE.json configs[2] — a batch of 256 x 3840x2160 VarDCT q90 (distance 1.0, effort 7) RGB frames -> RGBA8,
256 DISTINCT seeded frames encoded on this box by the reference's own encoder before anything is timed
(tools/make_bench_frames.py; the 8 committed frames of bench_data/ are cycled only if that encoder is unavailable — the line says which).
A "step" = one such batch; every frame is a full decode: host header/TOC/global-table parse, H2D of the frame tables, all HIP kernels
(LF/modular + AC entropy decode, dequant + inverse DCT, Gaborish/EPF, XYB->RGBA).  `value`: the compressed bytes and the RGBA output are
resident in HBM (jxlamd_decode_batch_resident + JXLAMD_OUT_DEVICE); the same steps with the compressed bytes handed over as HOST buffers
(their H2D inside the timed region) are timed right after and reported as `config.h2d_included_MPps`.  Nothing is cached
between frames or steps.  The strictly sequential single-frame latency (configs[1]) is reported in `config`.
N>1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank decodes its own frames — independent
units, no data-path collective — weak scaling; value = frames of all ranks / max-over-ranks time.
"""
import os
import sys


def _argv_int(name, default):
    for i, a in enumerate(sys.argv):
        if a == name and i + 1 < len(sys.argv):
            return int(sys.argv[i + 1])
        if a.startswith(name + "="):
            return int(a.split("=", 1)[1])
    return default


os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(16, min(64, _argv_int("--contexts", 16)))))   # one hardware queue per decoder context: contexts = HIP streams that must overlap (default is 4; contexts that share a queue serialise)
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES = [os.path.join(ROOT, "bench_data", f"syn4k_q90_seed{i}.jxl") for i in (map(int, os.environ["JXLAMD_BENCH_SEEDS"].split(",")) if os.environ.get("JXLAMD_BENCH_SEEDS") else range(8))]   # JXLAMD_BENCH_SEEDS: experiments on a subset
if os.environ.get("JXLAMD_BENCH_FILES"):          # experiments on other content (same frame size for all), e.g. bench_data/real4k_summer_nature.jxl
    FRAMES = [os.path.join(ROOT, f) for f in os.environ["JXLAMD_BENCH_FILES"].split(",")]
FRAMES = [f for f in FRAMES if os.path.exists(f)]
FRAME = FRAMES[0]


def distinct_frames(n, budget_s=420, kind="c3"):
    """n distinct seeded frames generated here (untimed input preparation, cached under /tmp); None if the reference encoder is unavailable."""
    import subprocess
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "jxlamd_bench_frames")
    try:
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_bench_frames.py"), "--out", out, "--count", str(n), "--kind", kind],
                       capture_output=True, text=True, timeout=budget_s, cwd="/tmp")
    except Exception:  # noqa: BLE001 — whatever was finished in time is used
        pass
    files = [os.path.join(out, (f"syn4k_q90_seed{i}.jxl" if kind == "c3" else f"mixed_seed{i}.jxl" if kind == "mixed" else f"syn4k_pq16_epf3_seed{i}.jxl")) for i in range(n)]
    if kind == "mixed":
        return files if all(os.path.exists(f) for f in files) else None
    files = [f for f in files if os.path.exists(f)]
    return files if len(files) > (len(FRAMES) if kind == "c3" else 0) else None


HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (guides/MI355X_MICROARCH.md)


def cpu_baseline(data, budget_s=12.0, c5=False):
    """The reference's own libjxl (oracle/_ref: libjxl 0.12.0 Android-x86_64 build, SSE2-only, JXL_HIGH_PRECISION=0,
    under the loader shim) timed on this host with the reference driver's call sequence and thread choice
    (JxlResizableParallelRunnerSuggestThreads, interop/JxlDecoding.cpp:112-114).  Checker/baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import jxl_ref
        if not jxl_ref.available():
            raise RuntimeError("oracle/_ref not present")
        import numpy as np
        px, info, _ = jxl_ref.decode(data, threads=0)
        mp = info["xsize"] * info["ysize"] / 1e6
        ncpu = os.cpu_count() or 1
        t0 = time.time(); n = 0; best = 1e9
        while time.time() - t0 < budget_s and n < 40:
            t = time.time(); jxl_ref.decode(data, threads=0); best = min(best, time.time() - t); n += 1
        t1 = time.time(); jxl_ref.decode(data, threads=1); one = time.time() - t1
        post_s = 0.0
        if c5:          # configs[4]: + the reference's own post stages (its sources compiled in place: oracle/_ref/libref_post.so) on the decoded RGBA16
            import ctypes as C
            L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_post.so"))
            hgt, wid = px.shape[:2]
            buf = np.ascontiguousarray(px); f16 = np.zeros((hgt, wid, 4), np.uint16)
            xy = (C.c_double * 8)(0.708, 0.292, 0.170, 0.797, 0.131, 0.046, 0.3127, 0.3290)
            tp = time.time()
            L.refpost_color_matrix(C.c_void_p(buf.ctypes.data), wid * 8, wid, hgt, 1, 16, info["primaries"], info["transfer_function"], xy, C.c_float(info["intensity_target"]), None)
            L.refpost_u16_to_f16(C.c_void_p(buf.ctypes.data), wid * 8, C.c_void_p(f16.ctypes.data), wid * 8, wid, hgt, 16)
            post_s = time.time() - tp
            best += post_s; one += post_s
        threaded = mp / best
        # batches of frames (the bench's workload): a pool of single-threaded reference decoders, one process per host core
        # (SURVEY.md §8d iii), in a fresh process tree (no fork of this CUDA-initialised process)
        pool = None
        try:
            if c5:
                raise RuntimeError("c5: the threaded single-frame figure (decode + post stages) is the baseline")
            import subprocess
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_pool_baseline.py"), FRAME, str(ncpu), "4"],
                               capture_output=True, text=True, timeout=180)
            pool = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:  # noqa: BLE001 — the threaded figure alone is still a valid baseline
            pool = None
        use_pool = pool is not None and pool["MPps"] > threaded
        return {"value": round(pool["MPps"] if use_pool else threaded, 2), "unit": "MP/s", "cores": int(pool["procs"]) if use_pool else min(ncpu, 135),
                "kind": "reference",
                "single_frame_threaded_MPps": round(threaded, 2), "pool_of_single_thread_decoders_MPps": (pool or {}).get("MPps"),
                "sample": (f"pool: {pool['frames']} decodes of the same 3840x2160 q90 frame by {pool['procs']} single-threaded reference decoder "
                           f"processes in {pool['wall_s']} s (per-worker 4 s windows); " if pool else "") +
                          f"threaded: {n} decodes of one frame, best-of, runner-suggested threads on {ncpu} host cores; "
                          f"1 thread: {mp / one:.1f} MP/s; libjxl 0.12.0 Android-x86_64 SSE2-only build under bionic shim" +
                          (f"; + the reference's colour matrix / tone map and u16 -> F16 stages (its own sources, single-threaded as in the reference): {post_s * 1e3:.0f} ms per frame" if c5 else "")}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "MP/s", "cores": 0, "kind": "reference", "sample": f"CPU baseline unavailable: {e}"}


def _general_contexts(J, decs):
    """decoder contexts that had to switch to the LF kernel build with the general lock-step loops (0 for libjxl's own streams)"""
    import ctypes as C
    f = J.api.lib().jxlamd_debug_lf_general
    f.argtypes = [C.c_void_p]; f.restype = C.c_int
    return sum(int(f(d._h)) for d in decs)


def _lf_retries(J, decs):
    """(flights decoded twice because the LF table pool of the launch was too small, table pool sizes the contexts ended with)"""
    import ctypes as C
    f = J.api.lib().jxlamd_debug_lf_retries
    f.argtypes = [C.c_void_p, C.POINTER(C.c_uint32 * 3)]
    tot, pools = 0, []
    for d in decs:
        o = (C.c_uint32 * 3)(); f(d._h, C.byref(o)); tot += int(o[0]); pools.append(int(o[2]))
    return tot, pools


def _sparse_state(J, decs):
    """(contexts whose last flight handed its coefficients over as sparse per-varblock lists, flights decoded again with the dense planes)"""
    import ctypes as C
    f = J.api.lib().jxlamd_debug_sparse
    f.argtypes = [C.c_void_p, C.POINTER(C.c_uint32 * 2)]
    on, missed = 0, 0
    for d in decs:
        o = (C.c_uint32 * 2)(); f(d._h, C.byref(o)); on += int(o[0]); missed += int(o[1])
    return on, missed


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def run_c4(args, rank, local, world):
    """BASELINE configs[3]: one huge VarDCT frame sharded by bands of 256-pixel group rows (jxl_coder_amd/shard.py).  A step =
    one decode of the whole frame by all ranks together: every rank parses the (same) bytes, decodes its bands — side by side on their own
    decoder contexts — and trades the LF / pixel halo rows of its borders with the neighbour ranks (RCCL send/recv in one group); borders
    between bands of the same GPU are handed over directly.  Outputs stay in HBM, one tensor per band."""
    import torch
    import torch.distributed as dist
    import jxl_coder_amd as J
    from jxl_coder_amd import shard
    size = int(os.environ.get("JXLAMD_C4_SIZE", "32768"))
    path = os.path.join(os.environ.get("JXLAMD_BENCH_DIR", "/tmp/jxlamd_bench_frames"), f"c4_{size}.jxl")
    if rank == 0 and not os.path.exists(path):
        import subprocess
        os.makedirs(os.path.dirname(path), exist_ok=True)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_bench_frames.py"), "--big", str(size), str(size), "--out", path], capture_output=True, text=True)
        if r.returncode:
            raise SystemExit("--workload c4 needs the reference's encoder (oracle/_ref) to make its frame on this box: " + r.stderr[-400:])
    if world > 1:
        dist.barrier()
    data = open(path, "rb").read()
    w, h = J.JxlCoder.getSize(data)
    ygroups = (h + 255) // 256
    nbands = int(os.environ.get("JXLAMD_C4_BANDS", str(max(8, world))))
    rows = shard.band_rows(ygroups, nbands)
    mine = [b for b in range(nbands) if shard.band_owner(b, nbands, world) == rank]
    outs = [torch.empty((min(rows[b][1] * 256, h) - rows[b][0] * 256) * w * 4, dtype=torch.uint8, device=f"cuda:{local}") for b in mine]

    def step():
        shard.decode_sharded(data, nbands=nbands, rank=rank, world=world, device=local, allowed_floats=False, outs=outs)
    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0)
    if rank != 0:
        return
    ms = elapsed / args.steps * 1e3
    alg = len(data) + w * h * 4
    cpu = {"value": None, "unit": "MP/s", "cores": 0, "kind": "reference", "sample": "skipped (--no-cpu-baseline)"}
    if not args.no_cpu_baseline:
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import jxl_ref        # checker / baseline only
            t = time.perf_counter(); jxl_ref.decode(data, threads=0); dt = time.perf_counter() - t
            cpu = {"value": round(w * h / 1e6 / dt, 1), "unit": "MP/s", "cores": os.cpu_count(), "kind": "reference",
                   "sample": f"one decode of the same {w}x{h} frame by the reference's libjxl 0.12 (oracle/_ref), threads = JxlResizableParallelRunnerSuggestThreads: {dt:.2f} s"}
        except Exception as e:  # noqa: BLE001
            cpu["sample"] = f"CPU baseline unavailable: {e}"
    print(json.dumps({
        "metric": "decoded MP/s (one 32768x32768-class VarDCT q90 frame, band-sharded)", "value": round(w * h / 1e6 / (ms / 1e3), 2), "unit": "MP/s", "n_gpus": world,
        "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"configs[3]: one {w}x{h} VarDCT q90 (distance 1.0, effort 7) frame, {len(data)} bytes, {ygroups * ((w + 255) // 256)} groups, decoded as {nbands} bands "
                               f"of group rows over {world} GPU(s) ({len(mine)} bands on rank 0, concurrent decoder contexts), EPF/LF halo rows between ranks by RCCL send/recv; "
                               "compressed bytes on the host (every rank parses them and uploads its bands' sections), RGBA8 band outputs resident in HBM",
                   "bands": nbands, "frame_bytes": len(data)},
        "roofline": {"bound": "hbm", "achieved": round(alg / (ms / 1e3) / 1e9, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(alg / (ms / 1e3) / 1e9 / 8000.0, 6), "traffic": None,
                     "kernel": "whole sharded decode (all kernels of all bands; per-kernel figures: the c3 line)", "algorithmic_bytes_per_launch": alg},
        "cpu_baseline": cpu}))


def run_mixed(args, rank, local, world):
    """Content that is NOT a q90 photograph, at the reference encoder's own defaults: a step = a batch of --batch (64) frames, by seed % 3
    a 1920x1080 screenshot (distance 1, effort 7: patch dictionary = a reference frame + the main frame), a 1920x1080 lossy RGBA photograph (VarDCT colour +
    squeezed lossy Modular alpha) and a 3840x2160 photograph at distance 12 (coded at half size and upsampled 2x) -> RGBA8.  Frames go through
    jxlamd_decode_batch_resident in flights like the c3 workload; value = OUTPUT megapixels / s."""
    import threading
    import ctypes as C
    import torch
    import torch.distributed as dist
    import jxl_coder_amd as J
    from jxl_coder_amd.shard import max_over_ranks
    argv = " ".join(sys.argv[1:])
    B = args.batch if "--batch" in argv else 64
    P = max(1, min(args.inflight if "--inflight" in argv else 32, B))      # (measured, 6 steps: 8 contexts x 16 frames 489 MP/s, 16 x 16: 615, 16 x 8: 464, 24 x 8: 450, 16 x 32: 936 — these streams are latency-bound, the more of them are resident the better)
    ndist = args.distinct if "--distinct" in argv else 64
    if world > 1:
        if rank == 0:
            distinct_frames(ndist, kind="mixed")
        dist.barrier()
    files = distinct_frames(ndist, kind="mixed")
    if not files:
        raise SystemExit("--workload mixed needs the reference's encoder (oracle/_ref) to make its frames on this box")
    datas = [open(f, "rb").read() for f in files]
    sizes = [J.JxlCoder.getSize(d) for d in datas]
    L = J.api.lib()
    outb = []
    for d in datas:
        n = C.c_size_t()
        if L.jxlamd_output_size(d, len(d), 0, C.byref(n)) != 0:
            raise SystemExit("jxlamd_output_size failed on a generated frame")
        outb.append(int(n.value))
    total = args.steps * B
    NCTX = max(1, min(args.contexts if "--contexts" in argv else 16, (total + P - 1) // P))
    decs = [J.JxlDecoder(local) for _ in range(NCTX)]
    d_ins = [torch.frombuffer(bytearray(d), dtype=torch.uint8).to(f"cuda:{local}") for d in datas]
    d_outs = [[torch.empty(max(outb), dtype=torch.uint8, device=f"cuda:{local}") for _ in range(P)] for _ in range(NCTX)]

    def run_frames(n):
        lock = threading.Lock(); todo = []; done = 0; errors = []; acc = {}
        while done < n:
            p = min(P, n - done); todo.append((done, p)); done += p
        todo.reverse()

        def worker(c):
            try:
                torch.cuda.set_device(local)
                while True:
                    with lock:
                        if not todo:
                            return
                        first, p = todo.pop()
                    ids = [(rank + first + j) % len(datas) for j in range(p)]
                    decs[c].decode_batch_to_device([datas[i] for i in ids], [t.data_ptr() for t in d_outs[c][:p]], [outb[i] for i in ids], [d_ins[i].data_ptr() for i in ids])
                    t = decs[c].last_timing()
                    with lock:
                        for k, v in t.items():
                            acc[k] = acc.get(k, 0.0) + v
                        acc["flights"] = acc.get("flights", 0) + 1
                        acc["mp"] = acc.get("mp", 0.0) + sum(sizes[i][0] * sizes[i][1] for i in ids) / 1e6
                        acc["bytes"] = acc.get("bytes", 0) + sum(len(datas[i]) + outb[i] for i in ids)
            except BaseException as e:  # noqa: BLE001 — re-raised in the main thread
                errors.append(e)
        th = [threading.Thread(target=worker, args=(c,)) for c in range(NCTX)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errors:
            raise errors[0]
        return acc
    run_frames(P * NCTX)
    for _ in range(args.warmup):
        run_frames(B)
    # sequential single-frame latency of one frame of each kind
    lat = {}
    for k, name in enumerate(("screenshot_1080p_patches", "rgba_1080p_lossy_alpha", "photo_4k_distance12_upsampled")):
        i = next((j for j in range(len(datas)) if j % 3 == k), None)
        if i is None:
            continue
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); decs[0].decode_to_device(datas[i], d_outs[0][0].data_ptr(), outb[i], data_dev_ptr=d_ins[i].data_ptr()); best = min(best, time.perf_counter() - t)
        lat[name] = round(best * 1e3, 3)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = run_frames(total)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    if rank != 0:
        return
    mp_total = acc["mp"] * world
    value = mp_total / elapsed
    gbps = acc["bytes"] * world / elapsed / 1e9
    flights = max(int(acc.get("flights", 1)), 1)
    cpu = {"value": None, "unit": "MP/s", "cores": 0, "kind": "reference", "sample": "skipped"}
    if not args.no_cpu_baseline and world == 1:
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import jxl_ref        # checker / baseline only
            tt, mm, n = 0.0, 0.0, 0
            t_end = time.time() + 12.0
            while time.time() < t_end and n < 3 * len(datas):
                i = n % len(datas)
                t = time.perf_counter(); jxl_ref.decode(datas[i], threads=0); tt += time.perf_counter() - t; mm += sizes[i][0] * sizes[i][1] / 1e6; n += 1
            cpu = {"value": round(mm / tt, 1), "unit": "MP/s", "cores": min(os.cpu_count() or 1, 135), "kind": "reference",
                   "sample": f"{n} decodes walking the same mixed batch, one frame at a time, by the reference's libjxl 0.12 (oracle/_ref), threads = JxlResizableParallelRunnerSuggestThreads: {tt:.1f} s"}
        except Exception as e:  # noqa: BLE001
            cpu["sample"] = f"CPU baseline unavailable: {e}"
    print(json.dumps({
        "metric": "decoded MP/s (mixed content: screenshots with patches, lossy RGBA, distance-12 upsampled photographs -> RGBA8)", "value": round(value, 2), "unit": "MP/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "value_inputs": "compressed bytes resident in HBM",
        "config": {"workload": f"mixed (not a BASELINE config; VERDICT r4 2c): one step = {B} frames, seed % 3 -> 1920x1080 screenshot d1 e7 (patches: reference frame + main frame) / 1920x1080 lossy RGBA photograph "
                               f"d1 e7 (VarDCT + squeezed Modular alpha) / 3840x2160 photograph d12 e7 (2x upsampled); {len(datas)} distinct frames made on this box by the reference's encoder; "
                               "every frame a complete decode; compressed bytes and RGBA8 outputs resident in HBM",
                   "frames_per_step_per_gpu": B, "frames_in_flight": P, "decoder_contexts": NCTX, "distinct_frames": len(datas), "output_MP_per_step": round(acc["mp"] / args.steps, 2),
                   "frame_bytes_mean": int(sum(map(len, datas)) / len(datas)), "single_frame_latency_ms": lat,
                   "stage_ms_per_flight": {k: round(v / flights, 3) for k, v in acc.items() if k.endswith("_ms")}},
        "roofline": {"bound": "hbm", "achieved": round(gbps, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBS, 6), "traffic": None,
                     "kernel": "whole decode (all kernels of all frames; per-kernel figures: the c3 line and profiles/)", "algorithmic_bytes_per_launch": int(acc["bytes"] / flights)},
        "cpu_baseline": cpu}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps; one step = one batch of --batch 4K frames (round 5: 20 by default — 8 steps are two flights per context, a run that is mostly start and tail: 11 451 - 12 551 MP/s where 20 steps give 14 083 - 14 177)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="frames per step (BASELINE configs[2]: 256 x 3840x2160 q90)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--contexts", type=int, default=16, help="decoder contexts taking flights alternately (overlaps one flight's LF stage with another's later stages)")
    ap.add_argument("--share", type=int, default=1, help="contexts per set of HF-phase pools (jxlamd_decoder_share_pools): with 2, --contexts 32 keeps 16 pool sets busy — a context's LF stage "
                    "overlaps its partner's PassGroup / reconstruction / filter stages")
    ap.add_argument("--inflight", type=int, default=64, help="frames decoded per batched flight (1 = strictly sequential)")
    ap.add_argument("--workload", choices=["c3", "c5", "c4", "mixed"], default="c3", help="c3 (default): BASELINE configs[2], 256 x 4K VarDCT q90 -> RGBA8.  c5: configs[4], a batch of 64 x "
                    "4K Rec.2100 PQ 16-bit EPF=3 frames -> RGBA16 -> colour matrix + Rec.2408 tone map -> RGBA_F16 (post stages fused, jxlamd_post_fused).  "
                    "c4: configs[3], ONE 32768x32768 VarDCT q90 frame (JXLAMD_C4_SIZE) as bands of group rows over the ranks (8 bands on one GPU), "
                    "halo rows by RCCL send/recv; a step = one decode of the frame (strong scaling).  mixed: 64 frames per step of content that is not a q90 photograph "
                    "(screenshots with patches, lossy RGBA, distance-12 upsampled photographs), see run_mixed")
    ap.add_argument("--c5-post", choices=["writer", "pass"], default="writer", help="--workload c5: A10 + A11 inside the decoder's writer (jxlamd_decoder_set_writer_post: the last "
                    "EPF stage emits RGBA_F16, the RGBA16 image is never stored) or as one fused pass over the stored RGBA16 (jxlamd_post_fused, rounds 2-3)")
    ap.add_argument("--distinct", type=int, default=256, help="distinct seeded frames to generate for the batch (SURVEY.md §8d: 256; 0 = cycle the 8 committed ones)")
    args = ap.parse_args()
    c5 = args.workload == "c5"
    if c5:                      # 16-bit outputs + F16 destinations: smaller flights than the RGBA8 batch (HBM), 64 frames per step
        argv = " ".join(sys.argv[1:])
        if "--batch" not in argv: args.batch = 64
        if "--contexts" not in argv: args.contexts = 16         # (8 contexts: 2 680 MP/s, 12 or 16: 3 200; flights of 16 or 24: 1 670 - 1 790 while the flat PassGroup kernel started at 4 096 groups, 2 770 since it starts at 1 024)
        if "--inflight" not in argv: args.inflight = 32
        if "--distinct" not in argv: args.distinct = 64
        if "--steps" not in argv: args.steps = 32              # four flights of 32 frames per context (round 6: 3 211 - 3 355 MP/s over four runs, +- 2.2 %; 8 steps = one flight each gave 2 557 - 3 103 in round 5)
        if "--warmup" not in argv: args.warmup = 8

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver does for N > 1
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the decode path has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    import jxl_coder_amd as J
    from jxl_coder_amd.shard import max_over_ranks
    if args.workload == "c4":
        return run_c4(args, rank, local, world)
    if args.workload == "mixed":
        if "--steps" not in " ".join(sys.argv[1:]): args.steps = 48      # six flights of 32 for each of the 16 contexts: the rate a run settles at (round 6, one box: 8 steps 2 508 - one flight per context, all in step -, 16 steps 1 652 - 1 933, 32 steps 1 815 - 2 193, 48 steps 2 098 - 2 253 MP/s)
        if "--warmup" not in " ".join(sys.argv[1:]): args.warmup = 8
        return run_mixed(args, rank, local, world)

    # the batch: distinct seeded frames (tools/make_bench_frames.py), cycled to --batch frames; rank r starts at seed r
    frames = FRAMES
    if c5 or (not os.environ.get("JXLAMD_BENCH_FILES") and not os.environ.get("JXLAMD_BENCH_SEEDS") and args.distinct > len(FRAMES)):
        kind = "c5" if c5 else "c3"
        if world > 1:                       # one rank prepares the inputs, the others wait for the files
            if rank == 0:
                distinct_frames(max(args.distinct, 1), kind=kind)
            dist.barrier()
        frames = distinct_frames(max(args.distinct, 1), kind=kind) or (None if c5 else FRAMES)
        if frames is None:
            raise SystemExit("--workload c5 needs the reference's encoder (oracle/_ref) to make its 4K PQ 16-bit frames on this box")
    datas = [open(f, "rb").read() for f in frames]
    data = datas[0]
    w, h = J.JxlCoder.getSize(data)
    assert all(J.JxlCoder.getSize(d) == (w, h) for d in datas)
    out_bytes = w * h * 4
    if os.environ.get("JXLAMD_BENCH_FILES") or c5:        # other content may decode to RGBA16
        import ctypes as _C
        _n = _C.c_size_t()
        if J.api.lib().jxlamd_output_size(data, len(data), J.api.JXLAMD_ALLOW_16BIT, _C.byref(_n)) == 0:
            out_bytes = int(_n.value)
    B = max(1, args.batch)
    total_frames = args.steps * B
    # A step's frames are issued in flights of P frames through jxlamd_decode_batch_resident: every frame is parsed, uploaded,
    # decoded and written separately (nothing is shared or cached between frames or steps), but the entropy stages of the P
    # frames of a flight go into ONE launch each.  A frame's entropy stages are serial per stream (4 LF-group + 135
    # AC wavefronts for one 4K frame) and leave the chip almost empty; a decode service fills it with frames in
    # flight.  --inflight 1 gives the strictly sequential single-frame number (also reported below).
    P = max(1, min(args.inflight, B))
    NCTX = max(1, min(args.contexts, (total_frames + P - 1) // P))
    decs = [J.JxlDecoder(local) for _ in range(NCTX)]
    SHARE = max(1, args.share)
    for c in range(NCTX):
        if c % SHARE:
            decs[c].share_pools(decs[c - c % SHARE])
    d_ins = [torch.frombuffer(bytearray(d), dtype=torch.uint8).to(f"cuda:{local}") for d in datas]   # compressed bytes resident in HBM
    c5_writer = c5 and args.c5_post == "writer"
    if c5_writer:                   # the decoder's writer emits the Bitmap format (RGBA_F16, API level 33: colour matrix + tone map apply): the decode's output IS the F16 buffer
        for dctx in decs:
            dctx.set_writer_post(True, J.PreferredColorConfig.RGBA_F16, 33)
        out_bytes = w * h * 8
    d_outs = [[torch.empty(out_bytes, dtype=torch.uint8, device=f"cuda:{local}") for _ in range(P)] for _ in range(NCTX)]
    d_f16 = [[torch.empty(w * h * 8, dtype=torch.uint8, device=f"cuda:{local}") for _ in range(P)] for _ in range(NCTX)] if (c5 and not c5_writer) else None   # the Bitmap buffers (RGBA_F16) of the separate post pass
    info0 = J.api.Info()
    J.api.lib().jxlamd_basic_info(data, len(data), J.api.C.byref(info0))
    import threading

    stagger_ms = float(os.environ.get("JXLAMD_BENCH_STAGGER_MS", "0"))

    def run_frames(n, resident=True):
        """n full decodes.  Flights of P frames; NCTX decoder contexts (own HIP stream + HBM buffers each) take flights
        alternately so that one flight's entropy stages overlap another's data-parallel stages."""
        acc = {}
        lock = threading.Lock()
        todo = []
        done = 0
        while done < n:
            p = min(P, n - done); todo.append((done, p)); done += p
        todo.reverse()
        errors = []

        def worker(c):
            try:
                _worker(c)
            except BaseException as e:  # noqa: BLE001 — re-raised in the main thread: a failed flight must fail the bench
                errors.append(e)

        def _worker(c):
            torch.cuda.set_device(local)
            if stagger_ms > 0:      # contexts enter their first flight one after another: their stages stay out of phase
                time.sleep(c * stagger_ms / 1e3)
            while True:
                with lock:
                    if not todo:
                        return
                    first, p = todo.pop()
                ids = [(rank + first + j) % len(datas) for j in range(p)]
                for attempt in (0, 1):
                    try:
                        if p == 1:
                            decs[c].decode_to_device(datas[ids[0]], d_outs[c][0].data_ptr(), out_bytes, data_dev_ptr=d_ins[ids[0]].data_ptr() if resident else None)
                        else:
                            decs[c].decode_batch_to_device([datas[i] for i in ids], [t.data_ptr() for t in d_outs[c][:p]], [out_bytes] * p,
                                                           [d_ins[i].data_ptr() for i in ids] if resident else None)
                        break
                    except J.InvalidJXLException:
                        # Safety net: a flight rejected by the decoder's own rANS final-state checks is decoded again
                        # inside the timed region and counted in "retried_flights" (0 since the kernels stopped using scratch).
                        if attempt:
                            raise
                        with lock:
                            acc["retried_flights"] = acc.get("retried_flights", 0) + 1
                t = decs[c].last_timing()
                if c5 and not c5_writer:          # A10 (Rec.2100 PQ -> Rec.2408 tone map -> sRGB) + A11 (u16 -> RGBA_F16) of every frame of the flight, one fused pass each
                    for j in range(p):
                        decs[c].post_fused_device(d_outs[c][j].data_ptr(), w, h, True, 16, True, info0.primaries, info0.transfer_function, info0.intensity_target,
                                                  J.PreferredColorConfig.RGBA_F16, False, False, 33, d_f16[c][j].data_ptr(), d_f16[c][j].numel())
                with lock:
                    for k, v in t.items():
                        acc[k] = acc.get(k, 0.0) + v
                    acc["flights"] = acc.get("flights", 0) + 1
                    acc["frames"] = acc.get("frames", 0) + p
        th = [threading.Thread(target=worker, args=(c,)) for c in range(NCTX)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errors:
            raise errors[0]
        assert acc.get("frames", 0) == n, (acc.get("frames", 0), n)
        return acc

    prime = run_frames(P * NCTX)             # untimed setup: every context allocates the HBM work buffers of a full flight
    warm = run_frames(args.warmup * B) if args.warmup > 0 else {}   # W untimed warmup steps
    if os.environ.get("JXLAMD_BENCH_ABLATE"):       # measurement only (stage floors): the timed steps leave stages of the flight out — the line is then NOT a decode rate
        J.api.lib().jxlamd_debug_set_ablate(int(os.environ["JXLAMD_BENCH_ABLATE"]))
    # sequential single-frame latency (one context; BASELINE configs[1]), reported next to the throughput
    lat = []
    for _ in range(3):
        t = time.perf_counter(); decs[0].decode_to_device(data, d_outs[0][0].data_ptr(), out_bytes, data_dev_ptr=d_ins[0].data_ptr())
        if c5 and not c5_writer:
            decs[0].post_fused_device(d_outs[0][0].data_ptr(), w, h, True, 16, True, info0.primaries, info0.transfer_function, info0.intensity_target,
                                      J.PreferredColorConfig.RGBA_F16, False, False, 33, d_f16[0][0].data_ptr(), d_f16[0][0].numel())
        lat.append(time.perf_counter() - t)
    seq_stage = decs[0].last_timing()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern = run_frames(total_frames)     # every C-ABI call returns when its pixels are in HBM
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    # the same steps with the compressed bytes arriving as HOST buffers: their H2D (one staging upload per flight) inside the timed region
    h2d_steps = max(1, args.steps)      # as many steps as the resident measurement (round 5: four steps understated it — a short run ends with idle contexts, the same effect that separates --steps 8 from --steps 20)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    h2d_acc = run_frames(h2d_steps * B, resident=False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed_h2d = max_over_ranks(time.perf_counter() - t1)

    if os.environ.get("JXLAMD_BENCH_CLOCKS"):
        # diagnostics: wall time and shader clock of the LF streams of context 0's last flight (decoded next to the other contexts)
        import ctypes as C
        import numpy as np
        L = J.api.lib()
        L.jxlamd_debug_lf_phases_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        tt = np.zeros((P, 4, 8), np.uint64)
        for f in range(P):
            L.jxlamd_debug_lf_phases_frame(decs[0]._h, f, 4, tt[f].ctypes.data)
        wall = (tt[:, :2, 6].astype(np.int64) - tt[:, :2, 0].astype(np.int64)) / 1e5
        mhz = tt[:, :2, 7].astype(np.float64) / np.maximum(wall, 1e-9) / 1e3
        print("[clocks] long LF streams of the last flight of context 0: wall ms min %.1f median %.1f max %.1f; shader clock MHz min %.0f median %.0f max %.0f"
              % (wall.min(), np.median(wall), wall.max(), mhz.min(), np.median(mhz), mhz.max()), file=sys.stderr)
        # when did each stream START relative to the first one of its launch (workgroups waiting to be placed), and when did the last one end
        t_start = tt[:, :, 0].astype(np.int64); t_end = tt[:, :, 6].astype(np.int64)
        ok = t_start > 0
        if ok.any():
            t0k = t_start[ok].min()
            st = (t_start[ok] - t0k) / 1e5
            print("[clocks] stream start after the launch's first stream (ms): median %.1f p90 %.1f max %.1f; long streams: median %.1f max %.1f; launch span %.1f ms; streams %d"
                  % (np.median(st), np.percentile(st, 90), st.max(), np.median((t_start[:, :2] - t0k) / 1e5), ((t_start[:, :2] - t0k) / 1e5).max(), (t_end[ok].max() - t0k) / 1e5, int(ok.sum())), file=sys.stderr)
    if rank == 0:
        frames = total_frames * world
        mp = w * h / 1e6
        value = frames * mp / elapsed
        mean_in = sum(len(d) for d in datas) / len(datas)
        algo_bytes = mean_in + (w * h * 8 if c5 else out_bytes)     # compressed read + RGBA (c5: RGBA_F16 Bitmap) written, per frame
        # dominant kernel of the TIMED region: the batched LF-group kernel (one launch per flight of P frames), timed
        # live with HIP events on the decoder's own stream (jxlamd_last_timing).  Algorithmic bytes per launch =
        # SURVEY.md §8(d) per-frame figure (compressed read + RGBA written) x frames per launch.
        flights = max(int(kern.get("flights", 1)), 1)
        names = {"lf_groups_ms": "k_lf_group_batch" if P > 1 else "k_lf_group", "pass_groups_ms": ("k_pass_flat (first sub-flight of the flight; + k_lf_smooth, k_pass_prep)" if P > 1 else "k_pass_group (+ k_lf_smooth)"),
                 "recon_ms": "rest of the HF phase: later sub-flights' k_pass_flat, k_recon_*, k_filter_sweep" if P > 1 else "k_recon_*", "filters_write_ms": "k_filter_sweep"}
        stages = {k: kern[k] / flights for k in names if k in kern}
        # the dominant kernel = the one with the largest total duration in rocprofv3 --stats of this command (profiles/): the two serial
        # entropy stages compete for it; both are one launch per flight and both are timed by the HIP events, so the run itself decides
        dom = "lf_groups_ms" if stages.get("lf_groups_ms", 0.0) >= stages.get("pass_groups_ms", 0.0) or P == 1 else "pass_groups_ms"
        dom_ms = stages[dom]
        frames_per_launch = total_frames / flights
        achieved = algo_bytes * frames_per_launch / (dom_ms * 1e-3) / 1e9
        seq = {k: round(v, 4) for k, v in seq_stage.items()}
        # HBM traffic of the dominant kernel from the PMC passes kept under profiles/ (FETCH_SIZE + WRITE_SIZE per frame, collected as
        # MI355X_MICROARCH.md prescribes: separate --pmc runs); null until a pass for this round's kernel is committed
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            traffic = int(tj["bytes_per_frame"][names[dom].split(" ")[0]] * frames_per_launch); traffic_src = tj.get("source")
        except Exception:  # noqa: BLE001
            pass
        line = {
            "metric": "decoded MP/s (4K Rec.2100 PQ 16-bit EPF3 -> tone map -> RGBA_F16)" if c5 else "decoded MP/s (4K VarDCT q90 -> RGBA8)", "value": round(value, 2), "unit": "MP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # which rate `value` is: the measurement contract of this build ("whole-job throughput with inputs already resident in HBM when the timed region
            # starts; the PCIe-inclusive rate is never `value`") against SURVEY.md §8(d), which counts the H2D of the compressed bytes — both are on the line
            "value_inputs": "compressed bytes resident in HBM", "value_h2d_included": round(h2d_steps * B * world * mp / elapsed_h2d, 2),
            "config": {"workload": (f"configs[4] on one GPU: one step = a batch of {B} x 3840x2160 Rec.2100 PQ 16-bit VarDCT (distance 1.0, effort 7, EPF forced to 3 iterations) "
                                    + ("frames -> colour matrix + Rec.2408 tone map -> RGBA_F16 INSIDE the decoder's writer (the last EPF stage emits the Bitmap format; the RGBA16 image is never stored) " if c5_writer else
                                       "frames -> RGBA16 -> colour matrix + Rec.2408 tone map -> RGBA_F16 (A10 + A11 fused into one pass per frame) ") if c5 else
                                    f"configs[2]: one step = a batch of {B} x 3840x2160 VarDCT q90 (distance 1.0, effort 7) RGB frames -> RGBA8 per GPU ") +
                                   f"({len(datas)} distinct seeded frames" + (" generated on this box by the reference's encoder" if len(datas) > len(FRAMES) else " cycled") +
                                   "), every frame a complete decode (host parse, table upload, all kernels); value: compressed bytes resident in HBM when the "
                                   "timed region starts; h2d_included_MPps: the same steps fed from host buffers (H2D included); RGBA output stays in HBM",
                       "h2d_included_MPps": round(h2d_steps * B * world * mp / elapsed_h2d, 2), "h2d_included_steps": h2d_steps, "distinct_frames": len(datas),
                       "frame_bytes_mean": int(mean_in), "frames_per_step_per_gpu": B, "frames_in_flight": P, "decoder_contexts": NCTX, "contexts_per_pool_set": SHARE, "contexts_on_general_lf_kernel": _general_contexts(J, decs), "flights_repeated_for_lf_pool": _lf_retries(J, decs)[0], "lf_pool_bytes": sorted(set(_lf_retries(J, decs)[1])), "retried_flights": int(kern.get("retried_flights", 0)),
                       "contexts_on_sparse_coefficient_lists": _sparse_state(J, decs)[0], "flights_repeated_with_dense_coefficients": _sparse_state(J, decs)[1],
                       "single_frame_latency_ms": round(min(lat) * 1e3, 3), "single_frame_MPps": round(mp / min(lat), 2),
                       "single_frame_stage_ms": seq,
                       "parallelism": f"frames sharded over {world} GPU(s), one process per GPU, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": names[dom], "kernel_ms": round(dom_ms, 4), "launches": flights,
                         # the same HIP-event average over EVERY batched launch of the process (priming + warm-up + timed + the H2D-inclusive
                         # steps): the figure to hold against the rocprofv3 --stats average of this command, which cannot tell them apart
                         "kernel_ms_all_launches": round(sum(a.get(dom, 0.0) for a in (prime, warm, kern, h2d_acc) if a) /
                                                         max(sum(int(a.get("flights", 0)) for a in (prime, warm, kern, h2d_acc) if a), 1), 4),
                         "algorithmic_bytes_per_launch": int(algo_bytes * frames_per_launch),
                         "whole_pipeline_GBps": round(value * 1e6 * (algo_bytes / (w * h)) / 1e9 / world, 2),
                         "stage_ms_per_flight": {k: round(v, 4) for k, v in stages.items()},
                         "note": "the entropy-decode kernels are latency/occupancy-bound (serial rANS streams), not "
                                 "bandwidth-bound; achieved = algorithmic bytes / duration of the dominant kernel"},
        }
        line["cpu_baseline"] = ({"value": None, "unit": "MP/s", "cores": 0, "kind": "reference", "sample": "skipped"}
                                if (args.no_cpu_baseline or world > 1) else cpu_baseline(data, c5=c5))
        # BASELINE configs[1] (one frame, latency) stated, not left to divide: this decoder's single-frame rate over the reference's libjxl decoding the same
        # frame with its thread pool on this box's host cores.  Below 1: a lone frame is bounded by ONE LfGroup stream — 196 608 samples walked by one lane at ~1 100 shader
        # clocks each (k_lf_group_general, ~90 ms of the ~112) — where a host core takes a few tens of cycles per sample; the GPU wins on batches, not on a lone frame.
        if os.environ.get("JXLAMD_BENCH_ABLATE"):
            line["config"]["ABLATED_STAGES_not_a_decode_rate"] = int(os.environ["JXLAMD_BENCH_ABLATE"])
        thr = line["cpu_baseline"].get("single_frame_threaded_MPps")
        line["config"]["single_frame_vs_cpu_threaded"] = round(line["config"]["single_frame_MPps"] / thr, 3) if thr else None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
