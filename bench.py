#!/usr/bin/env python3
"""bench.py — decoded MP/s of the JPEG XL decode hot path on MI355X (BASELINE.json metric).

Workload at N=1: BASELINE.json configs[1] — one 3840x2160 VarDCT q90 (distance 1.0, effort 7) RGB frame -> RGBA8.
A "step" = one full decode of that frame: host header/TOC/global-table parse, H2D of the frame tables, all HIP
kernels (LF/modular + AC entropy decode, dequant + inverse DCT, Gaborish/EPF, XYB->RGBA).  The compressed bytes and
the RGBA output are resident in HBM (jxlamd_decode_resident + JXLAMD_OUT_DEVICE); nothing is cached between steps.
N>1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank decodes its own frames — independent
units, no data-path collective — weak scaling; value = frames of all ranks / max-over-ranks time.
"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # more hardware queues: decoder contexts = HIP streams that must overlap (default is 4)
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME = os.path.join(ROOT, "bench_data", "syn4k_q90_seed0.jxl")
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (guides/MI355X_MICROARCH.md)


def cpu_baseline(data, budget_s=12.0):
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
        threaded = mp / best
        # batches of frames (the bench's workload): a pool of single-threaded reference decoders, one process per host core
        # (SURVEY.md §8d iii), in a fresh process tree (no fork of this CUDA-initialised process)
        pool = None
        try:
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
                          f"1 thread: {mp / one:.1f} MP/s; libjxl 0.12.0 Android-x86_64 SSE2-only build under bionic shim"}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "MP/s", "cores": 0, "kind": "reference", "sample": f"CPU baseline unavailable: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--contexts", type=int, default=8, help="decoder contexts taking flights alternately (overlaps one flight's LF stage with another's later stages)")
    ap.add_argument("--inflight", type=int, default=128, help="frames decoded per batched flight (1 = strictly sequential)")
    args = ap.parse_args()

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

    data = open(FRAME, "rb").read()
    w, h = J.JxlCoder.getSize(data)
    out_bytes = w * h * 4
    # Steps are issued in flights of P frames through jxlamd_decode_batch_resident: every frame is parsed, uploaded,
    # decoded and written separately (nothing is shared or cached between steps), but the entropy stages of the P
    # frames of a flight go into ONE launch each.  A frame's entropy stages are serial per stream (4 LF-group + 135
    # AC wavefronts for one 4K frame) and leave the chip almost empty; a decode service fills it with frames in
    # flight.  --inflight 1 gives the strictly sequential single-frame number (also reported below).
    P = max(1, min(args.inflight, args.steps))
    NCTX = max(1, min(args.contexts, (args.steps + P - 1) // P))
    decs = [J.JxlDecoder(local) for _ in range(NCTX)]
    dec = decs[0]
    d_in = torch.frombuffer(bytearray(data) + bytearray(64), dtype=torch.uint8).to(f"cuda:{local}")   # compressed bytes resident in HBM
    d_outs = [[torch.empty(out_bytes, dtype=torch.uint8, device=f"cuda:{local}") for _ in range(P)] for _ in range(NCTX)]
    import threading

    def run_steps(n):
        """n full decodes.  Flights of P frames; NCTX decoder contexts (own HIP stream + HBM buffers each) take flights
        alternately so that one flight's LF stage (256 wavefronts on the whole chip) overlaps another's later stages."""
        acc = {}
        lock = threading.Lock()
        todo = []
        done = 0
        while done < n:
            p = min(P, n - done); todo.append(p); done += p

        errors = []

        def worker(c):
            try:
                _worker(c)
            except BaseException as e:  # noqa: BLE001 — re-raised in the main thread: a failed flight must fail the bench
                errors.append(e)

        def _worker(c):
            torch.cuda.set_device(local)
            while True:
                with lock:
                    if not todo:
                        return
                    p = todo.pop()
                for attempt in (0, 1):
                    try:
                        if p == 1:
                            decs[c].decode_to_device(data, d_outs[c][0].data_ptr(), out_bytes, data_dev_ptr=d_in.data_ptr())
                        else:
                            decs[c].decode_batch_to_device([data] * p, [t.data_ptr() for t in d_outs[c][:p]], [out_bytes] * p, [d_in.data_ptr()] * p)
                        break
                    except J.InvalidJXLException:
                        # Safety net: a flight rejected by the decoder's own rANS final-state checks is decoded again
                        # inside the timed region and counted in "retried_flights" (0 since the kernels stopped using scratch).
                        if attempt:
                            raise
                        with lock:
                            acc["retried_flights"] = acc.get("retried_flights", 0) + 1
                t = decs[c].last_timing()
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

    prime = run_steps(P * NCTX)              # untimed setup: every context allocates the HBM work buffers of a full flight
    warm = run_steps(max(args.warmup, 0)) if args.warmup > 0 else {}   # W untimed warmup steps
    # sequential single-frame latency (one context), reported next to the throughput
    lat = []
    for _ in range(3):
        t = time.perf_counter(); decs[0].decode_to_device(data, d_outs[0][0].data_ptr(), out_bytes, data_dev_ptr=d_in.data_ptr()); lat.append(time.perf_counter() - t)
    seq_stage = decs[0].last_timing()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern = run_steps(args.steps)     # every C-ABI call returns when its pixels are in HBM
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    if rank == 0:
        frames = args.steps * world
        mp = w * h / 1e6
        value = frames * mp / elapsed
        algo_bytes = len(data) + out_bytes                       # compressed read + RGBA written, per frame
        # dominant kernel of the TIMED region: the batched LF-group kernel (one launch per flight of P frames), timed
        # live with HIP events on the decoder's own stream (jxlamd_last_timing).  Algorithmic bytes per launch =
        # SURVEY.md §8(d) per-frame figure (compressed read + RGBA written) x frames per launch.
        flights = max(int(kern.get("flights", 1)), 1)
        names = {"lf_groups_ms": "k_lf_group_batch" if P > 1 else "k_lf_group", "pass_groups_ms": ("k_pass_group_simt of the first sub-flight" if P > 1 else "k_pass_group") + " (+k_lf_smooth)",
                 "recon_ms": "rest of the HF phase: later sub-flights' k_pass_group_simt, k_recon_*, k_filter_b<*>" if P > 1 else "k_recon_small_b+k_recon_list_b", "filters_write_ms": "k_filter_b<0..4>"}
        stages = {k: kern[k] / flights for k in names if k in kern}
        dom = "lf_groups_ms"      # rocprofv3 --stats of this command: k_lf_group_batch has the largest total (profiles/r01_*bench.csv)
        dom_ms = stages[dom]
        frames_per_launch = args.steps / flights
        achieved = algo_bytes * frames_per_launch / (dom_ms * 1e-3) / 1e9
        seq = {k: round(v, 4) for k, v in seq_stage.items()}
        line = {
            "metric": "decoded MP/s (4K VarDCT q90 -> RGBA8)", "value": round(value, 2), "unit": "MP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: single 3840x2160 VarDCT q90 (distance 1.0, effort 7) RGB frame -> RGBA8 per step, "
                                   "compressed input and RGBA output resident in HBM; steps issued in flights of frames_in_flight frames",
                       "frame_bytes": len(data), "frames_per_step_per_gpu": 1, "frames_in_flight": P, "decoder_contexts": NCTX, "retried_flights": int(kern.get("retried_flights", 0)),
                       "single_frame_latency_ms": round(min(lat) * 1e3, 3), "single_frame_MPps": round(mp / min(lat), 2),
                       "single_frame_stage_ms": seq,
                       "parallelism": f"frames sharded over {world} GPU(s), no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6),
                         # PMC passes (profiles/r01_pmc_fetch_write_4k_single_frame.json): FETCH 1 869 KB + WRITE 4 479 KB per frame
                         "traffic": int(frames_per_launch * (1868.9 + 4478.8) * 1024),
                         "kernel": names[dom], "kernel_ms": round(dom_ms, 4), "launches": flights,
                         # the same HIP-event average over EVERY batched launch of the process (priming + warm-up + timed): the
                         # figure to hold against the rocprofv3 --stats average of this command, which cannot tell them apart
                         "kernel_ms_all_launches": round(sum(a.get(dom, 0.0) for a in (prime, warm, kern) if a.get("frames", 0) > 1 or a is kern) /
                                                         max(sum(int(a.get("flights", 0)) for a in (prime, warm, kern) if a.get("frames", 0) > 1 or a is kern), 1), 4),
                         "algorithmic_bytes_per_launch": int(algo_bytes * frames_per_launch),
                         "stage_ms_per_flight": {k: round(v, 4) for k, v in stages.items()},
                         "note": "the entropy-decode kernels are latency/occupancy-bound (one wavefront per serial rANS stream), not "
                                 "bandwidth-bound; achieved = algorithmic bytes / duration of the dominant kernel"},
        }
        line["cpu_baseline"] = ({"value": None, "unit": "MP/s", "cores": 0, "kind": "reference", "sample": "skipped"}
                                if (args.no_cpu_baseline or world > 1) else cpu_baseline(data))
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
