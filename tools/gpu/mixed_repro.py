"""Which frame kinds of bench.py --workload mixed decode (single decodes, one flight, several contexts), compared with the reference run live."""
import os, sys, subprocess, threading, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import jxl_coder_amd as J, jxl_ref
out = "/tmp/jxlamd_bench_frames"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_bench_frames.py"), "--out", out, "--count", str(n), "--kind", "mixed"], cwd="/tmp")
datas = [open(os.path.join(out, f"mixed_seed{i}.jxl"), "rb").read() for i in range(n)]
dec = J.JxlDecoder(0)
singles = []
for i, d in enumerate(datas):
    try:
        o, info = dec.decode_one_shot(d)
        diff = -1
        if not os.environ.get("FAST"):
            r, _, _ = jxl_ref.decode(d, threads=0)
            diff = int(np.abs(o.astype(int) - np.asarray(r).astype(int)).max()) if r.dtype == np.uint8 and r.shape == o.shape else -1
        print("single", i, o.shape, "max diff vs reference", diff, dec.last_timing().get("device_total_ms")); singles.append(o)
    except Exception as e:
        print("single", i, "FAILED", e); singles.append(None)
# one flight of all frames, then 8 contexts x flights concurrently
FL = int(os.environ.get("FLIGHT", "32"))
def flight(dc, tag, lo=0):
    idx = [i for i in range(lo, min(lo + FL, n)) if singles[i] is not None]
    outs = [torch.empty(singles[i].size, dtype=torch.uint8, device="cuda:0") for i in idx]
    ins = [torch.frombuffer(bytearray(datas[i]), dtype=torch.uint8).to("cuda:0") for i in idx]
    try:
        dc.decode_batch_to_device([datas[i] for i in idx], [t.data_ptr() for t in outs], [t.numel() for t in outs], [t.data_ptr() for t in ins])
        torch.cuda.synchronize()
        bad = [i for t, i in zip(outs, idx) if not np.array_equal(t.cpu().numpy().reshape(singles[i].shape), singles[i])]
        print(tag, "flight from", lo, "frames that differ from their single decode:", bad)
    except Exception as e:
        print(tag, "flight from", lo, "FAILED", e)
for lo in range(0, n, FL): flight(dec, "ctx0", lo)
decs = [J.JxlDecoder(0) for _ in range(8)]
for rep in range(3):
    ts = [threading.Thread(target=flight, args=(decs[k], f"rep{rep} ctx{k}", (k * FL) % n)) for k in range(8)]
    [t.start() for t in ts]; [t.join() for t in ts]
