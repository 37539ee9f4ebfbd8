#!/usr/bin/env python3
"""Kernel timeline of a bench run (rocprofv3 --kernel-trace CSV): how the stages of the concurrent flights share the chip in time.
Prints, for the steady-state half of the run: per kernel family the busy time (union of its dispatches), the mean number of dispatches
in flight, the share of wall time with no entropy kernel / no data-parallel kernel / nothing running, and the stretch of every family's
dispatches against its fastest ones."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("jxlamd::", "").replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r.get("Queue_Id", 0) or 0)))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo, hi = t0 + (t1 - t0) * 0.35, t0 + (t1 - t0) * 0.85          # steady state: skip priming / warm-up and the tail
def fam(n):
    if n.startswith("k_lf_group"): return "LF"
    if n.startswith("k_pass"): return "PASS"
    if n.startswith("k_recon"): return "RECON"
    if n.startswith("k_filter"): return "FILTER"
    return "other"
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot
win = hi - lo
byf = collections.defaultdict(list)
for s, e, n, q in ev:
    s2, e2 = max(s, lo), min(e, hi)
    if e2 > s2: byf[fam(n)].append((s2, e2))
print("window %.1f ms of %.1f ms; queues %d" % (win / 1e6, (t1 - t0) / 1e6, len({e[3] for e in ev})))
for f, iv in sorted(byf.items()):
    print("%-7s busy (union) %5.1f %%  mean dispatches in flight %5.2f  dispatches %d" % (f, 100 * union(iv) / win, sum(e - s for s, e in iv) / win, len(iv)))
allk = [x for iv in byf.values() for x in iv]
print("anything running %.1f %%" % (100 * union(allk) / win))
dp = byf["RECON"] + byf["FILTER"]; ent = byf["LF"] + byf["PASS"]
print("data-parallel (recon + filter) running %.1f %%, entropy running %.1f %%" % (100 * union(dp) / win, 100 * union(ent) / win))
# per kernel: duration distribution (stretch in the mix)
byk = collections.defaultdict(list)
for s, e, n, q in ev:
    if lo <= s <= hi: byk[n].append((e - s) / 1e6)
for n, d in sorted(byk.items(), key=lambda kv: -sum(kv[1]))[:10]:
    d.sort()
    print("%-28s n %4d  min %8.3f  p50 %8.3f  p90 %8.3f  max %8.3f ms  sum %9.1f" % (n[:28], len(d), d[0], d[len(d) // 2], d[len(d) * 9 // 10], d[-1], sum(d)))
# how many distinct flights' LF kernels overlap, sampled
import bisect
lf = sorted(byf["LF"]); ps = sorted(byf["PASS"])
samples = [lo + win * i / 400 for i in range(400)]
def active(iv, t): return sum(1 for s, e in iv if s <= t < e)
a_lf = [active(lf, t) for t in samples]; a_ps = [active(ps, t) for t in samples]; a_dp = [active(dp, t) for t in samples]
print("concurrent LF kernels: mean %.1f max %d | PASS mean %.1f max %d | recon+filter mean %.1f max %d" % (sum(a_lf) / 400, max(a_lf), sum(a_ps) / 400, max(a_ps), sum(a_dp) / 400, max(a_dp)))
