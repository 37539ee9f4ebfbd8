#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV: per-queue timeline of the long kernels and the union busy time (test/prof tool)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r['Start_Timestamp']) for r in rows)
ev = []
for r in rows:
    n = r['Kernel_Name']
    short = n.split('(')[0].replace('jxlamd::', '').replace('void ', '')
    ev.append((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, short, r.get('Queue_Id', '?')))
ev.sort()
tmax = max(e[1] for e in ev)
print('kernels', len(ev), 'span ms', tmax / 1e6)
big = [e for e in ev if e[1] - e[0] > 2e6]
for s, e, n, q in big[: int(sys.argv[2]) if len(sys.argv) > 2 else 80]:
    print(f'q{q:>3} {s/1e6:9.1f} -> {e/1e6:9.1f}  ({(e-s)/1e6:7.1f} ms) {n}')
# per-queue gaps: time between consecutive kernels on the same queue > 5 ms
byq = collections.defaultdict(list)
for e in ev: byq[e[3]].append(e)
for q, l in byq.items():
    busy = sum(e[1] - e[0] for e in l)
    print(f'queue {q}: {len(l)} kernels, busy {busy/1e6:.1f} ms of span {(l[-1][1]-l[0][0])/1e6:.1f} ms')
