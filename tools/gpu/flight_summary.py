#!/usr/bin/env python3
"""Summarise the JXLAMD_TRACE_FLIGHT=1 lines of a bench run (stderr): per flight the LF streams' start offsets / durations and the stage times.
usage: flight_summary.py bench.err [skip_first_flights]   (the first flights of every context include the one-time uploads: skipped by default = 16)"""
import re, sys, statistics as st
txt = open(sys.argv[1], errors="replace").read()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 16
fl = re.findall(r"\[flight \S+\] begin \S+ end \S+ \| n=(\d+) parse (\S+) prepare (\S+) launch (\S+) wait\+collect (\S+) ms \| GPU: uploads (\S+) LF (\S+) pass0 (\S+) rest (\S+)", txt)
lf = re.findall(r"LF streams (\d+): start offsets ms p50 (\S+) p75 (\S+) p90 (\S+) max (\S+) \| durations ms p50 (\S+) p90 (\S+) max (\S+) \| first start -> last end (\S+) ms \(stage (\S+)\)", txt)
fl, lf = fl[skip:], lf[skip:]
if not fl:
    sys.exit("no flight lines")
med = lambda xs: round(st.median(xs), 1)
print("flights %d | GPU stage ms (median): LF %s pass0 %s rest %s | host: parse %s wait+collect %s" % (
    len(fl), med([float(f[6]) for f in fl]), med([float(f[7]) for f in fl]), med([float(f[8]) for f in fl]), med([float(f[1]) for f in fl]), med([float(f[4]) for f in fl])))
if lf:
    print("LF streams: start offset p50 %s p90 %s (medians over flights; worst p90 %s) | duration p50 %s p90 %s | first start -> last end %s" % (
        med([float(x[1]) for x in lf]), med([float(x[3]) for x in lf]), max(float(x[3]) for x in lf), med([float(x[5]) for x in lf]), med([float(x[6]) for x in lf]), med([float(x[8]) for x in lf])))
