# kernel trace of a short bench run: per-queue timeline (tools/gpu/trace_summary.py) + the raw trace of one queue
ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; rm -rf /tmp/prof
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > /tmp/bench.log 2>&1
grep -v "^[WE]2026" /tmp/bench.log | tail -1 | cut -c1-300
mkdir -p $R/gpurun_out/trace
python $R/tools/gpu/trace_summary.py /tmp/prof/bench_kernel_trace.csv 0 > $R/gpurun_out/trace/summary.txt 2>&1
python - /tmp/prof/bench_kernel_trace.csv $R/gpurun_out/trace/timeline.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r['Start_Timestamp']) for r in rows)
byq = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('jxlamd::', '').replace('void ', '')
    byq[r.get('Queue_Id', '?')].append((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, n, r.get('Stream_Id', '?'), r.get('Grid_Size', '?'), r.get('Workgroup_Size', '?')))
with open(sys.argv[2], 'w') as f:
    f.write('columns: ' + ','.join(rows[0].keys()) + '\n')
    for q in sorted(byq, key=lambda k: -len(byq[k]))[:3]:
        l = sorted(byq[q])
        f.write(f'== queue {q}: {len(l)} kernels\n')
        prev = None
        for s, e, n, st, g, wg in l[-260:]:
            gap = (s - prev) / 1e6 if prev is not None else 0
            f.write(f'{s/1e6:10.3f} +{(e-s)/1e6:9.3f} ms gap {gap:8.3f}  {n[:40]:40s} stream {st} grid {g} wg {wg}\n')
            prev = e
PY
tail -25 $R/gpurun_out/trace/summary.txt
