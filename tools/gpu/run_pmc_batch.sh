# instruction counters of the batched kernels: one flight of 128 frames through one context
ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; rm -rf /tmp/pmcb
PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d /tmp/pmcb -o p -- python $R/bench.py --no-cpu-baseline --steps 128 --inflight 128 --contexts 1 --warmup 0 > /tmp/pmcb.log 2>&1
f=$(find /tmp/pmcb -name '*counter_collection.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('jxlamd::', '').replace('void ', '')
    acc[k][r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
for k, d in acc.items():
    if not k.startswith('k_'): continue
    out = {}
    for c, l in d.items():
        per = collections.defaultdict(float)
        for did, v in l: per[did] += v
        vals = sorted(per.values())
        out[c] = round(vals[-1])      # largest dispatch (a full flight)
    print(k, out)
PY
