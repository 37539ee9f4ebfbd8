# Counters of the batched (flight) kernels: one flight of 128 frames through one decoder context, separate --pmc passes
# (instruction mix / HBM fetch / HBM write).  Prints the largest dispatch of each kernel (= a full flight / sub-batch) and writes
# gpurun_out/pmc_batch/flight128.json.
ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_batch; mkdir -p $O
cd /tmp
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE")
# PMC_SETS="a b;c d": other counter sets, one pass each (e.g. the cache-request counters)
if [ -n "$PMC_SETS" ]; then IFS=';' read -ra SETS <<< "$PMC_SETS"; fi
for set in "${SETS[@]}"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  rm -rf /tmp/pmcb
  PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcb -o p -- python $R/bench.py --no-cpu-baseline --distinct 0 --steps 1 --batch 128 --inflight 128 --contexts 1 --warmup 0 > /tmp/pmcb.log 2>&1
  f=$(find /tmp/pmcb -name '*counter_collection.csv' | head -1)
  python - "$f" "$O/$tag.json" <<'PY'
import csv, sys, json, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('jxlamd::', '').replace('void ', '')
    acc[k][r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
res = {}
for k, d in acc.items():
    if not k.startswith('k_'): continue
    out = {}
    for c, l in d.items():
        per = collections.defaultdict(float)
        for did, v in l: per[did] += v
        out[c] = round(max(per.values()))      # largest dispatch (a full flight)
    res[k] = out
    print(k, out)
json.dump(res, open(sys.argv[2], 'w'), indent=1)
PY
done
python - "$O" <<'PY'
import json, glob, sys, os
m = {}
for f in glob.glob(os.path.join(sys.argv[1], '*.json')):
    if f.endswith('flight128.json'): continue
    for k, d in json.load(open(f)).items(): m.setdefault(k, {}).update(d)
json.dump(m, open(os.path.join(sys.argv[1], 'flight128.json'), 'w'), indent=1)
PY
