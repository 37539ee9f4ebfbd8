# PMC passes (counters only, no trace domains besides --kernel-trace) over sequential single-frame 4K decodes.
ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc; mkdir -p $O
cd /tmp
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE")
# PMC_SETS="a b;c d": other counter sets, one pass each
if [ -n "$PMC_SETS" ]; then IFS=';' read -ra SETS <<< "$PMC_SETS"; fi
for set in "${SETS[@]}"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$tag
  PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tools/prof_decode.py 2 > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
  python - "$f" "$O/$tag.json" <<'PY'
import csv, sys, json, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('jxlamd::', '').replace('void ', '')
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (k, r['Dispatch_Id'])
    if key not in seen: seen.add(key); cnt[k] += 1
out = {k: {c: v / cnt[k] for c, v in d.items()} | {'dispatches': cnt[k]} for k, d in acc.items()}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
for k, d in out.items():
    if k.startswith('k_'): print(k, {c: (round(v) if isinstance(v, float) else v) for c, v in d.items()})
PY
done
