# HBM traffic of the filter stage, per-stage kernels vs the fused LDS-tiled kernel (separate --pmc passes for FETCH_SIZE / WRITE_SIZE).
ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_filters; mkdir -p $O
cd /tmp
for fused in 0 1; do
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmcf
  JXLAMD_FUSED_FILTERS=$fused PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcf -o p -- python $R/tools/prof_decode.py 2 > /tmp/pmcf.log 2>&1
  f=$(find /tmp/pmcf -name '*counter_collection.csv' | head -1)
  python - "$f" "$O/fused${fused}_$set.json" <<'PY'
import csv, sys, json, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(float); cnt = collections.Counter(); seen = set()
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('jxlamd::', '').replace('void ', '')
    acc[k] += float(r['Counter_Value'])
    key = (k, r['Dispatch_Id'])
    if key not in seen: seen.add(key); cnt[k] += 1
out = {k: {'per_dispatch': v / cnt[k], 'dispatches': cnt[k]} for k, v in acc.items() if k.startswith('k_')}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(sys.argv[2].split('/')[-1], {k: round(v['per_dispatch']) for k, v in out.items()})
PY
done
done
