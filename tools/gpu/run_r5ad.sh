# round 5: instruction counts of the block-form loop by removal (JXLAMD_DEBUG_MOD 0 / 8 / 16 / 56, see run_r5ac.sh): rocprofv3 --pmc over three decodes of the
# default lossy RGBA 4K frame, k_mod_group only.  Counters only (+ --kernel-trace); one pass per set.
ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5ad; mkdir -p $O
cd $R
python - <<'PY'
import sys; sys.path[:0] = ['.', 'oracle', 'tools']
import jxl_ref, synth
open('/tmp/rgba4k_d1.jxl', 'wb').write(jxl_ref.encode(synth.photo_like(3840, 2160, seed=4, channels=4), effort=7, distance=1.0))
PY
cd /tmp
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC")
for v in 0 8 16 56; do
 for set in "${SETS[@]}"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$tag
  JXLAMD_DEBUG_MOD=$v JXLAMD_PROF_FILE=/tmp/rgba4k_d1.jxl PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tools/prof_decode.py 3 > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "dbg $v set [$set]: no counters"; tail -3 /tmp/pmc_$tag.log; continue; fi
  python - "$f" "$O/dbg${v}_$tag.json" $v <<'PY'
import csv, sys, json, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('jxlamd::', '').replace('void ', '')
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (k, r['Dispatch_Id'])
    if key not in seen: seen.add(key); cnt[k] += 1
out = {k: {c: v / cnt[k] for c, v in d.items()} | {'dispatches': cnt[k]} for k, d in acc.items()}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
for k, d in out.items():
    if k.startswith('k_mod_group') or k.startswith('k_pass_group'): print('dbg', sys.argv[3], k, {c: (round(v) if isinstance(v, float) else v) for c, v in d.items()})
PY
 done
done
