R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in "8 32" "16 32" "12 32" "16 24" "16 16"; do set -- $cfg
  python bench.py --workload c5 --no-cpu-baseline --steps 12 --warmup 3 --contexts $1 --inflight $2 2>/tmp/c5err.txt | python -c "
import json,sys
ok=False
for l in sys.stdin:
    if l.startswith(chr(123)): d=json.loads(l); ok=True; print('c5 $1 x $2', d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_flight'], d['config'].get('h2d_included_MPps'))
if not ok: print('c5 $1 x $2 failed')
"; tail -1 /tmp/c5err.txt | cut -c1-200; done
