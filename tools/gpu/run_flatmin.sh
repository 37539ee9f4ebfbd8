R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in "4096 c5 16 16" "1024 c5 16 16" "512 c5 16 16" "4096 c3 16 16" "1024 c3 16 16" "4096 c3 16 8" "512 c3 16 8" "4096 c3 16 24" "1024 c3 16 24"; do set -- $cfg
  JXLAMD_FLAT_MIN_GROUPS=$1 python bench.py --workload $2 --no-cpu-baseline --distinct 0 --steps 12 --warmup 3 --contexts $3 --inflight $4 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith(chr(123)): d=json.loads(l); print('flat_min $1 $2 $3 x $4', d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_flight'])
"; done
