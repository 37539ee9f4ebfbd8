R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ctx; mkdir -p $O; cd $R
for cfg in "16 64" "16 48" "16 32" "16 40" "16 64" "16 48"; do set -- $cfg
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --contexts $1 --inflight $2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith(chr(123)): d=json.loads(l); print('$1 x $2', d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_flight'], d['config'].get('h2d_included_MPps'))
"; done
