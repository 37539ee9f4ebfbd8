# convoy test: do the 16 contexts' stages run in phase?  Same bench with the contexts' first flights staggered.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/stagger; mkdir -p $O; cd $R
for st in 0 40 0 40 80; do
  JXLAMD_BENCH_STAGGER_MS=$st python bench.py --no-cpu-baseline --distinct 0 --steps 60 --warmup 4 2>$O/err_$st.txt > $O/s_$st.json
  python - $O/s_$st.json $st <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("stagger", sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], d["roofline"]["stage_ms_per_flight"])
PY
done
