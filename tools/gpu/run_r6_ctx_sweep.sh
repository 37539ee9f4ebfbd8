# contexts x frames in flight on the round's final build (the default 16 x 64 dates from round 4): alternating passes.  CFGS="16x64 12x64 ..." REPS="1 2"
ulimit -c 0; O=gpurun_out/ctxsweep; mkdir -p $O
for rep in ${REPS:-1 2}; do for cfg in ${CFGS:-16x64 24x64 32x64 16x32 32x32 12x64}; do
  c=${cfg%x*}; f=${cfg#*x}
  timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --contexts $c --inflight $f 2>$O/err_${cfg}_$rep.txt | tail -1 > $O/${cfg}_$rep.json
  python -c "import json; d=json.load(open('$O/${cfg}_$rep.json')); print('[ctx] contexts $c inflight $f rep $rep value', d['value'], 'ms/step', d['ms_per_step'], d['roofline']['stage_ms_per_flight'])" || tail -3 $O/err_${cfg}_$rep.txt
done; done
