# A/B the in-tree libjxlamd.so against jxl_coder_amd/libjxlamd_<name>.so with bench.py (alternating, 2 repetitions each).
# Usage: bash tools/gpu/run_ab2.sh <name> [bench args...]     Output: gpurun_out/ab2.log
ulimit -c 0
mkdir -p gpurun_out; : > gpurun_out/ab2.log
name=$1; shift
for rep in 1 2; do for v in main $name; do
  if [ $v = main ]; then unset JXLAMD_LIB; else export JXLAMD_LIB=$PWD/jxl_coder_amd/libjxlamd_$v.so; fi
  timeout 900 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['stage_ms_per_flight'])" >> gpurun_out/ab2.log 2>&1
done; done
unset JXLAMD_LIB
cat gpurun_out/ab2.log
