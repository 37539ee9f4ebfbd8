# Time the single-frame LF phases of several builds of libjxlamd.so on the same box: the in-tree build + jxl_coder_amd/libjxlamd_<name>.so
# for each name given.  Usage: bash tools/gpu/run_variants.sh v1 v3 ...   (output: gpurun_out/variants.log)
ulimit -c 0
mkdir -p gpurun_out
: > gpurun_out/variants.log
for v in main "$@"; do
  if [ $v = main ]; then unset JXLAMD_LIB; else export JXLAMD_LIB=$PWD/jxl_coder_amd/libjxlamd_$v.so; fi
  echo "== $v" >> gpurun_out/variants.log
  timeout 300 python tools/prof_decode.py 3 2>&1 | tail -6 >> gpurun_out/variants.log
done
unset JXLAMD_LIB
cat gpurun_out/variants.log
