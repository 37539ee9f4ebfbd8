# single-frame LF phases of the in-tree library vs variants (JXLAMD_LIB), alternating; also checks the pixels against the main library's
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do for v in main "$@"; do
  if [ $v = main ]; then unset JXLAMD_LIB; else export JXLAMD_LIB=$R/jxl_coder_amd/libjxlamd_$v.so; fi
  python tools/prof_decode.py 5 2>&1 | grep -E "^lf group 0|^4k" | tail -2 | cut -c1-260 | sed "s/^/$v: /"
  python - <<'PY'
import os, sys, hashlib
sys.path.insert(0, os.getcwd())
import jxl_coder_amd as J
d = J.JxlDecoder(0); out, _ = d.decode_one_shot(open("bench_data/syn4k_q90_seed0.jxl", "rb").read()); print("   pixels md5", hashlib.md5(out.tobytes()).hexdigest()[:12])
PY
done; done
