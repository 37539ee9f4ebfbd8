# kernel times of a default-settings lossy RGBA frame at 4K (VarDCT colour + squeezed, quantised alpha: what `cjxl -d 1` makes of a PNG with alpha): encoded on the
# box by the reference (oracle/_ref), decoded five times under rocprofv3 --kernel-trace --stats
ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/rgba4k; mkdir -p $O
cd $R
python - <<'PY'
import sys; sys.path[:0] = ['.', 'oracle', 'tools']
import jxl_ref, synth
img = synth.photo_like(3840, 2160, seed=4, channels=4)
open('/tmp/rgba4k_d1.jxl', 'wb').write(jxl_ref.encode(img, effort=7, distance=1.0))
open('/tmp/rgba4k_lossless_e3.jxl', 'wb').write(jxl_ref.encode(img, lossless=True, effort=3))
PY
cp /tmp/rgba4k_d1.jxl $R/gpurun_out/rgba4k/ 2>/dev/null
cd /tmp; rm -rf /tmp/profa
JXLAMD_PROF_FILE=/tmp/rgba4k_d1.jxl PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profa -o rgba -- python $R/tools/prof_decode.py 5 > $O/rgba4k_d1.log 2>&1
cp /tmp/profa/rgba_kernel_stats.csv $O/kernel_stats_rgba4k_d1.csv
grep "4k " $O/rgba4k_d1.log | tail -3
head -14 $O/kernel_stats_rgba4k_d1.csv | cut -c1-150
rm -rf /tmp/profb
JXLAMD_PROF_FILE=/tmp/rgba4k_lossless_e3.jxl PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb -o rgbal -- python $R/tools/prof_decode.py 5 > $O/rgba4k_lossless.log 2>&1
cp /tmp/profb/rgbal_kernel_stats.csv $O/kernel_stats_rgba4k_lossless_e3.csv
grep "4k " $O/rgba4k_lossless.log | tail -3
head -10 $O/kernel_stats_rgba4k_lossless_e3.csv | cut -c1-150
