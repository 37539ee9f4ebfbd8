ulimit -c 0
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o a -- python $R/tools/gpu/prof_asset.py asset_wide_gamut 5 > /tmp/a.log 2>&1
grep asset /tmp/a.log | tail -2
head -9 /tmp/pa/a_kernel_stats.csv | cut -c1-150
