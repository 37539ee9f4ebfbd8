# GPU-box smoke: the -m gpu parity tests, smoke(), one default bench line.   /usr/local/graft/bin/gpurun -- 'bash tools/gpu/run_check.sh'
ulimit -c 0
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
