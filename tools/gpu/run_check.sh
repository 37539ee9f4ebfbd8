ulimit -c 0
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/prof_decode.py 4 2>&1 | tail -5
timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100
