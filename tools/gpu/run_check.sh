ulimit -c 0
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/prof_decode.py 4 2>&1 | tail -6
