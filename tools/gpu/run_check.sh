ulimit -c 0
for i in 1 2 3; do timeout 900 python -m pytest tests/test_post_stages.py -x -q 2>&1 | tail -2; done
