cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_adaptive.py -m gpu -x -q 2>&1 | tail -5
