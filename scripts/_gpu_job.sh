cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_multilevel.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | head -20
