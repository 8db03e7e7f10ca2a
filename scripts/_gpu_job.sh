cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | head -20
