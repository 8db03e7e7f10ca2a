#!/bin/bash
# scratch GPU job (edited per experiment)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_parity_stage_general.py -x -q -m gpu -k "fixup or finish" 2>&1 | grep -E "passed|failed|Error|error|assert|^E " | tail -25 | tee gpurun_out/r03q_tests.txt
