#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|AssertionError:" | tail -12 | cut -c1-250 | tee gpurun_out/r03w_tests.txt
