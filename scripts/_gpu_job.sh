#!/bin/bash
# scratch GPU job (edited per experiment): the plain GPU suite
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED" | tail -8
