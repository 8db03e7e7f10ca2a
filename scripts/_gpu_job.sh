#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
ARTEMIS_POISON=1 timeout 1500 python -m pytest tests/test_adaptive.py tests/test_multilevel.py -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|AssertionError:" | tail -12 | cut -c1-250
ARTEMIS_POISON=1 timeout 1500 python3 scripts/determinism_check.py blast_amr 2 120 2>&1 | tail -3 | cut -c1-200
