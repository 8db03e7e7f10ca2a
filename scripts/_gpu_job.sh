#!/bin/bash
# scratch GPU job (edited per experiment)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_multilevel.py tests/test_adaptive.py tests/test_parity_disk.py tests/test_driver_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -15 | tee gpurun_out/r03n_tests.txt
timeout 300 python3 scripts/smr_timing.py 10 sph problem/polytropic_index=1.40 gas/de_switch=1e-2 | tail -1
ARTEMIS_NO_ML_FUSED=1 timeout 300 python3 scripts/smr_timing.py 10 sph problem/polytropic_index=1.40 gas/de_switch=1e-2 | tail -1
timeout 300 python3 scripts/smr_timing.py 10 | tail -1
