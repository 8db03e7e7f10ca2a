#!/bin/bash
# scratch GPU job (edited per experiment)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_nbody.py tests/test_adaptive.py tests/test_multilevel.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -15 | tee gpurun_out/r03u_tests.txt
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r03u_prof_amr -o p --output-format csv -- python3 scripts/amr_timing.py 5 128 128 16 16 gas/refine_thr=2.0 parthenon/mesh/x3min=-0.2 parthenon/mesh/x3max=0.2 > gpurun_out/r03u_amr.log 2>&1
f=$(find gpurun_out/r03u_prof_amr -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r03u_amr_kernel_stats.csv; grep nbody $f | cut -c1-200; grep zone-cyc gpurun_out/r03u_amr.log | cut -c1-250
rm -f gpurun_out/r03u_prof_amr/*kernel_trace.csv
