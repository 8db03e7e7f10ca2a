#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_diffusion.py tests/test_parity_ops.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -3
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r03x2_prof -o p --output-format csv -- python3 bench.py --workload disk_sph --no-cpu-baseline --steps 50 > gpurun_out/r03x2_disk.json 2>/dev/null
f=$(find gpurun_out/r03x2_prof -name "*kernel_stats.csv" | head -1); grep -E "viscous|stage_fused" $f | cut -c1-50,150-260; cut -c1-200 gpurun_out/r03x2_disk.json
rm -f gpurun_out/r03x2_prof/*kernel_trace.csv
timeout 300 python3 scripts/smr_timing.py 10 sph problem/polytropic_index=1.40 gas/de_switch=1e-2 | tail -1
timeout 300 python3 scripts/smr_timing.py 10 | tail -1
