#!/bin/bash
# scratch GPU job (edited per experiment)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py --workload disk_sph_smr --steps 20 --warmup 3 > gpurun_out/r03r_smr_line.json 2> gpurun_out/r03r_smr.err; cut -c1-1500 gpurun_out/r03r_smr_line.json; tail -3 gpurun_out/r03r_smr.err
timeout 900 python bench.py --workload disk_amr --steps 10 --warmup 3 > gpurun_out/r03r_amr_line.json 2> gpurun_out/r03r_amr.err; cut -c1-1800 gpurun_out/r03r_amr_line.json; tail -3 gpurun_out/r03r_amr.err
timeout 900 python3 scripts/amr_timing.py 10 128 128 32 32 gas/refine_thr=2.0 parthenon/mesh/x3min=-0.2 parthenon/mesh/x3max=0.2 2>&1 | tail -1 | cut -c1-300
