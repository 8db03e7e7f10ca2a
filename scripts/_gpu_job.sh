# scratch: the command of the most recent ad-hoc `gpurun -- 'bash scripts/_gpu_job.sh'` call (see scripts/final_gpu_job.sh
# for the evidence run)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
line() { python3 -c "
import json,sys;d=json.load(open('$1'));print('$2',d['value'],d['roofline']['launch_ms'],d['roofline']['frac'])"; }
timeout 300 python bench.py --workload ssheet_dust --n 4096 --no-cpu-baseline --steps 50 > gpurun_out/r03h_cfg3_4096.json 2>/dev/null; line gpurun_out/r03h_cfg3_4096.json cfg3_4096
ARTEMIS_NO_REDO=1 timeout 300 python bench.py --workload ssheet_dust --n 4096 --no-cpu-baseline --steps 50 > gpurun_out/r03h_cfg3_4096_noredo.json 2>/dev/null; line gpurun_out/r03h_cfg3_4096_noredo.json cfg3_4096_noredo
timeout 300 python bench.py --workload ssheet_dust --n 1024 --no-cpu-baseline --steps 100 > gpurun_out/r03h_cfg3_1024.json 2>/dev/null; line gpurun_out/r03h_cfg3_1024.json cfg3_1024
ARTEMIS_NO_REDO=1 timeout 300 python bench.py --workload ssheet_dust --n 1024 --no-cpu-baseline --steps 100 > gpurun_out/r03h_cfg3_1024_noredo.json 2>/dev/null; line gpurun_out/r03h_cfg3_1024_noredo.json cfg3_1024_noredo
timeout 300 python bench.py --workload ssheet_dust --n 1024 --dust 2 --no-cpu-baseline --steps 100 > gpurun_out/r03h_cfg3_1024_d2.json 2>/dev/null; line gpurun_out/r03h_cfg3_1024_d2.json cfg3_1024_2dust
timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-dropin > gpurun_out/r03h_hint.json 2>/dev/null; line gpurun_out/r03h_hint.json headline
timeout 1500 python -m pytest tests/test_parity_fused.py tests/test_parity_stage_general.py tests/test_driver_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -25 | tee gpurun_out/r03h_tests.txt
