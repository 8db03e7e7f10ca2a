# scratch: the command of the most recent ad-hoc `gpurun -- 'bash scripts/_gpu_job.sh'` call (see scripts/final_gpu_job.sh
# for the evidence run)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
line() { python3 -c "
import json,sys;d=json.load(open('$1'));print('$2',d['value'],d['roofline']['launch_ms'],d['roofline']['frac'])"; }
timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-dropin > gpurun_out/r03f_hint.json 2>/dev/null; line gpurun_out/r03f_hint.json hint
ARTEMIS_NO_TINY_HINT=1 timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-dropin > gpurun_out/r03f_nohint.json 2>/dev/null; line gpurun_out/r03f_nohint.json nohint
ARTEMIS_NO_REDO=1 timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-dropin > gpurun_out/r03f_noredo.json 2>/dev/null; line gpurun_out/r03f_noredo.json noredo
timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-dropin > gpurun_out/r03f_hint2.json 2>/dev/null; line gpurun_out/r03f_hint2.json hint_again
timeout 1500 python -m pytest tests/test_parity_fused.py tests/test_parity_ops.py tests/test_parity_stage_general.py tests/test_parity_geometry.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -25 | tee gpurun_out/r03f_tests.txt
