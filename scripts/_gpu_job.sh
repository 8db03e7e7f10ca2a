# scratch: the command of the most recent ad-hoc `gpurun -- 'bash scripts/_gpu_job.sh'` call (see scripts/final_gpu_job.sh
# for the evidence run)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
line() { python3 -c "
import json,sys;d=json.load(open('$1'));print('$2',d['value'],d['roofline']['launch_ms'],d['roofline']['frac'])"; }
for v in v0 v2; do
  cp build_variants/lib_$v.so artemis_amd/lib/libartemis_hip.so
  ARTEMIS_NO_REDO=1 timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-dropin > gpurun_out/r03d_${v}_noredo.json 2>/dev/null; line gpurun_out/r03d_${v}_noredo.json ${v}_noredo
  timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-dropin > gpurun_out/r03d_${v}_redo.json 2>/dev/null; line gpurun_out/r03d_${v}_redo.json ${v}_redo
done
for v in v0 v2; do
  cp build_variants/lib_$v.so artemis_amd/lib/libartemis_hip.so
  ARTEMIS_NO_REDO=1 timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-dropin > gpurun_out/r03d_${v}_noredo2.json 2>/dev/null; line gpurun_out/r03d_${v}_noredo2.json ${v}_noredo_again
done
cp build_variants/lib_v2.so artemis_amd/lib/libartemis_hip.so
timeout 1500 python -m pytest tests/test_parity_fused.py tests/test_parity_stage_general.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -25 | tee gpurun_out/r03d_tests.txt
