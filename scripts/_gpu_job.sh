set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_stage_general.py -m gpu -x -q > gpurun_out/r02g_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r02g_tests.log
ARTEMIS_STAGE2D_OCC1=1 timeout 900 python -m pytest tests/test_parity_stage_general.py tests/test_parity_sources.py -m gpu -x -q -k "hydro or sources or drag" > gpurun_out/r02g_tests_occ1.log 2>&1; echo "tests occ1 rc=$?"
tail -5 gpurun_out/r02g_tests_occ1.log
timeout 300 python -m pytest tests/test_driver_gpu.py -m gpu -x -q -k "dusty or shearing or drag_deck or advection" 2>&1 | tail -5
for v in "" "ARTEMIS_STAGE2D_OCC1=1" "ARTEMIS_NO_STAGE2D=1" "ARTEMIS_STAGE2D_OCC1=1 ARTEMIS_STAGE2D_ROWS=16"; do
  env $v timeout 300 python bench.py --workload ssheet_dust --n 4096 --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 [$v]', '%.4e'%d['value'], d['roofline']['frac'], d['roofline']['launch_ms'])"
done
