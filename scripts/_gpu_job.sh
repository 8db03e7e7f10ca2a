cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_stage_general.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | head -20
timeout 1500 python -m pytest tests/test_driver_gpu.py -m gpu -q -x -k "curvilinear or axisymmetric or disk or blast" 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | head -20
for w in blast_sph blast_cyl disk_sph; do timeout 300 python scripts/curv_timing.py $w; done
ARTEMIS_SETUP_TIMING=1 timeout 300 python scripts/smr_timing.py 2 2>&1 | grep -E "setup|blocks"
