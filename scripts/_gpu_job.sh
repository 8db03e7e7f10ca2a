cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
ARTEMIS_SETUP_TIMING=1 timeout 300 python scripts/smr_timing.py 5 2>&1 | grep -E "setup|blocks"
timeout 900 python -m pytest tests/test_multilevel.py tests/test_driver_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | head
timeout 900 python3 scripts/pmc_traffic.py --tag r02p --workload disk_sph
timeout 900 python3 scripts/pmc_traffic.py --tag r02p --workload ssheet_dust
timeout 900 python3 scripts/pmc_traffic.py --tag r02p
