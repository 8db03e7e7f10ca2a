cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_multilevel.py tests/test_adaptive.py tests/test_parity_disk.py tests/test_driver_gpu.py -x -q -m gpu > gpurun_out/drv_tests.log 2>&1
grep -E "passed|failed|^E " gpurun_out/drv_tests.log | head -10
