cd /root/repo
timeout 900 python -m pytest tests/test_driver_gpu.py -q -m gpu -k "long_sedov" > gpurun_out/t.log 2>&1; grep -E "passed|failed|^E  |Error" gpurun_out/t.log | head -12
