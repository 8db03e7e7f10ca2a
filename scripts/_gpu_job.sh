cd /root/repo
timeout 600 python -m pytest tests/test_adaptive.py -q -m gpu -k "distance_table" > gpurun_out/t.log 2>&1; grep -E "passed|failed|^E  |Error" gpurun_out/t.log | head -8
