cd /root/repo
timeout 900 python -m pytest tests/test_parity_ops.py tests/test_parity_fused.py tests/test_parity_stage_general.py tests/test_parity_geometry.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|^E  .*Assert|mismatch" gpurun_out/t.log | head -8
