cd /root/repo
timeout 600 python -m pytest tests/test_parity_stage_general.py -q -m gpu -k "vanishing" > gpurun_out/t.log 2>&1; grep -E "passed|failed|^E  .*Assert|entries above|Error" gpurun_out/t.log | head -8
