cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_parity_ops.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|^E " gpurun_out/t.log | head
timeout 1500 python -m pytest tests/test_driver_gpu.py tests/test_parity_fused.py -x -q -m gpu > gpurun_out/t2.log 2>&1; grep -E "passed|failed|^E " gpurun_out/t2.log | head
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/flux_line.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/flux_line.json').read())
print(d['value'], d['dropin'])
PY
