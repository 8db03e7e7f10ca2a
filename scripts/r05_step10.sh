cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_parity_disk.py tests/test_multilevel.py tests/test_parity_ops.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|^FAILED" | tail -5
timeout 2400 python -m pytest tests/test_driver_gpu.py -m gpu -q -k "disk or binary or alpha" 2>&1 | grep -E "passed|failed|Error|^FAILED" | tail -5
timeout 900 python bench.py --workload disk_sph --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('disk_sph', d['value'], d['ms_per_step'], d['roofline']['frac'])"
