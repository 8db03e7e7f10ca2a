cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_driver_gpu.py tests/test_parity_fused.py tests/test_parity_ops.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|^E  " gpurun_out/t.log | head -6
timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | cut -c60-135
ARTEMIS_NO_X1_GHOSTS=1 timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | cut -c60-135
timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | cut -c60-135
