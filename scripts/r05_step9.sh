cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_driver_gpu.py tests/test_config0_linwave1d.py tests/test_parity_disk.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|^FAILED" | tail -5
timeout 2400 python -m pytest tests/test_adaptive.py -m gpu -x -q -k "not bench_size" 2>&1 | grep -E "passed|failed|Error" | tail -3
