cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_stage_general.py tests/test_driver_gpu.py -m gpu -x -q -k "stage or sheet or dust or strat" 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python bench.py --workload ssheet_dust --n 4096 --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['launch_ms'])"; done
timeout 300 python scripts/tuned2d_timing.py 4096
timeout 300 python scripts/cfg3_timing.py 1 1024
