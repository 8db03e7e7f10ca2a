tag=${1:-c1}
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_stage_general.py tests/test_parity_fused.py tests/test_parity_sources.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_driver_gpu.py -m gpu -x -q -k "strat or ssheet or config3 or stage2d or dust" 2>&1 | tail -3
for n in 1024 4096; do timeout 300 python bench.py --workload ssheet_dust --n $n --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$n', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])"; done
timeout 300 python bench.py --workload ssheet_dust --n 1024 --dust 2 --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2dust', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])"
