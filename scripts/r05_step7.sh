tag=${1:-a6}
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_sources.py tests/test_parity_stage_general.py tests/test_nbody.py -m gpu -x -q 2>&1 | tail -3
timeout 2400 python -m pytest tests/test_driver_gpu.py tests/test_config0_linwave1d.py tests/test_parity_disk.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 2400 python -m pytest tests/test_adaptive.py -m gpu -x -q -k "not bench_size" 2>&1 | grep -E "passed|failed|Error" | tail -3
for w in disk_sph sedov3d; do
timeout 900 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tee gpurun_out/${tag}_${w}_line.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
