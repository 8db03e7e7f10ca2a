tag=${1:-a7}
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_parity_diffusion.py tests/test_parity_ops.py tests/test_driver_gpu.py tests/test_multilevel.py tests/test_nbody.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 2400 python -m pytest tests/test_adaptive.py -m gpu -x -q -k "not bench_size" 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 900 python bench.py --workload disk_amr --steps 20 --warmup 5 --no-cpu-baseline --no-remesh-leg 2>/dev/null | tee gpurun_out/${tag}_disk_amr_line.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('amr', d['value'], d['ms_per_step'], d['roofline']['frac'])"
bash scripts/prof_any.sh ${tag}_amr scripts/amr_timing.py 5 128 128 16 16 gas/refine_thr=2.0 parthenon/mesh/x3min=-0.2 parthenon/mesh/x3max=0.2 | head -12
