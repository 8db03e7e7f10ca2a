tag=${1:-h1}
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_fused.py tests/test_parity_stage_general.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-dropin --no-overlap-emulation --steps 100 2>/dev/null | tee gpurun_out/${tag}_bench_line.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])"
