tag=${1:-a1}
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py --workload disk_sph_smr --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tee gpurun_out/${tag}_disk_sph_smr_line.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('smr', d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout 900 python bench.py --workload disk_amr --steps 20 --warmup 5 --no-cpu-baseline --no-remesh-leg 2>/dev/null | tee gpurun_out/${tag}_disk_amr_line.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('amr', d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout 300 python scripts/smr_timing.py 20 | tail -2
bash scripts/prof_any.sh ${tag}_amr scripts/amr_timing.py 5 128 128 16 16 gas/refine_thr=2.0 parthenon/mesh/x3min=-0.2 parthenon/mesh/x3max=0.2 | head -18
