set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -m gpu -q --durations=15 -x > gpurun_out/r02h_suite.log 2>&1; echo "suite rc=$?"
tail -30 gpurun_out/r02h_suite.log
for n in 1024 4096; do
timeout 300 python bench.py --workload ssheet_dust --n $n --no-cpu-baseline --steps 50 2>/dev/null > gpurun_out/r02h_cfg3_$n.json; cat gpurun_out/r02h_cfg3_$n.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 $n', '%.4e'%d['value'], d['roofline']['frac'], d['roofline']['launch_ms'])"
done
timeout 300 python bench.py --workload ssheet_dust --n 4096 --dust 2 --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 4096 dust2', '%.4e'%d['value'], d['roofline']['frac'], d['roofline']['launch_ms'])"
