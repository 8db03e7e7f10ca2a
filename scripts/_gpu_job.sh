cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_parity_ops.py tests/test_parity_fused.py tests/test_parity_geometry.py tests/test_multilevel.py tests/test_config0_linwave1d.py tests/test_parity_sources.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed|^E " gpurun_out/t.log | head
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/bc_prof -o p --output-format csv -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-dropin > gpurun_out/bc_line.json 2>/dev/null
cut -c60-130 gpurun_out/bc_line.json
f=$(find gpurun_out/bc_prof -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
for r in list(csv.DictReader(open('$f')))[:6]:
    print(r['Name'][30:100], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
timeout 300 python scripts/curv_timing.py disk_sph
timeout 300 python scripts/smr_timing.py 20 | tail -1
