cd /root/repo
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/dt_prof4 -o p --output-format csv -- python3 bench.py --workload disk_sph --no-cpu-baseline --steps 50 > gpurun_out/dt_line4.json 2>/dev/null
cut -c70-130 gpurun_out/dt_line4.json
f=$(find gpurun_out/dt_prof4 -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
for r in list(csv.DictReader(open('$f')))[:8]:
    if 'dt_kernel' in r['Name']: print(r['Name'][40:100], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/dt_prof5 -o p --output-format csv -- python3 bench.py --no-cpu-baseline --steps 40 > gpurun_out/dt_line5.json 2>/dev/null
f=$(find gpurun_out/dt_prof5 -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv, json
for r in list(csv.DictReader(open('$f')))[:14]:
    if 'dt_kernel' in r['Name']: print(r['Name'][40:100], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
print(json.loads(open('gpurun_out/dt_line5.json').read())['dropin']['per_task'])
PY
timeout 600 python -m pytest tests/test_parity_diffusion.py tests/test_parity_ops.py -x -q -m gpu 2>&1 | tail -1
