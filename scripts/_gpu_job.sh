cd /root/repo
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/bc_prof3 -o p --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dropin > gpurun_out/bc_line3.json 2>/dev/null
cut -c60-135 gpurun_out/bc_line3.json
f=$(find gpurun_out/bc_prof3 -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
for r in list(csv.DictReader(open('$f')))[:4]:
    print(r['Name'][30:100], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | cut -c60-135
