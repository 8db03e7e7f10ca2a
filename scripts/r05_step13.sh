cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_capi_load.py -m gpu -q 2>&1 | tail -2
timeout 1500 python bench.py --workload disk_amr --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b2_disk_amr_line.json 2> gpurun_out/b2_disk_amr.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/b2_disk_amr_line.json').read().strip().splitlines()[-1])
print('amr', d['value'], d['ms_per_step'], d['roofline']['frac'])
r=d['remesh']; print('forced ms', r['ms_mean'], r['ms_mean_split'])
b=r['batched']; print([ (e['created'], round(e['ms']), round(e['ms_build_state'])) for e in b['events']]); print('bytes/zone', b['bytes_per_zone_now'], b.get('bytes_per_zone_peak'), b['over_cycle_mean'], b['over_cycle_max'])
PY
