cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
for x in 0 1 0 1; do
for w in disk_sph; do
if [ $x = 1 ]; then export ARTEMIS_XNARROW=1; else unset ARTEMIS_XNARROW; fi
timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('xnarrow $x $w', '%.4g' % d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
done; done
for x in 0 1; do
if [ $x = 1 ]; then export ARTEMIS_XNARROW=1; else unset ARTEMIS_XNARROW; fi
for w in blast_sph disk_cyl; do timeout 300 python scripts/curv_timing.py $w | sed "s/^/xnarrow $x /"; done
done
unset ARTEMIS_XNARROW
timeout 900 python bench.py --workload disk_amr --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['remesh']; b=r['batched']
print('amr', d['value'], 'forced', r['ms_mean'], r['ms_mean_split']); print([(e['created'], round(e['ms']), round(e['ms_build_state'])) for e in b['events']], b['over_cycle_mean'], b['over_cycle_max'])"
