cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
for c in 0 8 16 32 64; do for v in 0 8 16 32; do
ARTEMIS_CURV_KCHUNK=$c ARTEMIS_VISC_KCHUNK=$v timeout 300 python bench.py --workload disk_sph --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('curv_kchunk $c visc_kchunk $v', '%.4g' % d['value'], d['ms_per_step'])"
done; done
