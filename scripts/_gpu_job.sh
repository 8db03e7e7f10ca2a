cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_fused.py tests/test_parity_stage_general.py tests/test_driver_gpu.py -m gpu -x -q -k "fused or stage or sedov or overlap or rccl or curvilinear" 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --no-dropin --steps 100 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('256', d['value'])"
timeout 300 python bench.py --no-cpu-baseline --no-dropin --steps 100 --n 192 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('192', d['value'])"
ARTEMIS_FUSED_KCHUNK=16 timeout 300 python bench.py --no-cpu-baseline --no-dropin --steps 100 --n 192 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('192 fixed 16', d['value'])"
timeout 300 python bench.py --no-cpu-baseline --no-dropin --steps 100 --n 320 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('320', d['value'])"
ARTEMIS_FUSED_KCHUNK=16 timeout 300 python bench.py --no-cpu-baseline --no-dropin --steps 100 --n 320 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('320 fixed 16', d['value'])"
for w in blast_sph disk_sph; do timeout 300 python scripts/curv_timing.py $w; done
