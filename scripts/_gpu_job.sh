set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_fused.py tests/test_driver_gpu.py -m gpu -x -q -k "fused or sedov or overlap or rccl" 2>&1 | tail -5
for cfg in "sw16:" "nosw16:ARTEMIS_FUSED_NO_SWIZZLE=1" "sw32:ARTEMIS_FUSED_KCHUNK=32" "sw64:ARTEMIS_FUSED_KCHUNK=64"; do
  tag=${cfg%%:*}; ev=${cfg#*:}
  ( [ -n "$ev" ] && export $ev; timeout 300 python bench.py --no-cpu-baseline --no-dropin --steps 200 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['roofline']['launch_ms'])"
    timeout 900 python3 scripts/pmc_traffic.py --tag r02j_$tag )
done
