cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_driver_gpu.py tests/test_capi_load.py tests/test_parity_ops.py -m gpu -x -q -k "dropin or capi or prim_to_cons or abi" 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value']); print({k:v for k,v in d['dropin'].items() if k!='note'})"
