cd /root/repo
export TMPDIR=/tmp
for m in 1 2; do
ARTEMIS_FORCE_OVERLAP=1 timeout 300 python bench.py --no-cpu-baseline --no-dropin --overlap-mode $m 2>&1 | cut -c60-135
done
timeout 300 python bench.py --no-cpu-baseline --no-dropin 2>&1 | cut -c60-135
ARTEMIS_FORCE_OVERLAP=1 timeout 600 python -m pytest tests/test_driver_gpu.py -x -q -m gpu -k "overlap or sedov or blast" 2>&1 | tail -2
