set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_multilevel.py -m gpu -x -q --durations=10 -k "disk" > gpurun_out/r02d_tests.log 2>&1; echo "tests rc=$?"
tail -30 gpurun_out/r02d_tests.log
timeout 1500 python -m pytest tests/ -m gpu -q --durations=25 > gpurun_out/r02d_suite.log 2>&1; echo "suite rc=$?"
tail -45 gpurun_out/r02d_suite.log
