set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_multilevel.py -m gpu -x -q --durations=10 > gpurun_out/r02c_tests.log 2>&1; echo "tests rc=$?"
tail -30 gpurun_out/r02c_tests.log
cat /sys/fs/cgroup/cpu.max; nproc
timeout 600 python scripts/cpu_scaling.py 128 > gpurun_out/r02c_cpu_scaling.txt 2>&1; cat gpurun_out/r02c_cpu_scaling.txt
