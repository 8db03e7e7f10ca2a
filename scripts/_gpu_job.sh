set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_driver_gpu.py -m gpu -x -q -k "rccl or timeout or dropin" > gpurun_out/r02b_tests.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r02b_tests.log
timeout 900 python bench.py > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err; echo "bench rc=$?"; cat gpurun_out/r02b_bench.json; tail -3 gpurun_out/r02b_bench.err
timeout 600 python bench.py --loopback --blocks-per-gpu 2 --no-cpu-baseline --steps 100 > gpurun_out/r02b_bench_loopback.json 2> gpurun_out/r02b_bench_loopback.err; echo "loopback rc=$?"; cat gpurun_out/r02b_bench_loopback.json; tail -3 gpurun_out/r02b_bench_loopback.err
lscpu | head -20 > gpurun_out/r02b_lscpu.txt; numactl -H >> gpurun_out/r02b_lscpu.txt 2>&1
