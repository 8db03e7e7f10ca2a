set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_config0_linwave1d.py tests/test_driver_gpu.py -m gpu -x -q -k "hip_driver_equals_oracle or timeout or dropin or sync_free or blast3d_fused" > gpurun_out/r02a_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r02a_tests.log
timeout 600 python bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; echo "bench rc=$?"; cat gpurun_out/r02a_bench.json; tail -3 gpurun_out/r02a_bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02a_prof -o p --output-format csv -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/r02a_prof.log 2>&1; echo "prof rc=$?"
find gpurun_out/r02a_prof -name "*kernel_stats.csv" | head -1 | xargs head -12
timeout 900 python3 scripts/pmc_traffic.py --tag r02a; echo "pmc rc=$?"
