set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | head -20
timeout 300 python scripts/smr_timing.py 20
timeout 300 python scripts/curv_timing.py disk_sph
timeout 300 python bench.py --no-cpu-baseline --steps 100 | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02o_smr_prof -o p --output-format csv -- python3 scripts/smr_timing.py 10 > gpurun_out/r02o_smr_prof.log 2>&1
find gpurun_out/r02o_smr_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02o_smr_kernel_stats.csv; head -14 gpurun_out/r02o_smr_kernel_stats.csv | cut -c1-70,140-260
