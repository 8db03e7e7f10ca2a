cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_diffusion.py tests/test_parity_stage_general.py tests/test_capi_load.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | head -20
timeout 1500 python -m pytest tests/test_driver_gpu.py tests/test_multilevel.py -m gpu -q -x -k "disk or visc or diffusion or conduction or alpha or multilevel or gaussian" 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | head -20
for w in disk_sph disk_cyl; do timeout 300 python scripts/curv_timing.py $w; done
timeout 300 python scripts/smr_timing.py 20 2>&1 | grep blocks
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02q_disk_prof -o p --output-format csv -- python3 scripts/curv_timing.py disk_sph > gpurun_out/r02q_disk_prof.log 2>&1
find gpurun_out/r02q_disk_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02q_disk_sph_kernel_stats.csv; head -8 gpurun_out/r02q_disk_sph_kernel_stats.csv | cut -c1-90,150-230
