cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_parity_stage_general.py tests/test_parity_disk.py tests/test_multilevel.py tests/test_nbody.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|^FAILED" | tail -5
{
echo "## scripts/determinism_check.py disk_planet_dust_amr 3 30 (three runs of the 2-D configs[4] deck: path, blocks, remeshes, dt, hash of every leaf, hash of the particle forces)"
timeout 900 python scripts/determinism_check.py disk_planet_dust_amr 3 30 2>&1 | tail -4
echo "## scripts/sedov256_check.py (the headline deck to t = 0.1: conservation, shock radius against Sedov-Taylor)"
timeout 900 python scripts/sedov256_check.py 2>&1 | tail -2
} | tee gpurun_out/r05z_checks.txt
