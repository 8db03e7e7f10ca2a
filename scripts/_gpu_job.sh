cd /root/repo
for sh in 11 27 1009 424242; do
ARTEMIS_SEED_SHIFT=$sh timeout 900 python -m pytest tests/test_parity_ops.py tests/test_parity_fused.py tests/test_parity_stage_general.py tests/test_parity_diffusion.py tests/test_parity_sources.py tests/test_parity_geometry.py tests/test_parity_refine.py -q -m gpu > gpurun_out/seed_$sh.log 2>&1; echo "shift $sh: $(grep -E 'passed|failed' gpurun_out/seed_$sh.log | tail -1)"; grep -E "^FAILED" gpurun_out/seed_$sh.log | head -5
done
