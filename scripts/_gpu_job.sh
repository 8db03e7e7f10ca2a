cd /root/repo
for sh in 0 11 27 1009 5 77; do
ARTEMIS_SEED_SHIFT=$sh timeout 900 python -m pytest tests/test_parity_fused.py tests/test_parity_stage_general.py tests/test_parity_ops.py -q -m gpu -k "vanishing" > gpurun_out/seed_$sh.log 2>&1; echo "shift $sh: $(grep -E 'passed|failed' gpurun_out/seed_$sh.log | tail -1)"; grep -E "^E  .*Assert" gpurun_out/seed_$sh.log | head -3
done
