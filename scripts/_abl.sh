(timeout 600 python -m pytest tests/test_parity_diffusion.py -m gpu -x -q 2>&1 | tail -3)
for a in 0 2 15; do echo "ABL $a"; ARTEMIS_VS_ABL=$a scripts/prof_kernels.sh r4abl --workload disk_sph --n 256 --steps 6 --warmup 2 --no-cpu-baseline | grep viscous_source | cut -c100-160; done
