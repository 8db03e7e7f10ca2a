cd /root/repo
export TMPDIR=/tmp
PMC_SQ_KERNELS=viscous timeout 900 python3 scripts/pmc_sq.py visc -- bench.py --workload disk_sph --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -8
