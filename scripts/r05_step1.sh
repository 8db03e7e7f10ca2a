# round-5 iteration job: curvilinear parity tests + timings + VALU counts of the disk_sph kernels
tag=${1:-s1}
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_stage_general.py tests/test_parity_fused.py tests/test_parity_disk.py tests/test_parity_geometry.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/${tag}_tests.txt; cat gpurun_out/${tag}_tests.txt
for w in disk_sph blast_sph blast_cyl disk_cyl; do timeout 300 python scripts/curv_timing.py $w; done 2>&1 | grep zc/s | tee gpurun_out/${tag}_curv.txt
timeout 300 python bench.py --workload disk_sph --no-cpu-baseline --steps 50 2>/dev/null | tee gpurun_out/${tag}_disk_sph_line.json | cut -c1-200
PMC_SQ_GROUPS=0,1 PMC_SQ_KERNELS=stage_curv,viscous_source timeout 600 python3 scripts/pmc_sq.py ${tag}_disk_sph -- bench.py --workload disk_sph --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_disk_sph_pmc_sq.txt 2>&1; tail -6 gpurun_out/${tag}_disk_sph_pmc_sq.txt
bash scripts/prof_kernels.sh ${tag}_disk_sph --workload disk_sph --no-cpu-baseline --steps 50 | head -8
