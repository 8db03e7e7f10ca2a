cd /root/repo
export TMPDIR=/tmp
tag=r02z
timeout 600 python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err; cut -c1-200 gpurun_out/${tag}_bench_line.json
timeout 300 python bench.py --workload ssheet_dust --n 4096 --no-cpu-baseline --steps 50 2>/dev/null > gpurun_out/${tag}_cfg3_line.json
timeout 300 python bench.py --workload disk_sph --no-cpu-baseline --steps 50 2>/dev/null > gpurun_out/${tag}_disk_sph_line.json
cut -c1-200 gpurun_out/${tag}_cfg3_line.json gpurun_out/${tag}_disk_sph_line.json
