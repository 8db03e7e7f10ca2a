cd /root/repo
export TMPDIR=/tmp
for w in blast_sph disk_sph; do timeout 300 python scripts/curv_timing.py $w; done
timeout 300 python bench.py --no-cpu-baseline --no-dropin --steps 200 | cut -c1-160
