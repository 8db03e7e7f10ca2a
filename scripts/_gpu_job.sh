cd /root/repo
export TMPDIR=/tmp
timeout 600 python scripts/smr_timing.py 20 sph problem/polytropic_index=1.40 gas/de_switch=1e-2
