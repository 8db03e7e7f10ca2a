cd /root/repo
export TMPDIR=/tmp
timeout 900 python3 scripts/pmc_traffic.py --tag r02z
