cd /root/repo
export TMPDIR=/tmp
timeout 300 python scripts/tuned2d_timing.py 4096
timeout 300 python scripts/tuned2d_timing.py 1024
timeout 300 python scripts/tuned2d_timing.py 256
