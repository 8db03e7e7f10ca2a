cd /root/repo
timeout 300 python scripts/curv_timing.py disk_sph | cut -c1-70
ARTEMIS_FUSED_KCHUNK=64 timeout 300 python scripts/curv_timing.py disk_sph | cut -c1-70
ARTEMIS_FUSED_KCHUNK=16 timeout 300 python scripts/curv_timing.py disk_sph | cut -c1-70
timeout 300 python scripts/curv_timing.py disk_sph | cut -c1-70
