# Copy the outputs of `scripts/final_gpu_job.sh <tag>` (merged back under gpurun_out/) into profiles/<round>_*.
tag=${1:-r06z}
ROUND=${ROUND:-r06}
cd /root/repo
g=gpurun_out
for f in pmc_traffic disk_sph_pmc_traffic cfg3_pmc_traffic cfg3_1024_pmc_traffic cfg3_1024_2dust_pmc_traffic disk_sph_smr_pmc_traffic disk_amr_pmc_traffic pmc_sq disk_sph_pmc_sq; do cp $g/${tag}_$f.json profiles/${ROUND}_$f.json; done
for f in bench_line cfg3_line cfg3_1024_line cfg3_1024_2dust_line disk_sph_line disk_sph_smr_line disk_amr_line disk_amr_block32_line disk_amr_remesh_in_timed_region_line disk_sph_smr_loopback_line disk_amr_loopback_line linwave3d_ppm_line linwave3d_plm_line; do cp $g/${tag}_$f.json profiles/${ROUND}_$f.json; done
cp $g/${tag}_identity.txt profiles/${ROUND}_identity.txt
for f in bench bench_default cfg3 cfg3_1024 disk_sph smr_cart smr_sph amr; do cp $g/${tag}_${f}_kernel_stats.csv profiles/${ROUND}_${f}_kernel_stats.csv; done
{
  echo "# scripts/final_gpu_job.sh $tag on one MI355X (gpurun), $(date -u +%Y-%m-%d) -- GPU suite, smoke, SMR timings, curvilinear timings"
  echo "## pytest -m gpu"; cat $g/${tag}_tests.txt
  echo "## __graft_entry__.smoke()"; cat $g/${tag}_smoke.txt
  echo "## scripts/smr_timing.py 20 (inputs/disk/disk_cart.in as shipped) and ... sph (refined spherical disk)"; cat $g/${tag}_smr.txt
  echo "## scripts/amr_timing.py 10 128 128 16 16 gas/refine_thr=2.0 x3 in [-0.2, 0.2] (the configs[4] combination in 3-D)"; cat $g/${tag}_amr.txt
  echo "## scripts/curv_timing.py"; cat $g/${tag}_curv.txt
} > profiles/${ROUND}_final_suite_smoke_timings.txt
python - <<PY
import json
from bench import library_identity
for f, scope in (("pmc_traffic", "fused"), ("disk_sph_pmc_traffic", "all"), ("cfg3_pmc_traffic", "all"), ("cfg3_1024_pmc_traffic", "all"), ("cfg3_1024_2dust_pmc_traffic", "all"), ("disk_sph_smr_pmc_traffic", "all"), ("disk_amr_pmc_traffic", "all")):
    rec = json.load(open("profiles/${ROUND}_%s.json" % f))
    print(f, "record identity == loaded library:", rec["library_identity"] == library_identity(scope), round(rec["hbm_bytes_per_launch"] / 1e9, 3), "GB")
PY
