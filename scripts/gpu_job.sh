# One gpurun call of round 6's development loop: identity check, then whatever PARTS names.
#   PARTS="suite loopback lines" scripts/gpu_job.sh <tag>
PARTS=${PARTS:-"suite loopback lines"}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
tag=${1:-j}
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -c "from artemis_amd import build as b; b.build_hip(); print('library source sha', b.verify())" 2>&1 | tail -1 | tee gpurun_out/${tag}_identity.txt || exit 1
if has suite; then
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/${tag}_tests.txt; tail -3 gpurun_out/${tag}_tests.txt
fi
if has refined; then
timeout 1800 python -m pytest tests/test_multilevel.py tests/test_adaptive.py tests/test_parity_refine.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/${tag}_tests_refined.txt; tail -3 gpurun_out/${tag}_tests_refined.txt
fi
if has loopback; then
timeout 600 python bench.py --workload disk_sph_smr --loopback --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_smr_loopback_line.json 2> gpurun_out/${tag}_smr_loopback.err; tail -3 gpurun_out/${tag}_smr_loopback.err; cut -c1-300 gpurun_out/${tag}_smr_loopback_line.json
timeout 900 python bench.py --workload disk_amr --loopback --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_amr_loopback_line.json 2> gpurun_out/${tag}_amr_loopback.err; tail -3 gpurun_out/${tag}_amr_loopback.err; cut -c1-300 gpurun_out/${tag}_amr_loopback_line.json
fi
if has lines; then
timeout 600 python bench.py --no-cpu-baseline --steps 100 2>/dev/null > gpurun_out/${tag}_bench_line.json; cut -c1-200 gpurun_out/${tag}_bench_line.json
timeout 300 python bench.py --workload ssheet_dust --n 1024 --no-cpu-baseline --steps 100 2>/dev/null > gpurun_out/${tag}_cfg3_1024_line.json; cut -c1-200 gpurun_out/${tag}_cfg3_1024_line.json
timeout 300 python bench.py --workload disk_sph --no-cpu-baseline --steps 50 2>/dev/null > gpurun_out/${tag}_disk_sph_line.json; cut -c1-200 gpurun_out/${tag}_disk_sph_line.json
timeout 600 python bench.py --workload disk_sph_smr --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null > gpurun_out/${tag}_disk_sph_smr_line.json; cut -c1-200 gpurun_out/${tag}_disk_sph_smr_line.json
timeout 900 python bench.py --workload disk_amr --no-cpu-baseline --no-remesh-leg --steps 20 --warmup 5 2>/dev/null > gpurun_out/${tag}_disk_amr_line.json; cut -c1-200 gpurun_out/${tag}_disk_amr_line.json
fi
if has amrprof; then
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_amr_prof -o p --output-format csv -- python3 scripts/amr_timing.py 5 128 128 16 16 gas/refine_thr=2.0 parthenon/mesh/x3min=-0.2 parthenon/mesh/x3max=0.2 > gpurun_out/${tag}_amr_prof.log 2>&1
find gpurun_out/${tag}_amr_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_amr_kernel_stats.csv
rm -rf gpurun_out/${tag}_amr_prof
tail -2 gpurun_out/${tag}_amr_prof.log
fi
if has drag; then
timeout 1800 python -m pytest tests -m gpu -q -x -k "drag or dust or disk or nbody or config4 or cfg4" 2>&1 | tail -15 > gpurun_out/${tag}_tests_drag.txt; tail -3 gpurun_out/${tag}_tests_drag.txt
fi
if has amrline; then
timeout 900 python bench.py --workload disk_amr --no-cpu-baseline --no-remesh-leg --steps 20 --warmup 5 2>/dev/null > gpurun_out/${tag}_disk_amr_line.json; cut -c1-200 gpurun_out/${tag}_disk_amr_line.json
fi
if has cfg3; then
timeout 1200 python -m pytest tests/test_driver_gpu.py tests/test_parity_stage_general.py -m gpu -q -x -k "strat or sheet or stage2d or row_march or 2d or dusty" 2>&1 | tail -8 > gpurun_out/${tag}_tests_cfg3.txt; tail -3 gpurun_out/${tag}_tests_cfg3.txt
for n in 1024 4096; do timeout 300 python bench.py --workload ssheet_dust --n $n --no-cpu-baseline --steps 100 2>/dev/null > gpurun_out/${tag}_cfg3_${n}_line.json; cut -c1-220 gpurun_out/${tag}_cfg3_${n}_line.json; done
timeout 300 python bench.py --workload ssheet_dust --n 1024 --dust 2 --no-cpu-baseline --steps 100 2>/dev/null > gpurun_out/${tag}_cfg3_1024_2dust_line.json; cut -c1-220 gpurun_out/${tag}_cfg3_1024_2dust_line.json
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_cfg3_prof -o p --output-format csv -- python3 bench.py --workload ssheet_dust --n 1024 --no-cpu-baseline --steps 100 > gpurun_out/${tag}_cfg3_prof.log 2>&1
find gpurun_out/${tag}_cfg3_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_cfg3_1024_kernel_stats.csv
rm -rf gpurun_out/${tag}_cfg3_prof
fi
if has cart; then
timeout 2400 python -m pytest tests -m gpu -q -x -k "cart or Cart or multilevel or stage_general or gravity or visc or diffusion or blast" 2>&1 | tail -8 > gpurun_out/${tag}_tests_cart.txt; tail -3 gpurun_out/${tag}_tests_cart.txt
timeout 300 python scripts/smr_timing.py 20 | tee gpurun_out/${tag}_smr.txt
timeout 300 python scripts/smr_timing.py 20 sph problem/polytropic_index=1.40 gas/de_switch=1e-2 | tee -a gpurun_out/${tag}_smr.txt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_smrc_prof -o p --output-format csv -- python3 scripts/smr_timing.py 10 > gpurun_out/${tag}_smrc_prof.log 2>&1
find gpurun_out/${tag}_smrc_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_smr_cart_kernel_stats.csv
rm -rf gpurun_out/${tag}_smrc_prof
fi
if has ppm; then
for r in ppm plm; do timeout 300 python bench.py --workload linwave3d --recon $r --steps 30 --warmup 5 2>gpurun_out/${tag}_linwave_$r.err > gpurun_out/${tag}_linwave_${r}_line.json; cut -c1-200 gpurun_out/${tag}_linwave_${r}_line.json; tail -2 gpurun_out/${tag}_linwave_$r.err; done
fi
if has disk; then
timeout 2400 python -m pytest tests -m gpu -q -k "disk or multilevel or adaptive or bc or boundary or ic or refine" 2>&1 | tail -8 > gpurun_out/${tag}_tests_disk.txt; tail -3 gpurun_out/${tag}_tests_disk.txt
timeout 300 python bench.py --workload disk_sph --no-cpu-baseline --steps 50 2>/dev/null > gpurun_out/${tag}_disk_sph_line.json; cut -c1-200 gpurun_out/${tag}_disk_sph_line.json
timeout 600 python bench.py --workload disk_sph_smr --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null > gpurun_out/${tag}_disk_sph_smr_line.json; cut -c1-200 gpurun_out/${tag}_disk_sph_smr_line.json
timeout 900 python bench.py --workload disk_amr --no-cpu-baseline --no-remesh-leg --steps 20 --warmup 5 2>/dev/null > gpurun_out/${tag}_disk_amr_line.json; cut -c1-200 gpurun_out/${tag}_disk_amr_line.json
timeout 900 python bench.py --workload disk_amr --amr-block 32 --no-cpu-baseline --no-remesh-leg --steps 20 --warmup 5 2>/dev/null > gpurun_out/${tag}_disk_amr32_line.json; cut -c1-200 gpurun_out/${tag}_disk_amr32_line.json
fi
if has remesh; then
timeout 600 python scripts/remesh_profile.py > gpurun_out/${tag}_remesh_profile.txt 2>&1; tail -80 gpurun_out/${tag}_remesh_profile.txt
fi
if has sphprof; then
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_sph_prof -o p --output-format csv -- python3 bench.py --workload disk_sph --no-cpu-baseline --steps 50 > gpurun_out/${tag}_sph_prof.log 2>&1
find gpurun_out/${tag}_sph_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_disk_sph_kernel_stats.csv
rm -rf gpurun_out/${tag}_sph_prof
fi
if has headline; then
timeout 2400 python -m pytest tests/test_parity_fused.py tests/test_driver_gpu.py tests/test_multilevel.py -m gpu -q 2>&1 | tail -6 > gpurun_out/${tag}_tests_fused.txt; grep -E "passed|failed" gpurun_out/${tag}_tests_fused.txt
timeout 600 python bench.py --no-cpu-baseline --steps 100 2>/dev/null > gpurun_out/${tag}_bench_line.json; python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_bench_line.json')); print('headline', d['value'], d['roofline']['frac'], 'emulation', d['config'].get('overlap_emulation_zcps'), 'dropin', {k: v for k, v in d.get('dropin', {}).items() if k.startswith('fused') or k == 'per_task'})"
fi
if has amrfull; then
timeout 2400 python -m pytest tests/test_adaptive.py tests/test_multilevel.py -m gpu -q 2>&1 | tail -6 > gpurun_out/${tag}_tests_amr.txt; grep -E "passed|failed" gpurun_out/${tag}_tests_amr.txt
timeout 1500 python bench.py --workload disk_amr --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null > gpurun_out/${tag}_disk_amr_full_line.json; python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_disk_amr_full_line.json')); r=d['remesh']; b=r['batched']; print('value', d['value'], 'frac', d['roofline']['frac'], 'single ms', r['ms_mean'], 'over cycle', r['remesh_over_cycle'], 'batched ms mean/max', b['ms_mean'], b['ms_max'], 'over cycle', b['over_cycle_mean'], 'B/zone now/peak', b['bytes_per_zone_now'], b['bytes_per_zone_peak'], 'build share', b['build_state_share'])"
timeout 1500 python bench.py --workload disk_amr --steps 24 --warmup 5 --no-cpu-baseline --no-remesh-leg --remesh-in-timed-region 2>/dev/null > gpurun_out/${tag}_disk_amr_remesh_in_timed_region_line.json; python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_disk_amr_remesh_in_timed_region_line.json')); r=d['remesh_in_timed_region']; print('with remeshes in the timed region', d['value'], 'cycle ms', r['cycle_ms_with_them'], 'events', [(e['created'], e['destroyed'], round(e['ms'],1)) for e in r['events']])"
fi
