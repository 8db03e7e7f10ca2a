# Round-end evidence run (one gpurun call): GPU suite, smoke, bench lines, rocprof stats, PMC traffic records.
# Outputs under gpurun_out/<tag>_*; scripts/collect_evidence.sh copies what is to be judged into profiles/<round>_*.
# PARTS selects what runs (suite ~6 minutes; pmc ~5; sq ~1; bench ~6; prof ~3; timings ~1 -- the PMC passes took 100 minutes
# as long as every profiled process recompiled the library, artemis_amd/build.py::_tool_env); PMC_LIST / BENCH_LIST /
# PROF_LIST select within a part.  ROUND names the profiles/ prefix the bench lines look their records up under.
PARTS=${PARTS:-"suite pmc sq bench prof timings"}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
tag=${1:-r06z}
ROUND=${ROUND:-r06}
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
# the binary is the tree: rebuild whatever differs from the sources that travelled (a no-op when nothing does), verify
# the identities compiled into the library, and put them into the evidence
python3 -c "from artemis_amd import build as b; b.build_hip(force=bool(int('${FORCE_BUILD:-0}'))); print('library source sha', b.verify())" | tee gpurun_out/${tag}_identity.txt || exit 1
if has suite; then
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error|FAILED" | tail -5 > gpurun_out/${tag}_tests.txt; cat gpurun_out/${tag}_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${tag}_smoke.txt
fi
if has pmc; then
# ---- HBM traffic (PMC; separate FETCH_SIZE / WRITE_SIZE passes).  The Sedov record is a mean over full-size launches of the
# two headline instantiations only (scripts/pmc_traffic.py asserts the launch count; bench.py refuses a record without it)
# (PMC_LIST selects a subset: every record is two profiled runs, 3-5 minutes)
wantm() { case " ${PMC_LIST:-sedov disk_sph cfg3_1024 cfg3 cfg3_2dust smr amr} " in *" $1 "*) return 0;; *) return 1;; esac; }
wantm sedov && timeout 1200 python3 scripts/pmc_traffic.py --tag ${tag}
wantm disk_sph && timeout 900 python3 scripts/pmc_traffic.py --tag ${tag} --workload disk_sph
wantm cfg3_1024 && timeout 900 python3 scripts/pmc_traffic.py --tag ${tag} --workload ssheet_dust --n 1024
wantm cfg3 && timeout 900 python3 scripts/pmc_traffic.py --tag ${tag} --workload ssheet_dust --n 4096
wantm cfg3_2dust && timeout 900 python3 scripts/pmc_traffic.py --tag ${tag} --workload ssheet_dust --n 1024 --out gpurun_out/${tag}_cfg3_1024_2dust_pmc_traffic.json --dust 2
wantm smr && timeout 1200 python3 scripts/pmc_traffic.py --tag ${tag} --workload disk_sph_smr
wantm amr && timeout 1800 python3 scripts/pmc_traffic.py --tag ${tag} --workload disk_amr
fi
if has pmcsmr; then
timeout 1700 python3 scripts/pmc_traffic.py --tag ${tag} --workload disk_sph_smr
fi
if has sq; then
# ---- SQ counters: the headline kernel alone (no drop-in legs, no overlap emulation: 256^3 launches only), the disk march
PMC_SQ_GROUPS=0,1 PMC_SQ_RECORD=${tag} timeout 600 python3 scripts/pmc_sq.py ${tag} -- bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-dropin --no-overlap-emulation > gpurun_out/${tag}_pmc_sq.txt 2>&1
PMC_SQ_GROUPS=0,1 PMC_SQ_KERNELS=stage_curv,viscous_source timeout 600 python3 scripts/pmc_sq.py ${tag}_disk_sph -- bench.py --workload disk_sph --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_disk_sph_pmc_sq.txt 2>&1
cp gpurun_out/sq_${tag}_disk_sph.json gpurun_out/${tag}_disk_sph_pmc_sq.json
fi
# (on the box: the bench lines below quote the records measured by this or an earlier call)
for f in pmc_traffic disk_sph_pmc_traffic cfg3_pmc_traffic cfg3_1024_pmc_traffic cfg3_1024_2dust_pmc_traffic disk_sph_smr_pmc_traffic disk_amr_pmc_traffic pmc_sq; do cp gpurun_out/${tag}_$f.json profiles/${ROUND}_$f.json 2>/dev/null; done
if has bench; then
# ---- bench lines
# (BENCH_LIST selects a subset)
wantb() { case " ${BENCH_LIST:-sedov cfg3 cfg3_1024 cfg3_2dust disk_sph smr amr amr32 amr_remesh smr_loop amr_loop linwave} " in *" $1 "*) return 0;; *) return 1;; esac; }
wantb sedov && timeout 900 python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err; cat gpurun_out/${tag}_bench_line.json | cut -c1-400
wantb cfg3 && timeout 300 python bench.py --workload ssheet_dust --n 4096 --no-cpu-baseline --steps 50 2>/dev/null > gpurun_out/${tag}_cfg3_line.json
wantb cfg3_1024 && timeout 300 python bench.py --workload ssheet_dust --n 1024 --steps 100 2>/dev/null > gpurun_out/${tag}_cfg3_1024_line.json
wantb cfg3_2dust && timeout 300 python bench.py --workload ssheet_dust --n 1024 --dust 2 --no-cpu-baseline --steps 100 2>/dev/null > gpurun_out/${tag}_cfg3_1024_2dust_line.json
wantb disk_sph && timeout 300 python bench.py --workload disk_sph --no-cpu-baseline --steps 50 2>/dev/null > gpurun_out/${tag}_disk_sph_line.json
wantb smr && timeout 600 python bench.py --workload disk_sph_smr --steps 40 --warmup 5 2>/dev/null > gpurun_out/${tag}_disk_sph_smr_line.json
wantb amr && timeout 1500 python bench.py --workload disk_amr --steps 20 --warmup 5 2>/dev/null > gpurun_out/${tag}_disk_amr_line.json
# ... the deck's own 32^3 blocks, the mesh changing inside the timed region, the N-rank legs through RCCL on one device, the PPM gap
wantb amr32 && timeout 900 python bench.py --workload disk_amr --amr-block 32 --steps 20 --warmup 5 --no-cpu-baseline --no-remesh-leg 2>/dev/null > gpurun_out/${tag}_disk_amr_block32_line.json
wantb amr_remesh && timeout 1500 python bench.py --workload disk_amr --steps 24 --warmup 5 --no-cpu-baseline --no-remesh-leg --remesh-in-timed-region 2>/dev/null > gpurun_out/${tag}_disk_amr_remesh_in_timed_region_line.json
wantb smr_loop && timeout 600 python bench.py --workload disk_sph_smr --loopback --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/${tag}_disk_sph_smr_loopback_line.json
wantb amr_loop && timeout 900 python bench.py --workload disk_amr --loopback --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/${tag}_disk_amr_loopback_line.json
wantb linwave && for r in ppm plm; do timeout 300 python bench.py --workload linwave3d --recon $r --steps 30 --warmup 5 2>/dev/null > gpurun_out/${tag}_linwave3d_${r}_line.json; done
cut -c1-300 gpurun_out/${tag}_cfg3_line.json gpurun_out/${tag}_cfg3_1024_line.json gpurun_out/${tag}_disk_sph_line.json gpurun_out/${tag}_disk_sph_smr_line.json gpurun_out/${tag}_disk_amr_line.json 2>/dev/null
fi
if has prof; then
# ---- rocprofv3 kernel statistics.  The headline profile holds 256^3 launches of the two headline instantiations ONLY
prof() { # prof <name> <program args ...>
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_${name}_prof -o p --output-format csv -- python3 "$@" > gpurun_out/${tag}_${name}_prof.log 2>&1
  find gpurun_out/${tag}_${name}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_${name}_kernel_stats.csv
}
# (PROF_LIST selects a subset: a gpurun call is limited to an hour)
wantp() { case " ${PROF_LIST:-bench bench_default cfg3 cfg3_1024 disk_sph smr_cart smr_sph amr} " in *" $1 "*) return 0;; *) return 1;; esac; }
wantp bench && prof bench bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-dropin --no-overlap-emulation
wantp bench_default && prof bench_default bench.py --steps 200 --warmup 10 --no-cpu-baseline
wantp cfg3 && prof cfg3 bench.py --workload ssheet_dust --n 4096 --no-cpu-baseline --steps 50
wantp cfg3_1024 && prof cfg3_1024 bench.py --workload ssheet_dust --n 1024 --no-cpu-baseline --steps 100
wantp disk_sph && prof disk_sph bench.py --workload disk_sph --no-cpu-baseline --steps 50
wantp smr_cart && prof smr_cart scripts/smr_timing.py 10
wantp smr_sph && prof smr_sph scripts/smr_timing.py 10 sph problem/polytropic_index=1.40 gas/de_switch=1e-2
wantp amr && prof amr scripts/amr_timing.py 5 128 128 16 16 gas/refine_thr=2.0 parthenon/mesh/x3min=-0.2 parthenon/mesh/x3max=0.2
rm -f gpurun_out/${tag}_*prof/*kernel_trace.csv gpurun_out/${tag}_*prof/*/*kernel_trace.csv
fi
if has timings; then
# ---- refined meshes: the shipped Cartesian SMR disk, the refined spherical disk (configs[3]'s combination), both paths; configs[4] in 3-D
timeout 300 python scripts/smr_timing.py 20 | tee gpurun_out/${tag}_smr.txt
ARTEMIS_NO_ML_FUSED=1 timeout 300 python scripts/smr_timing.py 20 | sed 's/^/per-task chain: /' | tee -a gpurun_out/${tag}_smr.txt
timeout 300 python scripts/smr_timing.py 20 sph problem/polytropic_index=1.40 gas/de_switch=1e-2 | tee -a gpurun_out/${tag}_smr.txt
ARTEMIS_NO_ML_FUSED=1 timeout 300 python scripts/smr_timing.py 20 sph problem/polytropic_index=1.40 gas/de_switch=1e-2 | sed 's/^/per-task chain: /' | tee -a gpurun_out/${tag}_smr.txt
timeout 900 python3 scripts/amr_timing.py 10 128 128 16 16 gas/refine_thr=2.0 parthenon/mesh/x3min=-0.2 parthenon/mesh/x3max=0.2 2>&1 | tail -1 | tee gpurun_out/${tag}_amr.txt
timeout 900 python3 scripts/amr_timing.py 10 128 128 32 32 gas/refine_thr=2.0 parthenon/mesh/x3min=-0.2 parthenon/mesh/x3max=0.2 2>&1 | tail -1 | sed 's/^/32^3 blocks (the deck\x27s own block size): /' | tee -a gpurun_out/${tag}_amr.txt
for w in blast_sph blast_cyl disk_sph disk_cyl disk_axi; do timeout 300 python scripts/curv_timing.py $w; done | tee gpurun_out/${tag}_curv.txt
fi
head -4 gpurun_out/${tag}_bench_kernel_stats.csv 2>/dev/null | cut -c1-200
