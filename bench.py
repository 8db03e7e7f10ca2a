#!/usr/bin/env python3
"""bench.py -- zone-cycles/s of the Artemis hydro update on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: under a launcher (WORLD_SIZE set) this process is one rank; without one the
process becomes a plain parent that starts N child processes of this file (one per LOCAL_RANK, rendezvous
on 127.0.0.1), relays rank 0's JSON line and exits non-zero if any child does -- it never touches the GPU
itself (self_launch() runs before torch or the HIP library are imported).

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): the
reference's Sedov deck inputs/blast/blast.in in 3-D -- Cartesian 256^3 cells per GPU, gas only,
HLLC + PLM, rk2, cfl 0.3, gamma 1.4, outflow, nghost 2, floors 1e-10 -- weak-scaled over the
GPUs of one node (256x256x512, 256x512x512, 256x512x1024 for 2/4/8 GPUs, one 256^3 mesh block
per rank, ranks cut along x3 then x2, never along x1: rank grid 1x1x2, 1x2x2, 1x2x4 -- see DECOMPOSITION).  One "step" = one full cycle: every RK stage (fused flux/update/source/c2p kernel +
ghost fill), the CFL reduction and, for N > 1, the halo exchange and the dt all-reduce.
Initial data are generated on the host and are resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (the fused stage
kernel) with the ALGORITHMIC bytes of SURVEY.md section 8(d) -- 240 B per cell-stage -- over
its HIP-event-timed launch duration; `cpu_baseline` times the CPU oracle (a restatement of
the reference's CPU path, NOT the Artemis executable, which cannot be built here) on the host
cores for a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# cpu_baseline leg: pin the oracle's OpenMP threads (read when the OpenMP runtime starts, so set before
# anything loads it)
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")

ALG_BYTES_PER_CELL_STAGE = 240.0   # 30 doubles
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8.0 TB/s spec


# Rank grid (x1, x2, x3) per GPU count.  BASELINE.md 2.2 suggests 2x2x2 -> 512^3 at 8 GPUs; this bench keeps the same
# zones per GPU and the same total (2^27 zones at 8 GPUs) but never cuts x1: an x1 face slab is two zones out of
# every 260-zone row (16-byte pieces of 128-byte lines for the pack / unpack kernels and for the boundary shell of
# the stage kernel), while x2 / x3 slabs are whole rows.  1x2x4 also never gives a rank more faces to exchange than
# 2x2x2 does (interior ranks 3, end ranks 2; 2x2x2: 3 everywhere).  DESIGN.md section 6 states the same grid.
DECOMPOSITION = {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (1, 2, 4)}


def self_launch(argv, n):
    """`python bench.py --gpus N` without a launcher: start N children of this file (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, one per GPU), relay rank 0's stdout (the JSON line), exit non-zero if any child fails.  Runs before
    anything that could initialise the GPU; the children are ordinary child processes (no exec of this process).
    ARTEMIS_BENCH_CHILD_CMD (a JSON list) replaces `[python, bench.py]` -- the hook of tests/test_bench_launcher.py."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    child = json.loads(os.environ["ARTEMIS_BENCH_CHILD_CMD"]) if os.environ.get("ARTEMIS_BENCH_CHILD_CMD") else \
        [sys.executable, os.path.abspath(__file__)]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs on this driver; a user's value wins)
        env.pop("ARTEMIS_BENCH_CHILD_CMD", None)
        procs.append(subprocess.Popen(child + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)
    reader.start()  # (a blocking read here would never notice another rank dying while rank 0 waits for it)
    rcs = [None] * n
    deadline = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(rc not in (None, 0) for rc in rcs) and deadline is None:
            deadline = time.time() + float(os.environ.get("ARTEMIS_BENCH_GRACE_S", "30"))  # a rank died: the others may sit in a collective for ever
        if deadline is not None and time.time() > deadline:
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.kill()  # exactly the children started above
                    rcs[r] = p.wait()
        time.sleep(0.05)
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print("bench.py: rank(s) failed: %s" % bad, file=sys.stderr, flush=True)
        raise SystemExit(next(rc for _, rc in bad) or 1)
    reader.join()
    sys.stdout.buffer.write(got[0])
    sys.stdout.flush()
    raise SystemExit(0)


def overrides(n_gpus, per_gpu, steps_total, extra=()):
    # weak scaling grows the mesh along x3, then x2: ranks are never cut along x1, so rows stay
    # 256 cells long and the boundary shell of the stage kernel stays thin
    shape = DECOMPOSITION.get(n_gpus)
    if shape is None:
        raise SystemExit("--gpus must be 1, 2, 4 or 8")
    ov = ["gas/riemann=hllc", "problem/symmetry=spherical", "problem/radius=0.03",
          "problem/samples=0", "parthenon/time/tlim=-1.0", "parthenon/time/nlim=-1"]  # every evolve() passes its budget
    for d, (s, n) in enumerate(zip(shape, per_gpu), start=1):
        half = float(s) * n / per_gpu[0] * (per_gpu[0] / 256.0)  # dx = 2/256 whatever the size
        ov += [f"parthenon/mesh/nx{d}={s * n}", f"parthenon/meshblock/nx{d}={n}",
               f"parthenon/mesh/x{d}min={-half}", f"parthenon/mesh/x{d}max={half}"]
    return ov + list(extra)


def cpu_baseline(n, cycles, threads):
    """Time the CPU oracle on `threads` host cores: n^3 Sedov, `cycles` cycles after one warm-up.
    (The oracle's pgen and its first cycle first-touch every array inside the same OpenMP
    k-j decomposition the sweeps use, so pages sit on the NUMA node of the thread that works on them.)"""
    from oracle.oracle import Oracle
    o = Oracle((n, n, n), (-n / 256.0,) * 3, (n / 256.0,) * 3, ng=2, reconstruct="plm",
               riemann="hllc", gamma=1.4, dfloor=1e-10, siefloor=1e-10, cfl=0.3,
               bc=("outflow",) * 6, integrator="rk2", nthreads=threads)
    o.pgen_blast(radius=0.03, internal_energy=1.0, p0=1e-5, d0=1.0, samples=0)
    o.evolve(-1.0, 1)
    t0 = time.perf_counter()
    done = o.evolve(-1.0, 1 + cycles)
    dt = time.perf_counter() - t0
    return n ** 3 * done / dt, dt, done


def cpu_baseline_disk(kind, cycles, threads):
    """The disk workloads on the CPU oracle, bounded: `sph` = inputs/disk/disk_sph.in as shipped, 128 x 64 x 64 (gas, point-mass
    gravity, alpha viscosity, rotating frame, ic conditions); `cyl_dust` = the cylindrical disk of configs[4] on a
    uniform 128 x 128 x 16 mesh over |z| < 0.2 (bench.py's root mesh) with one dust species, simple_dust drag, the planet as N-body gravity, alpha
    viscosity and the rotating frame (the oracle has no adaptive mesh of its own: oracle/adaptive.py drives it per block)."""
    import math
    from oracle.oracle import Oracle
    pi = 3.141592653589793
    if kind == "sph":
        nx = (128, 64, 64)
        o = Oracle(nx, (0.2, 1.059856161608513, -pi), (5.6, 2.081736491981280, pi), ng=2, reconstruct="plm", riemann="hlle",
                   gamma=1.4, dfloor=1e-10, siefloor=1e-10, cfl=0.3, integrator="rk2", coordinates="spherical",
                   bc=("ic", "ic", "ic", "ic", "periodic", "periodic"), nthreads=threads)
        o.set_gravity_point(mass=1.0)
        o.set_rotating_frame(1.0, 0.0)
        o.set_viscosity("alpha", alpha=1e-3, r0=1.0, Omega0=1.0)
    else:
        nx = (128, 128, 16)
        o = Oracle(nx, (0.3, -pi, -0.2), (4.3, pi, 0.2), ng=2, reconstruct="plm", riemann="hllc", gamma=1.4, dfloor=1e-10,
                   siefloor=1e-10, cfl=0.3, integrator="rk2", coordinates="cylindrical", ns_dust=1, dust_reconstruct="plm",
                   dust_riemann="hlle", dust_dfloor=1e-10, dust_cfl=0.3,
                   bc=("ic", "ic", "periodic", "periodic", "ic", "ic"), nthreads=threads)
        parts = [dict(GM=1.0, pos=(-0.01, 0.0, 0.0), vel=(0.0, -0.01, 0.0), rs=0.0, spline=0, couple=1),
                 dict(GM=0.01, pos=(0.99, 0.0, 0.0), vel=(0.0, 0.99, 0.0), rs=0.03, spline=0, couple=1)]
        o.set_gravity_nbody(parts, frame_correction=True, gm=1.01)
        o.set_rotating_frame(1.0, 0.0)
        o.set_viscosity("alpha", alpha=1e-3, r0=1.0, Omega0=math.sqrt(1.01))
        o.set_drag("simple_dust", "constant", tau=[0.1])
    o.pgen_disk(r0=1.0, rho0=1.0, dslope=-2.25, flare=0.25, h0=0.05, dens_min=1e-10, pres_min=1e-15, polytropic_index=1.0)
    o.evolve(62.8, 1)
    t0 = time.perf_counter()
    o.evolve(62.8, 1 + cycles)
    dt = time.perf_counter() - t0
    return nx[0] * nx[1] * nx[2] * cycles / dt, dt, cycles, nx


def host_cores():
    """What this process may actually use: logical CPUs in its affinity mask, clipped by a cgroup CPU
    quota, and the SMT width (cpu_baseline runs one thread per physical core)."""
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    smt = 1
    try:
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        smt = max(1, len(sib.replace("-", ",").split(",")))
    except Exception:
        pass
    # a CPU quota is in units of CPU time: run that many threads; without one, one thread per physical core
    usable = max(1, logical // smt) if quota is None else max(1, min(logical, int(quota)))
    return {"logical": logical, "cgroup_quota": quota, "smt": smt, "physical_usable": usable}


def library_identity(scope="fused"):
    """Identity of the code a PMC record was measured on, as the LOADED LIBRARY reports it (artemis_amd/build.py compiles
    the content hashes in): scope "fused" = the translation unit of the tuned stage kernel (source, every header it
    reaches, flags, compiler) -- the Sedov headline; "all" = every source of the library (whole-stage records of the other
    workloads, which run several kernels per stage).  A record is quoted next to a timing only while the library that
    produced the timing reports the identity the record carries (scripts/pmc_traffic.py writes it the same way)."""
    from artemis_amd import capi
    return capi.object_sha("kernels_fused") if scope == "fused" else capi.source_sha()


def measured_traffic(name):
    """HBM bytes per launch (Sedov) / per stage (other workloads) from the newest profiles/r*<name>pmc_traffic.json
    whose identity is the loaded library's, else None (a stale record is never reported)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*%spmc_traffic.json" % name)), reverse=True):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        if name == "" and rec.get("workload", "sedov3d") != "sedov3d":
            continue
        want = library_identity(rec.get("sha_scope", "fused"))
        if rec.get("library_identity") != want or rec.get("env"):  # the loaded library's own code, default knobs only
            continue
        if name == "" and (not rec.get("headline_launches") or rec.get("headline_launches") != rec.get("headline_launches_expected")):
            # a Sedov record must be a mean over full-size launches of the two headline instantiations and nothing
            # else (round 4's mixed in the half-size shell / bulk launches of the overlap emulation): refuse it
            continue
        return rec.get("hbm_bytes_per_launch"), os.path.relpath(path, ROOT)
    return None, None


def measured_valu():
    """VALU wave-instructions per launch of the headline stage kernel from the newest profiles/r*_pmc_sq.json whose
    identity is the loaded library's (scripts/pmc_sq.py with PMC_SQ_RECORD), else None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        if rec.get("library_identity") == library_identity("fused") and not rec.get("env"):
            return rec.get("valu_wave_instructions_per_launch"), os.path.relpath(path, ROOT)
    return None, None


def cpu_baseline_ssheet(n, ndust, cycles, threads):
    """The config-3 workload on the CPU oracle: n^2 dusty shearing sheet with drag."""
    from oracle.oracle import Oracle
    o = Oracle((n, n, 1), (-1.0, -1.0, -0.2), (1.0, 1.0, 0.2), ng=2, ns_gas=1, ns_dust=ndust, reconstruct="plm",
               riemann="hllc", dust_reconstruct="plm", dust_riemann="hlle", gamma=1.000001, dfloor=1e-10,
               siefloor=1e-10, dust_dfloor=1e-10, cfl=0.3, dust_cfl=0.3,
               bc=("extrap", "extrap", "inflow", "inflow", "extrap", "extrap"), integrator="rk2", nthreads=threads)
    o.set_rotating_frame(1.0, 1.5)
    o.set_gravity_point(1e-5, soft=0.03)
    o.set_drag("simple_dust", "constant", tau=[0.1] * ndust)
    o.pgen_strat(rho0=1.0, dens_min=1e-10, h=0.05)
    o.evolve(-1.0, 1)
    t0 = time.perf_counter()
    done = o.evolve(-1.0, 1 + cycles)
    dt = time.perf_counter() - t0
    return n * n * done / dt, dt, done, n


def batch_of_leaves(sim, fraction=0.025):
    """Every 1/fraction-th leaf below the finest level, spread over the Z-ordered list: what a criterion that fires on a
    moving feature tags in one go."""
    top = max(sim.block_level(b) for b in range(sim.nblocks))
    cand = [g for g in range(sim.nblocks) if sim.block_level(g) < top]
    step = max(1, int(round(1.0 / fraction)))
    return cand[step // 2::step]


def record_remesh(sim, events, what, call):
    """Run one remesh-triggering call and append what it did (leaves, milliseconds, split) to events."""
    n0 = sim.remeshes
    changed = call()
    if changed and sim.remeshes != n0:
        lv, sec = sim.last_remesh()
        events.append({"what": what, "leaves_before": lv[0], "leaves_after": lv[1], "created": lv[2], "destroyed": lv[3],
                       "ms": 1.0e3 * sec[0], "ms_build_state": 1.0e3 * sec[1], "ms_hand_over": 1.0e3 * sec[2]})
    return changed


def main():
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner through C
    # stdio when a communicator is created): keep the caller's stdout for the line and send everything else that
    # lands on descriptor 1 to stderr.
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=256, help="cells per GPU per dimension")
    ap.add_argument("--path", default="fused", choices=["fused", "unfused"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the `dropin` legs (fused + cons + whole-block PrimToCons, and the per-task chain)")
    ap.add_argument("--no-overlap-emulation", action="store_true",
                    help="skip the `overlap_emulation` leg (two half-size blocks, shell-first + bulk launches): profiles of "
                         "the headline kernel then hold 256^3 launches only")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: exchange halos after the whole stage kernel instead of behind its bulk")
    ap.add_argument("--overlap-mode", type=int, default=1, choices=[1, 2],
                    help="1 (default): shell launch, then bulk launch, the exchange behind an event between them; "
                         "2: one launch, shell workgroups first + a device counter a one-wave kernel polls")
    ap.add_argument("--blocks-per-gpu", type=int, default=1,
                    help="diagnostic: cut each rank's 256^3 into this many mesh blocks along x3")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: if the native C++ RCCL transport cannot be created, carry on over torch.distributed's nccl "
                         "backend (reported in config.transport) instead of failing -- a scaling line is the product's own "
                         "transport unless this is given")
    ap.add_argument("--loopback", action="store_true",
                    help="diagnostic: route block-to-block slabs of ONE GPU through RCCL send/recv-to-self")
    ap.add_argument("--no-remesh-leg", action="store_true", help="disk_amr: skip the remesh measurements after the timed region")
    ap.add_argument("--remesh-in-timed-region", action="store_true",
                    help="disk_amr: inject the batched refinement tags of the remesh leg INSIDE the timed region (every "
                         "4th cycle ~2.5 %% of the leaves; the criterion merges them again derefine_count cycles later), so "
                         "that `value` includes the cost of a mesh that keeps changing")
    ap.add_argument("--workload", default="sedov3d", choices=["sedov3d", "ssheet_dust", "disk_sph", "disk_sph_smr", "disk_amr", "linwave3d"],
                    help="sedov3d = the headline metric (BASELINE configs[1]); ssheet_dust = SURVEY config 3 "
                         "(2-D dusty shearing sheet with drag, general fused stage; --n is the mesh edge, 1 GPU); "
                         "disk_sph = BASELINE configs[3] without refinement (inputs/disk/disk_sph.in, spherical-polar "
                         "alpha disk; --n scales the 128x64x64 deck mesh by n/128, 1 GPU)")
    ap.add_argument("--dust", type=int, default=1, help="ssheet_dust: number of dust species")
    ap.add_argument("--recon", default="ppm", choices=["plm", "ppm"],
                    help="linwave3d (inputs/linwave/linear_wave.in at n^3 in one block, HLLC, periodic): the reconstruction -- ppm "
                         "runs the cell-centred general stage (PPM is not in the tile marches: DESIGN.md section 8), plm the "
                         "tuned tile march; the pair states the gap")
    ap.add_argument("--amr-block", type=int, default=16, choices=[16, 32],
                    help="disk_amr: zones per block edge -- 16 (default: the stress case, 7 064 blocks at N = 1) or 32 (the block "
                         "size of the reference's deck inputs/disk/disk_nbody_cyl.in)")
    ap.add_argument("--uniform", action="store_true",
                    help="sedov3d diagnostic (SURVEY 8d): the same deck with an empty blast region, i.e. a uniform gas at "
                         "rest -- same kernels, no shocks -- to separate branch-divergence effects from the rest")
    ap.add_argument("--cpu-n", type=int, default=256)
    ap.add_argument("--cpu-cycles", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the cpu_baseline leg (0 = all host cores)")
    args = ap.parse_args()
    if args.gpus not in DECOMPOSITION:
        raise SystemExit("--gpus must be 1, 2, 4 or 8")
    if args.gpus > 1 and args.workload in ("ssheet_dust", "disk_sph", "linwave3d"):
        raise SystemExit("--workload %s is a single-GPU measurement (one mesh block); the N-GPU workloads are sedov3d, "
                         "disk_sph_smr and disk_amr" % args.workload)
    if args.gpus > 1 and args.remesh_in_timed_region:
        raise SystemExit("--remesh-in-timed-region injects tags by local leaf index: a single-GPU measurement")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(sys.argv[1:], args.gpus)  # does not return
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from artemis_amd import capi
    from artemis_amd.driver import RcclComm, Simulation

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    L = capi.load()
    capi.check(L.artemis_rt_set_device(local_rank))
    comm = None
    dist = None
    if world > 1 or args.loopback:
        # Data path: the native C++ RCCL transport (csrc/driver/comm_rccl.cpp), called straight from the
        # driver's streams.  torch.distributed is used for CPU-side plumbing only (gloo): shipping rank 0's
        # ncclUniqueId and the barrier / max-over-ranks around the timed region.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        share = None
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")

            def share(uid):
                box = [uid]
                dist.broadcast_object_list(box, src=0)
                return box[0]
        else:
            capi.check(L.artemis_hip_set_option(b"loopback_comm", 1))
        transport_note = None
        try:
            comm = RcclComm(rank, world, share)
            assert comm.count == world, (comm.count, world)
            ok = 1
        except Exception as e:  # noqa: BLE001 -- reported in the JSON line, never silent
            if world == 1:
                raise
            comm, ok, transport_note = None, 0, repr(e)
        if world > 1:
            # every rank must take the same transport
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and not args.allow_fallback:
                raise SystemExit("bench.py: the native RCCL transport could not be created on every rank (%s); "
                                 "--allow-fallback measures over torch.distributed's nccl backend instead" % transport_note)
            if int(flag.item()) == 0:
                # Opt-in last resort (--allow-fallback): torch.distributed's nccl (= RCCL) backend behind the same
                # artemis_comm_t callbacks.  The JSON line says so; the native path is the product.
                from artemis_amd.driver import TorchComm
                if comm is not None:
                    comm.close()
                print("bench.py: native RCCL transport unavailable (%s); falling back to torch.distributed nccl"
                      % transport_note, file=sys.stderr, flush=True)
                comm = TorchComm(torch.device("cuda", local_rank), group=dist.new_group(backend="nccl"))
                comm.count = world
                comm.fallback = transport_note or "another rank failed to create the native communicator"
                comm.barrier = lambda g=comm.group: dist.barrier(group=g)
                comm.close = lambda: None

    per_gpu = (args.n, args.n, args.n)
    if args.workload == "ssheet_dust":
        deck = os.path.join(ROOT, "inputs", "ssheet", "ssheet.in")
        n = str(args.n)
        ov = ["parthenon/mesh/nx1=" + n, "parthenon/mesh/nx2=" + n, "parthenon/meshblock/nx1=" + n,
              "parthenon/meshblock/nx2=" + n, "physics/dust=true", "physics/drag=true",
              "dust/nspecies=%d" % args.dust, "dust/cfl=0.3", "dust/reconstruct=plm", "dust/riemann=hlle",
              "dust/dfloor=1.0e-10", "dust/stopping_time/type=constant",
              "dust/stopping_time/tau=" + ",".join(["0.1"] * args.dust), "drag/type=simple_dust",
              "parthenon/time/tlim=-1.0", "parthenon/time/nlim=-1"]
        make_sim = lambda: Simulation(deck, ov)
        sim = make_sim()
    elif args.workload == "disk_sph":
        deck = os.path.join(ROOT, "inputs", "disk", "disk_sph.in")
        sc = max(1, args.n // 128)
        dims = (128 * sc, 64 * sc, 64 * sc)
        ov = ["parthenon/time/nlim=-1"]
        for d, m in enumerate(dims, 1):
            ov += ["parthenon/mesh/nx%d=%d" % (d, m), "parthenon/meshblock/nx%d=%d" % (d, m)]
        make_sim = lambda: Simulation(deck, ov)
        sim = make_sim()
    elif args.workload == "linwave3d":
        deck = os.path.join(ROOT, "inputs", "linwave", "linear_wave.in")
        n = args.n
        ov = ["parthenon/time/nlim=-1", "problem/nperiod=1000", "gas/reconstruct=" + args.recon,
              "parthenon/mesh/nghost=%d" % (3 if args.recon == "ppm" else 2)]
        for d in (1, 2, 3):
            ov += ["parthenon/mesh/nx%d=%d" % (d, n), "parthenon/meshblock/nx%d=%d" % (d, n)]
        make_sim = lambda: Simulation(deck, ov)
        sim = make_sim()
    elif args.workload == "disk_sph_smr":
        # BASELINE configs[3] (an 8-GPU configuration): the spherical-polar disk deck x 2 in 32^3 blocks with a level-1
        # static region around the midplane (scripts/smr_timing.py sph; tests/test_multilevel.py runs its small form).
        # N ranks: the root mesh grows N-fold along x3 (azimuth; same domain, same zones per GPU), the Z-ordered leaf
        # list is cut into N consecutive runs, ghost zones between runs travel through the native RCCL transport.
        deck = os.path.join(ROOT, "inputs", "disk", "disk_sph.in")
        ov = ["parthenon/time/nlim=-1", "parthenon/mesh/nx1=256", "parthenon/mesh/nx2=128", "parthenon/mesh/nx3=%d" % (128 * args.gpus),
              "parthenon/mesh/refinement=static", "parthenon/static_refinement1/level=1",
              "parthenon/static_refinement1/x1min=0.7", "parthenon/static_refinement1/x1max=1.9",
              "parthenon/static_refinement1/x2min=1.3", "parthenon/static_refinement1/x2max=1.85",
              "parthenon/static_refinement1/x3min=-3.0", "parthenon/static_refinement1/x3max=3.0",
              "problem/polytropic_index=1.40", "gas/de_switch=1e-2"]
        make_sim = lambda: Simulation(deck, ov, comm=comm)
        sim = make_sim()
    elif args.workload == "disk_amr":
        # BASELINE configs[4] (an 8-GPU configuration), 3-D: inputs/disk/disk_nbody_cyl.in + a planet + one dust species
        # with drag + the rotating frame + adaptive refinement to four levels (scripts/amr_timing.py; the 2-D form runs
        # against the adaptive oracle in tests/test_adaptive.py).  N ranks: the root mesh grows N-fold along x2 (azimuth),
        # Z-order runs of leaves per rank, blocks migrate at remeshes.  --amr-block 32 = the deck's own block size.
        deck = os.path.join(ROOT, "inputs", "disk", "disk_nbody_cyl.in")
        mb = args.amr_block
        ov = ["parthenon/mesh/nx1=128", "parthenon/mesh/nx2=%d" % (128 * args.gpus), "parthenon/mesh/nx3=%d" % max(16, mb),
              "parthenon/mesh/x3min=-0.2", "parthenon/mesh/x3max=0.2",
              "parthenon/meshblock/nx1=%d" % mb, "parthenon/meshblock/nx2=%d" % mb, "parthenon/meshblock/nx3=%d" % mb,
              "parthenon/mesh/refinement=adaptive", "parthenon/mesh/numlevel=4", "parthenon/mesh/derefine_count=5",
              "gas/refine_field=pressure", "gas/refine_type=gradient", "gas/refine_thr=2.0",
              "physics/rotating_frame=true", "rotating_frame/omega=1.0",
              "physics/dust=true", "dust/nspecies=1", "dust/cfl=0.3", "dust/reconstruct=plm", "dust/riemann=hlle",
              "dust/dfloor=1e-10", "physics/drag=true", "drag/type=simple_dust", "dust/stopping_time/type=constant",
              "dust/stopping_time/tau=0.1", "dust/sizes=1.0",
              "nbody/particle2/mass=1.0e-2", "nbody/particle2/couple=1", "nbody/particle2/soft/type=plummer",
              "nbody/particle2/soft/radius=0.03", "nbody/particle2/initialize/x=1.0", "nbody/particle2/initialize/vy=1.0",
              "parthenon/time/nlim=-1"]
        make_sim = lambda: Simulation(deck, ov, comm=comm)
        sim = make_sim()
    else:
        deck = os.path.join(ROOT, "inputs", "blast", "blast.in")
        extra = []
        if args.blocks_per_gpu > 1:
            extra = ["parthenon/meshblock/nx3=%d" % (args.n // args.blocks_per_gpu)]
        if args.uniform:
            extra = extra + ["problem/radius=0.0"]
        make_sim = lambda: Simulation(deck, overrides(args.gpus, per_gpu, args.warmup + args.steps, extra), comm=comm)
        sim = make_sim()
    if args.path == "unfused":
        sim.set_path("unfused")
    force_overlap = L.artemis_hip_get_option(b"force_overlap") > 0  # (ARTEMIS_FORCE_OVERLAP in the environment)
    # (shell-first overlap is the uniform-mesh stage's; on refined meshes the exchange is ordered behind the stage kernels)
    want_overlap = (world > 1 or args.loopback or force_overlap) and not args.no_overlap and args.workload == "sedov3d"
    sim.set_overlap(args.overlap_mode if want_overlap else 0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if comm is not None:
            comm.barrier()  # RCCL all-reduce + stream sync: every rank's device work is done
        torch.cuda.synchronize()

    wait_timeout = False
    try:
        sim.evolve(args.warmup)
        bad = 0
    except RuntimeError as e:
        # overlap mode 2's wait kernel gave up (its polling wave was not co-resident with the stage kernel's grid):
        # the driver has already switched overlapping off; say so and carry on with two launches per stage
        if "timed out" not in str(e) or not want_overlap:
            raise
        bad = 1
    if world > 1:  # every rank takes the same mode
        flag = torch.tensor([bad], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        bad = int(flag.item())
    if bad:  # start again from the initial condition (the failed run's ghost zones may hold unfinished shell data)
        wait_timeout = True
        sim.close()
        sim = make_sim()
        if args.path == "unfused":
            sim.set_path("unfused")
        sim.set_overlap(1)
        sim.evolve(args.warmup)
    overlap_used = sim.overlap
    # The timed region runs the production loop (device-resident dt, hipGraph replay on one rank, no
    # per-kernel events); per-kernel durations for `roofline` come from a separate short leg below.
    barrier()
    t0 = time.perf_counter()
    if args.workload == "disk_amr":  # the mesh changes between cycles: count the zones of every cycle
        zone_cycles, done, remesh0 = 0, 0, sim.remeshes
        timed_events = []
        for cyc in range(args.steps):
            if args.remesh_in_timed_region and cyc % 4 == 3:
                record_remesh(sim, timed_events, "injected tags", lambda: sim.inject_refine_tags(batch_of_leaves(sim)))
            zone_cycles += sim.total_zones
            r_before = sim.remeshes
            done += sim.evolve(1)
            if args.remesh_in_timed_region and sim.remeshes != r_before:
                lv, sec = sim.last_remesh()
                timed_events.append({"what": "criterion (after cycle)", "leaves_before": lv[0], "leaves_after": lv[1], "created": lv[2],
                                     "destroyed": lv[3], "ms": 1.0e3 * sec[0], "ms_build_state": 1.0e3 * sec[1], "ms_hand_over": 1.0e3 * sec[2]})
    else:
        done = sim.evolve(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    sim_remeshes_timed = (sim.remeshes - remesh0) if args.workload == "disk_amr" else 0
    assert done == args.steps, (done, args.steps)
    elapsed_min = elapsed_max = elapsed
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        tmin = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        elapsed = elapsed_max = float(t.item())
        elapsed_min = float(tmin.item())
    remesh_leg = None
    if args.workload == "disk_amr" and not args.no_remesh_leg and world == 1 and not args.loopback:  # (the legs pick leaves by local index: one rank)
        # The remesh machinery on THIS mesh, outside the timed region: five leaves below the finest level, spread over
        # the Z-ordered list, are split one after the other as if the criterion had tagged them (2:1 balance, new
        # state, hand-over, tables), one cycle in between.  The disk is in equilibrium, so the deck's own criterion
        # leaves the 3-D mesh alone for hundreds of cycles (the thin-slab form of tests/amr_cases.py remeshes every
        # few cycles and is what the parity tests run).
        n_timed, s_timed = sim.remesh_seconds()[:2]
        remeshes_timed = int(sim.remeshes - remesh0)
        blocks0, each = sim.nblocks, []
        sim.device_bytes(reset_peak=True)
        cand = [g for g in range(sim.nblocks) if sim.block_level(g) < max(sim.block_level(b) for b in range(sim.nblocks))]
        picks = [cand[(2 * q + 1) * len(cand) // 10] for q in range(5)] if cand else []
        parts0 = sim.remesh_seconds()
        for gid in picks:
            before = sim.remesh_seconds()[1]
            if sim.force_refine(min(gid, sim.nblocks - 1)):
                each.append(1.0e3 * (sim.remesh_seconds()[1] - before))
            sim.evolve(1)
        parts1 = sim.remesh_seconds()
        split_ms = {k: 1.0e3 * (parts1[q] - parts0[q]) / max(1, len(each))
                    for q, k in ((2, "build_state"), (3, "hand_over"), (4, "tagging_incl_cycles_without_remesh"))}
        torch.cuda.synchronize()
        cycle_ms = 1.0e3 * elapsed / args.steps
        remesh_leg = {"remeshes_in_timed_region": remeshes_timed, "seconds_in_timed_region": s_timed,
                      "forced": len(each), "ms_each": each, "ms_mean": (sum(each) / len(each)) if each else None,
                      "ms_mean_split": split_ms,
                      "cycle_ms": cycle_ms, "remesh_over_cycle": (sum(each) / len(each) / cycle_ms) if each else None,
                      "blocks_before": blocks0, "blocks_after": sim.nblocks,
                      "device_bytes_now": sim.device_bytes()[0], "device_bytes_peak_during_remesh": sim.device_bytes()[1],
                      "what": "five forced single-leaf refinements through the ordinary remesh path (artemis_sim_force_refine), "
                              "one cycle apart, after the timed region"}
    if remesh_leg is not None:
        # ... and MANY leaves at once, the way a criterion that follows a moving feature tags them: five times, four cycles
        # apart, ~2.5 % of the leaves (spread over the Z-ordered list) get the tag +1 next to the deck's own tags; the
        # criterion does not want them refined, so `derefine_count` (5) cycles later the SAME leaves merge again -- remeshes
        # of that size by the ordinary path, tagged by gas/refine_* itself.  (The deck's own criterion alone leaves this
        # mesh unchanged for hundreds of cycles: the disk is in equilibrium and features move one finest zone in ~6 cycles.)
        events = []
        sim.device_bytes(reset_peak=True)
        zones_at = []
        for cyc in range(28):
            if cyc % 4 == 0 and cyc < 20:
                record_remesh(sim, events, "injected tags (+1 on ~2.5 % of the leaves)", lambda: sim.inject_refine_tags(batch_of_leaves(sim)))
            r_before = sim.remeshes
            sim.evolve(1)
            if sim.remeshes != r_before:
                lv, sec = sim.last_remesh()
                events.append({"what": "the deck's criterion (merges what it does not want refined)", "leaves_before": lv[0],
                               "leaves_after": lv[1], "created": lv[2], "destroyed": lv[3], "ms": 1.0e3 * sec[0],
                               "ms_build_state": 1.0e3 * sec[1], "ms_hand_over": 1.0e3 * sec[2]})
            zones_at.append(sim.total_zones)
        torch.cuda.synchronize()
        big = [e for e in events if (e["created"] + e["destroyed"]) >= 0.02 * e["leaves_before"]]
        cur_b, peak_b = sim.device_bytes()
        remesh_leg["batched"] = {
            "events": events, "remeshes": len(events), "remeshes_changing_2pct_of_leaves": len(big),
            "ms_mean": (sum(e["ms"] for e in big) / len(big)) if big else None,
            "ms_max": max((e["ms"] for e in big), default=None),
            "build_state_share": (sum(e["ms_build_state"] for e in big) / max(1e-30, sum(e["ms"] for e in big))) if big else None,
            "over_cycle_mean": (sum(e["ms"] for e in big) / len(big) / remesh_leg["cycle_ms"]) if big else None,
            "over_cycle_max": (max(e["ms"] for e in big) / remesh_leg["cycle_ms"]) if big else None,
            "device_bytes_now": cur_b, "device_bytes_peak": peak_b, "zones_now": sim.total_zones,
            "device_bytes_cached": sim.cached_bytes(),  # (of device_bytes_now: free buffers the library's cache holds for the next remesh)
            "bytes_per_zone_live": (cur_b - sim.cached_bytes()) / max(1, sim.total_zones),
            "bytes_per_zone_now": cur_b / max(1, sim.total_zones), "bytes_per_zone_peak": peak_b / max(1, min(zones_at)),
            "what": "28 cycles after the timed region; five batches of injected +1 tags, the merges the criterion orders five "
                    "cycles after each"}
    kms, nlaunch = 0.0, 0
    if args.workload not in ("disk_sph_smr", "disk_amr"):  # (refined meshes: whole-stage accounting below)
        sim.set_kernel_timing(True)   # HIP events around the dominant kernel, on the stream it is launched on
        sim.evolve(max(5, min(args.steps, 40)))
        kms, nlaunch = sim.kernel_ms()
        sim.set_kernel_timing(False)
    hist = sim.history()
    overlap_emulation = None
    if (args.workload == "sedov3d" and args.gpus == 1 and not args.loopback and not args.no_overlap_emulation and sim.uses_tuned_kernel
            and args.blocks_per_gpu == 1 and not force_overlap):
        # What a rank of an N > 1 run does per stage, emulated on this one GPU: the same zones as two blocks stacked along
        # x3 with the shell-first / bulk launch order forced (ARTEMIS_FORCE_OVERLAP: the boundary shell of every block is
        # launched first and its slabs are packed on the comm stream while the bulk runs; the "link" is a device copy).
        # The N > 1 expectation per GPU, before any xGMI time, is this number rather than the one-block headline.
        capi.check(L.artemis_hip_set_option(b"force_overlap", 1))
        try:
            s2 = Simulation(deck, overrides(1, per_gpu, 64, ["parthenon/meshblock/nx3=%d" % (args.n // 2)]))
            s2.set_overlap(1)
            s2.evolve(5)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n2 = s2.evolve(max(10, min(args.steps, 40)))
            torch.cuda.synchronize()
            overlap_emulation = s2.total_zones * n2 / (time.perf_counter() - t1)
            s2.close()
        finally:
            L.artemis_hip_set_option(b"force_overlap", 0)
    dropin = None
    if args.workload == "sedov3d" and args.gpus == 1 and not args.loopback and not args.no_dropin and sim.uses_tuned_kernel:
        # What a Parthenon host sees: (i) the fused kernel also writing `cons` on the last
        # stage + the whole-block PrimToCons (FillDerived) after every stage's boundary fill; (ii) the
        # per-task chain of INTEGRATION.md section 2.  Same deck, same state, outside the timed region.
        def leg(cycles):
            sim.evolve(2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_ = sim.evolve(cycles)
            torch.cuda.synchronize()
            return sim.total_zones * n_ / (time.perf_counter() - t1)
        sim.set_dropin(1)
        r_cons = leg(max(10, min(args.steps, 60)))
        sim.set_dropin(2)
        r_cons_g = leg(max(10, min(args.steps, 60)))
        sim.set_dropin(0)
        sim.set_path("unfused")
        r_task = leg(max(5, min(args.steps, 25)))
        sim.set_path("fused")
        zc_bytes = 2.0 * ALG_BYTES_PER_CELL_STAGE  # rk2: 480 B per zone-cycle
        dropin = {"unit": "zone-cycles/s",
                  "fused": None, "fused_cons_p2c": r_cons, "fused_cons_ghost_p2c": r_cons_g, "per_task": r_task,
                  "frac_fused_cons_p2c": r_cons * zc_bytes / 1.0e9 / HBM_PEAK_GBS,
                  "frac_fused_cons_ghost_p2c": r_cons_g * zc_bytes / 1.0e9 / HBM_PEAK_GBS,
                  "frac_per_task": r_task * zc_bytes / 1.0e9 / HBM_PEAK_GBS,
                  "note": "fused = `value` (primitives in, primitives out; cons never materialised); "
                          "fused_cons_p2c = same kernel also writing cons on the last stage + the whole-block PrimToCons "
                          "a Parthenon FillDerived runs after every stage's boundary fill; fused_cons_ghost_p2c = the kernel stores cons of "
                          "the zones it updates in every stage and only the ghost zones go through PrimToCons after the fill "
                          "(cons equally current after every stage); per_task = one kernel per "
                          "Parthenon task (CalculateFluxes, epilogue = ApplyUpdate..ConsToPrim, BCs, PrimToCons). "
                          "Fractions use the same 480 B per zone-cycle."}
    rccl_ranks = comm.count if comm is not None else 0
    nblocks_global = sim.nblocks_global
    load_balance = sim.load_balance
    rank_zones = [sim.local_zones]
    if world > 1:
        rz = torch.zeros(world, dtype=torch.int64)
        rz[rank] = sim.local_zones
        dist.all_reduce(rz)
        rank_zones = [int(v) for v in rz.tolist()]
    total_zones = sim.total_zones
    local_zones = sim.local_zones
    fused = sim.uses_fused_path

    if rank == 0:
        value = total_zones * args.steps / elapsed
        if args.workload == "disk_amr":
            value = zone_cycles / elapsed
        out = {
            "metric": "cell-updates/sec (zone-cycles/s), 256^3/GPU Sedov",
            "value": value, "unit": "zone-cycles/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1.0e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "inputs/blast 3-D Sedov (BASELINE configs[1]): Cartesian %d^3 cells/GPU, "
                            "gas HLLC+PLM, rk2, cfl 0.3, gamma 1.4, outflow, nghost 2, radius 0.03, "
                            "samples 0%s" % (args.n, " -- UNIFORM-STATE diagnostic (radius 0)" if args.uniform else ""),
                "cells_per_gpu": local_zones, "path": "fused" if fused else "unfused",
                "decomposition": "%d rank(s) as a %dx%dx%d grid (x1, x2, x3; never cut along x1: x1 face slabs are 16-byte "
                                 "pieces of every row, x2 / x3 slabs whole rows; same zones per GPU and in total as BASELINE.md's "
                                 "2x2x2), one %d^3 mesh block each, face-slab halo exchange%s"
                                 % ((args.gpus,) + DECOMPOSITION[args.gpus] + (args.n, "" if world == 1 else
                                    (" on RCCL, not overlapped" if not overlap_used else
                                     " on RCCL on a second stream behind the bulk of the stage kernel"))),
                "rank_grid": list(DECOMPOSITION[args.gpus]),
                "overlap_mode": overlap_used,  # 0 none, 1 shell + bulk launches, 2 one launch + device counter
                "overlap_wait_timeout": wait_timeout,  # mode 2's wait kernel gave up during warm-up -> mode 1 was used
                "ncclCommCount": rccl_ranks,
                "ncclCommCount_matches_gpus": (rccl_ranks == args.gpus) if comm is not None else None,
                "rank_ms_per_step": {"min": 1.0e3 * elapsed_min / args.steps, "max": 1.0e3 * elapsed_max / args.steps},
                "transport": None if comm is None else
                             ("FALLBACK torch.distributed nccl backend (native RCCL transport failed: %s)" % comm.fallback
                              if getattr(comm, "fallback", None) else
                              "native C++ RCCL (ncclSend/ncclRecv groups per stage, ncclAllReduce(min) on the device dt); "
                              "ncclCommCount = %d rank(s)" % rccl_ranks),
                "total_energy_check": float(hist[4]),
            },
        }
        if overlap_emulation is not None:
            out["config"]["overlap_emulation_zcps"] = overlap_emulation
            out["config"]["overlap_emulation"] = ("the same zones as two blocks stacked along x3, shell-first + bulk launches forced "
                                                  "(mode 1) on one GPU: what one rank of an N > 1 run does per stage, before link time")
        if args.workload == "disk_sph":
            out["metric"] = "cell-updates/sec (zone-cycles/s), spherical-polar alpha disk"
            out["config"]["workload"] = ("BASELINE configs[3] without mesh refinement: inputs/disk/disk_sph.in scaled to "
                                         "%d x %d x %d, gas, point-mass gravity, alpha viscosity, rotating frame, ic "
                                         "conditions, HLLE + PLM_G, rk2" % dims)
            out["config"]["decomposition"] = "1 rank, one mesh block"
            # whole-stage accounting (several kernels per stage): algorithmic bytes of SURVEY 8(d) over the stage time
            alg = ALG_BYTES_PER_CELL_STAGE * local_zones
            stage_ms = 1.0e3 * elapsed / args.steps / 2.0
            traffic, traffic_src = (measured_traffic("disk_sph_") if dims == (256, 128, 128) else (None, None))
            out["roofline"] = {"bound": "hbm", "achieved": alg / (stage_ms * 1.0e-3) / 1.0e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": alg / (stage_ms * 1.0e-3) / 1.0e9 / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_source": traffic_src,
                               "kernel": ("whole stage: %s + boundary conditions"
                                          % ("viscous_source_kernel (ZeroDiffusionFlux + ViscousFlux + DiffusionUpdate's sums, one tile march) + "
                                             "stage_curv_kernel (fluxes, update, sources, ConsToPrim, dt in one tile march, geometry in LDS tables)"
                                             if sim.stage_kernel == "stage_curv_kernel" else
                                             ("viscous pre-pass + viscous-flux pass + stage_fused_kernel<curvilinear>"
                                              if sim.stage_kernel.startswith("stage_fused_kernel") else
                                              "3 flux kernels + epilogue + PrimToCons (per-task chain)"))),
                               "launch_ms": stage_ms, "launches_timed": 2 * args.steps, "algorithmic_bytes_per_launch": alg}
        elif args.workload in ("disk_sph_smr", "disk_amr"):
            smr = args.workload == "disk_sph_smr"
            levels = sorted(set(sim.block_level(b) for b in range(sim.nblocks)))
            out["metric"] = ("cell-updates/sec (zone-cycles/s), spherical-polar alpha disk with static refinement" if smr else
                             "cell-updates/sec (zone-cycles/s), cylindrical disk + planet + dust, 4-level adaptive refinement")
            out["config"]["workload"] = (
                ("BASELINE configs[3]'s combination on %d GPU(s): inputs/disk/disk_sph.in x 2 (256 x 128 x %d root, 32^3 blocks) + a "
                 "level-1 static region around the midplane, gas, point-mass gravity, alpha viscosity, rotating frame, ic "
                 "conditions, HLLE + PLM_G, rk2; %d blocks, %d zones" % (args.gpus, 128 * args.gpus, nblocks_global, total_zones)) if smr else
                ("BASELINE configs[4]'s combination on %d GPU(s), 3-D: inputs/disk/disk_nbody_cyl.in (128 x %d x %d root over |z| < 0.2, "
                 "%d^3 blocks) + planet (N-body gravity, integrator none) + one dust species with simple_dust drag + rotating "
                 "frame + alpha viscosity + adaptive refinement on the pressure gradient, numlevel 4; %d blocks on levels (rank 0) %s, "
                 "%d zones at the end, %d remeshes in the timed region" % (args.gpus, 128 * args.gpus, max(16, args.amr_block), args.amr_block, nblocks_global, levels, total_zones, int(sim_remeshes_timed))))
            out["config"]["decomposition"] = (
                "%d rank(s): the Z-ordered leaf list (%d blocks) cut into consecutive runs of equal cost, one per rank; the root mesh "
                "grows with the rank count along the azimuth (%s) so that zones per GPU stay fixed; ghost zones, coarse-buffer "
                "fills and flux corrections between runs go through %s, one message per peer and phase; remeshes migrate whole "
                "blocks" % (args.gpus, nblocks_global, "x3" if smr else "x2",
                            "the native C++ RCCL transport" if comm is not None else "device copies (one rank)"))
            out["config"]["blocks_global"] = nblocks_global
            out["config"]["blocks_rank0"] = sim.nblocks
            out["config"]["zones_per_rank"] = rank_zones
            out["config"]["load_balance_max_over_mean"] = load_balance
            out["config"]["loopback"] = bool(args.loopback)
            out["config"]["rank_grid"] = None  # (Z-order runs, not a brick grid)
            out["config"]["stage_path"] = sim.stage_kernel
            bps = ALG_BYTES_PER_CELL_STAGE if smr else 8.0 * 5.0 * (6 + 4)  # 8 B * 5 * (6 ns_gas + 4 ns_dust)
            stage_ms = 1.0e3 * elapsed / args.steps / 2.0
            alg = bps * (total_zones if smr else zone_cycles / args.steps)
            # measured HBM bytes per stage over EVERY kernel of the run (scripts/pmc_traffic.py --workload ...; hash-matched)
            traffic, traffic_src = measured_traffic(args.workload + "_") if fused else (None, None)
            out["roofline"] = {"bound": "hbm", "achieved": alg / (stage_ms * 1.0e-3) / 1.0e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": alg / (stage_ms * 1.0e-3) / 1.0e9 / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_source": traffic_src,
                               "kernel": "whole stage (stage kernels, diffusion fluxes, flux correction, block-graph exchange, "
                                         "conditions; per cycle also the timestep%s)" % ("" if smr else ", tagging and remeshes"),
                               "launch_ms": stage_ms, "launches_timed": 2 * args.steps, "algorithmic_bytes_per_launch": alg}
            if args.remesh_in_timed_region:  # (with or without the leg that follows the timed region)
                ev = [e for e in timed_events if e.get("created", 0) + e.get("destroyed", 0) > 0]
                big = [e for e in ev if e["created"] + e["destroyed"] >= 0.02 * e["leaves_before"]]
                cyc_ms = 1.0e3 * elapsed / args.steps
                out["remesh_in_timed_region"] = {
                    "events": ev, "remeshes": len(ev), "remeshes_changing_2pct_of_leaves": len(big),
                    "ms_mean_of_those": (sum(e["ms"] for e in big) / len(big)) if big else None,
                    "cycle_ms_with_them": cyc_ms,
                    "what": "every fourth cycle of the timed region the workload's own criterion is injected at a lowered "
                            "threshold (artemis_sim_inject_refine_tags on ~2.5 % of the leaves); `value` counts the zones of "
                            "every cycle and its time includes these remeshes"}
            if remesh_leg:
                out["remesh"] = remesh_leg
                out["remesh_ms_mean"] = remesh_leg["ms_mean"]
                out["device_bytes_peak"] = remesh_leg["device_bytes_peak_during_remesh"]
        elif args.workload == "linwave3d":
            out["metric"] = "cell-updates/sec (zone-cycles/s), %d^3 linear wave, %s + HLLC" % (args.n, args.recon.upper())
            out["config"]["workload"] = ("inputs/linwave/linear_wave.in in 3-D at %d^3 (one block, periodic), gas, HLLC + %s, rk2, "
                                         "cfl 0.9: what a deck outside the tile marches' coverage runs on" % (args.n, args.recon.upper()))
            out["config"]["decomposition"] = "1 rank, one mesh block"
            out["config"]["stage_path"] = sim.stage_kernel
            if nlaunch:
                alg = ALG_BYTES_PER_CELL_STAGE * local_zones
                achieved = alg / (kms * 1.0e-3) / 1.0e9
                out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                   "traffic": None, "kernel": sim.stage_kernel, "launch_ms": kms, "launches_timed": nlaunch,
                                   "algorithmic_bytes_per_launch": alg}
        elif args.workload == "ssheet_dust":
            # B_alg per cell-stage = 8 B * 5 * (6 ns_gas + 4 ns_dust); one "launch" = one stage
            # of the general fused path (gas kernel + dust kernel + drag/aux/c2p finish)
            bps = 8.0 * 5.0 * (6 + 4 * args.dust)
            out["metric"] = "cell-updates/sec (zone-cycles/s), %d^2 dusty shearing sheet" % args.n
            out["config"]["workload"] = ("SURVEY config 3: inputs/ssheet strat problem, Cartesian %d^2, gas + %d dust "
                                         "species, simple_dust drag, shearing box, point-mass gravity, extrap/inflow "
                                         "BCs, HLLC gas / HLLE dust + PLM, rk2" % (args.n, args.dust))
            out["config"]["decomposition"] = "1 rank, one mesh block"
            if fused and nlaunch:
                alg = bps * local_zones
                achieved = alg / (kms * 1.0e-3) / 1.0e9
                kname = sim.stage_kernel
                traffic, traffic_src = measured_traffic(("cfg3_" if args.n == 4096 else "cfg3_%d_" % args.n) +
                                                        ("" if args.dust == 1 else "%ddust_" % args.dust))
                out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                                   "kernel": ("stage2d_kernel: gas + dust fluxes, update, sources, drag, aux, ConsToPrim, dt in one "
                                              "launch per stage (2-D row march)") if kname == "stage2d_kernel" else
                                             "general fused stage: stage_cell_kernel<gas> + <dust> + simple_drag_kernel<finish>",
                                   "launch_ms": kms, "launches_timed": nlaunch, "algorithmic_bytes_per_launch": alg}
        elif args.workload == "sedov3d" and fused and nlaunch:
            alg = ALG_BYTES_PER_CELL_STAGE * local_zones  # bytes per launch (one stage, one rank)
            achieved = alg / (kms * 1.0e-3) / 1.0e9
            # PMC counters cannot be read from inside the process: scripts/pmc_traffic.py measures them
            # (separate rocprofv3 --pmc passes of this very command) and tags the record with the hash of
            # the kernel sources; a record measured on other sources is not reported.
            traffic, traffic_src = (measured_traffic("") if args.n == 256 else (None, None))
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_source": traffic_src,
                               "kernel": "stage_fused_kernel<hllc,plm>", "launch_ms": kms,
                               "launches_timed": nlaunch,
                               "algorithmic_bytes_per_launch": alg}
            # The kernel is bound by fp64 instruction issue, not by HBM: quote that roofline too.  floor = VALU
            # wave-instructions per launch (SQ_INSTS_VALU, profiles/) x 4 issue cycles / (1024 SIMDs x 2.4 GHz).
            valu, valu_src = (measured_valu() if args.n == 256 else (None, None))
            if valu:
                floor_ms = valu * 4.0 / (1024.0 * 2.4e9) * 1.0e3
                out["roofline"]["fp64_issue"] = {"bound": "valu", "valu_wave_instructions_per_launch": valu,
                                                 "lane_instructions_per_zone_stage": valu * 64.0 / local_zones,
                                                 "issue_floor_ms": floor_ms, "frac": floor_ms / kms, "source": valu_src,
                                                 "peak": "1024 SIMDs x 16 fp64 lanes x 2.4 GHz (MI355X_MICROARCH.md)"}
            if dropin:
                dropin["fused"] = value
                dropin["frac_fused"] = value * 2.0 * ALG_BYTES_PER_CELL_STAGE / 1.0e9 / HBM_PEAK_GBS
                out["dropin"] = dropin
        if args.workload == "linwave3d":
            pass  # (a gap measurement: no CPU leg)
        elif args.workload in ("disk_sph", "disk_sph_smr", "disk_amr") and not args.no_cpu_baseline:
            hc = host_cores()
            threads = args.cpu_threads or hc["physical_usable"]
            kind = "cyl_dust" if args.workload == "disk_amr" else "sph"
            v, secs, cyc, cnx = cpu_baseline_disk(kind, 3, threads)
            out["cpu_baseline"] = {
                "value": v, "unit": "zone-cycles/s", "cores": threads, "kind": "port", "threads": threads, "host": hc,
                "sample": "CPU oracle (C++ restatement of the reference's CPU path, OpenMP) on %d threads: %s at %d x %d x %d "
                          "(uniform mesh), %d cycles in %.1f s"
                          % (threads, "the cylindrical disk + planet + dust + drag + viscosity of configs[4]" if kind == "cyl_dust"
                             else "inputs/disk/disk_sph.in", cnx[0], cnx[1], cnx[2], cyc, secs)}
        elif args.workload in ("disk_sph", "disk_sph_smr", "disk_amr"):
            pass
        elif args.workload == "ssheet_dust" and not args.no_cpu_baseline:
            hc = host_cores()
            threads = args.cpu_threads or hc["physical_usable"]
            v, secs, cyc, cn = cpu_baseline_ssheet(min(args.n, 512), args.dust, args.cpu_cycles, threads)
            out["cpu_baseline"] = {
                "value": v, "unit": "zone-cycles/s", "cores": threads, "kind": "port", "threads": threads, "host": hc,
                "sample": "CPU oracle (C++ restatement of the reference's CPU path, OpenMP over rows) on %d threads (the "
                          "container's CPU quota, else one per physical core), same problem at %d^2, %d cycles, %.1f s"
                          % (threads, cn, cyc, secs)}
        elif args.gpus == 1 and not args.no_cpu_baseline:
            hc = host_cores()
            threads = args.cpu_threads or hc["physical_usable"]
            v, secs, cyc = cpu_baseline(args.cpu_n, args.cpu_cycles, threads)
            v1, secs1, cyc1 = cpu_baseline(128, 2, 1)  # the same code on ONE core (128^3: ~4 s)
            out["cpu_baseline"] = {
                "value": v, "unit": "zone-cycles/s", "cores": threads, "kind": "port",
                "threads": threads, "per_core": v / threads, "one_thread": v1, "host": hc,
                "parallel_efficiency": v / threads / v1,
                "sample": "CPU oracle (C++ restatement of the reference's CPU path, NOT the Artemis executable; OpenMP over "
                          "k-j rows, arrays first-touched by the sweeping threads, OMP_PROC_BIND=%s OMP_PLACES=%s): Sedov %d^3, "
                          "%d cycles in %.1f s on %d threads (= the container's CPU quota, else one per physical core; %d logical CPUs); one thread: Sedov 128^3, "
                          "%d cycles in %.1f s.  Thread scaling of this restatement on this host: profiles/r02_cpu_scaling.txt"
                          % (os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES"), args.cpu_n, cyc, secs, threads,
                             hc["logical"], cyc1, secs1)}
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    sim.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
