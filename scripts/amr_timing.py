#!/usr/bin/env python3
"""Throughput of BASELINE configs[4]'s combination on one MI355X, zone-cycles/s over a few cycles (remeshes included):
inputs/disk/disk_nbody_cyl.in (cylindrical disk, N-body gravity, alpha viscosity, `ic` conditions) + a planet on a
circular orbit + one dust species with drag + the rotating frame + adaptive refinement to FOUR levels on the pressure
gradient (tests/amr_cases.py runs the 2-D form of it against the adaptive oracle).
    python scripts/amr_timing.py [cycles] [root nx1 nx2 nx3] [block] [overrides ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from artemis_amd.driver import Simulation
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    nx = [int(v) for v in sys.argv[2:5]] if len(sys.argv) > 4 else [128, 128, 32]
    mb = int(sys.argv[5]) if len(sys.argv) > 5 else 16
    extra = sys.argv[6:]
    ov = ["parthenon/mesh/nx1=%d" % nx[0], "parthenon/mesh/nx2=%d" % nx[1], "parthenon/mesh/nx3=%d" % nx[2],
          "parthenon/meshblock/nx1=%d" % mb, "parthenon/meshblock/nx2=%d" % mb, "parthenon/meshblock/nx3=%d" % mb,
          "parthenon/mesh/refinement=adaptive", "parthenon/mesh/numlevel=4", "parthenon/mesh/derefine_count=5",
          "gas/refine_field=pressure", "gas/refine_type=gradient", "gas/refine_thr=0.8",
          "physics/rotating_frame=true", "rotating_frame/omega=1.0",
          "physics/dust=true", "dust/nspecies=1", "dust/cfl=0.3", "dust/reconstruct=plm", "dust/riemann=hlle",
          "dust/dfloor=1e-10", "physics/drag=true", "drag/type=simple_dust", "dust/stopping_time/type=constant",
          "dust/stopping_time/tau=0.1", "dust/sizes=1.0",
          "nbody/particle2/mass=1.0e-2", "nbody/particle2/couple=1", "nbody/particle2/soft/type=plummer",
          "nbody/particle2/soft/radius=0.03", "nbody/particle2/initialize/x=1.0", "nbody/particle2/initialize/vy=1.0",
          "parthenon/time/nlim=-1"] + extra
    t0 = time.perf_counter()
    s = Simulation(os.path.join(ROOT, "inputs", "disk", "disk_nbody_cyl.in"), ov)
    t1 = time.perf_counter()
    s.evolve(3)
    torch.cuda.synchronize()
    r0, t2, zc = s.remeshes, time.perf_counter(), 0
    for _ in range(n):
        zc += s.total_zones
        s.evolve(1)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    lv = [s.block_level(b) for b in range(s.nblocks)]
    rs = s.remesh_seconds()
    print("remesh: n %d total %.3f s (build %.3f, hand-over %.3f; tagging over all cycles %.3f) device bytes now / peak %.2f / %.2f GB"
          % (rs[0], rs[1], rs[2], rs[3], rs[4], s.device_bytes()[0] / 1e9, s.device_bytes()[1] / 1e9), flush=True)
    print("blocks", s.nblocks, "levels", sorted(set(lv)), "zones", s.total_zones, "setup %.2f s" % (t1 - t0), "cycles", n,
          "remeshes", s.remeshes - r0, "%.2f ms/cycle" % (1e3 * (t3 - t2) / n), "%.3e zone-cycles/s" % (zc / (t3 - t2)),
          "kernel", s.stage_kernel, "dt", s.dt, flush=True)
    s.close()


if __name__ == "__main__":
    main()
