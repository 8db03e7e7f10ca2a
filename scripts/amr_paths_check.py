#!/usr/bin/env python3
"""BASELINE configs[4]'s combination at bench size (128 x 128 x 16 root, 16^3 blocks, four levels: ~7000 blocks, 29 M zones)
on the one-kernel stages against the per-task chain: same mesh, same dt, every leaf of both fluids equal bit for bit after
a few cycles; the particle forces to round-off (their sums are formed in a different order on the two paths).
    python scripts/amr_paths_check.py [cycles]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from artemis_amd.driver import Simulation
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    ov = ["parthenon/mesh/nx1=128", "parthenon/mesh/nx2=128", "parthenon/mesh/nx3=16", "parthenon/meshblock/nx1=16",
          "parthenon/meshblock/nx2=16", "parthenon/meshblock/nx3=16", "parthenon/mesh/x3min=-0.2", "parthenon/mesh/x3max=0.2",
          "parthenon/mesh/refinement=adaptive", "parthenon/mesh/numlevel=4", "parthenon/mesh/derefine_count=5",
          "gas/refine_field=pressure", "gas/refine_type=gradient", "gas/refine_thr=2.0",
          "physics/rotating_frame=true", "rotating_frame/omega=1.0",
          "physics/dust=true", "dust/nspecies=1", "dust/cfl=0.3", "dust/reconstruct=plm", "dust/riemann=hlle",
          "dust/dfloor=1e-10", "physics/drag=true", "drag/type=simple_dust", "dust/stopping_time/type=constant",
          "dust/stopping_time/tau=0.1", "dust/sizes=1.0",
          "nbody/particle2/mass=1.0e-2", "nbody/particle2/couple=1", "nbody/particle2/soft/type=plummer",
          "nbody/particle2/soft/radius=0.03", "nbody/particle2/initialize/x=1.0", "nbody/particle2/initialize/vy=1.0",
          "parthenon/time/nlim=-1"]
    deck = os.path.join(ROOT, "inputs", "disk", "disk_nbody_cyl.in")
    a, b = Simulation(deck, ov), Simulation(deck, ov)
    b.set_path("unfused")
    assert a.uses_fused_path and not b.uses_fused_path and a.nblocks == b.nblocks
    a.evolve(n), b.evolve(n)
    assert a.nblocks == b.nblocks and a.dt == b.dt and a.time == b.time and a.ncycle == b.ncycle == n
    bad = 0
    for blk in range(a.nblocks):
        for f in ("gas.prim", "dust.prim"):
            x, y = a.interior(a.field(f, blk)), b.interior(b.field(f, blk))
            if f == "gas.prim":
                x, y = x[[0, 1, 2, 3, 5]], y[[0, 1, 2, 3, 5]]
            bad += int(not np.array_equal(x, y))
    fa, fb = a.nbody_force(), b.nbody_force()
    rel = np.abs(fa - fb).max() / max(np.abs(fb).max(), 1e-300)
    print("blocks %d zones %d cycles %d kernel [%s] vs [%s]: dt equal, %d of %d leaf arrays differ, particle forces rel. diff %.2e"
          % (a.nblocks, a.total_zones, n, a.stage_kernel, b.stage_kernel, bad, 2 * a.nblocks, rel))
    assert bad == 0 and rel < 1e-11
    a.close(), b.close()


if __name__ == "__main__":
    main()
