#!/usr/bin/env python3
"""Time artemis_hip_apply_bc on one 256^3 block for different sets of faces (microseconds per call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from artemis_amd.pack import MeshBlockPack
n = 256
mb = MeshBlockPack(1, (n, n, n), [(0, 0, 0)], [(1, 1, 1)], ng=2, ns_gas=1, ns_dust=0, reconstruct="plm", riemann="hllc", gamma=1.4,
                   dfloor=1e-10, siefloor=1e-10, with_fluxes=False)
mb.gas_prim.normal_()
for name, bc in (("all six outflow", ["outflow"] * 6), ("x1 only", ["outflow"] * 2 + ["none"] * 4), ("x2 only", ["none"] * 2 + ["outflow"] * 2 + ["none"] * 2),
                 ("x3 only", ["none"] * 4 + ["outflow"] * 2), ("x2 + x3", ["none"] * 2 + ["outflow"] * 4), ("all six reflecting", ["reflecting"] * 6)):
    for _ in range(5):
        mb.ApplyBoundaryConditions([bc])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        mb.ApplyBoundaryConditions([bc])
    torch.cuda.synchronize()
    print("%-20s %7.1f us" % (name, (time.perf_counter() - t0) / 200 * 1e6))
