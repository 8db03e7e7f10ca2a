"""The shipped disk decks to their own tlim = 62.8 (ten orbits at R = 1) on the GPU: how far the density
drifts from the initial equilibrium (the measure of tst/scripts/disk/disk.py, which stops after 10 cycles)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artemis_amd.driver import Simulation
for g in sys.argv[1:] or ["axi", "cyl", "sph"]:
    s = Simulation(os.path.join(ROOT, "inputs", "disk", f"disk_{g}.in"), [])
    d0 = [s.interior(s.field("gas.prim", b))[0].copy() for b in range(s.nblocks)]
    t = time.time()
    s.evolve()
    w = time.time() - t
    num = sum((a * (s.interior(s.field("gas.prim", b))[0] - a) ** 2).sum() for b, a in enumerate(d0))
    den = sum(a.sum() for a in d0)
    P = np.concatenate([s.interior(s.field("gas.prim", b)).reshape(6, -1) for b in range(s.nblocks)], axis=1)
    print("disk_%s: blocks %d, t = %.3f, cycles %d, wall %.1f s, density error %.3e, finite %s, min rho %.2e, min sie %.2e"
          % (g, s.nblocks, s.time, s.ncycle, w, np.sqrt(num) / den, bool(np.isfinite(P).all()), P[0].min(), P[5].min()), flush=True)
