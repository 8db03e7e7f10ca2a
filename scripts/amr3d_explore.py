import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import amr_cases
from artemis_amd.driver import Simulation
import collections
for n, nz, thr, planet, dc, h0, rlim, zp in [(16, 8, 1.5, 3e-2, 3, 0.2, (0.5, 2.5), 0.08), (16, 8, 1.5, 6e-2, 2, 0.2, (0.5, 2.5), 0.08),
                                         (16, 8, 1.3, 3e-2, 1, 0.2, (0.5, 2.5), 0.08), (16, 8, 1.2, 2e-2, 2, 0.2, (0.5, 2.5), 0.12),
                                         (16, 8, 1.5, 3e-2, 3, 0.2, (0.5, 2.5), 0.0)]:
    case = amr_cases.disk_planet_dust_amr(n=n, planet=planet, thr=thr, nz=nz, zlim=0.2, derefine_count=dc, h0=h0, rlim=rlim, nphi=32, zp=zp)
    try:
        s = Simulation(amr_cases.DECK(*case["deck"]), case["overrides"])
    except Exception as e:
        print(n, nz, thr, planet, "failed", repr(e)[:200]); continue
    r0 = s.remeshes
    hist = []
    for c in range(80):
        s.evolve(1)
        hist.append((s.nblocks, s.remeshes - r0))
    lv = collections.Counter(s.block_level(b) for b in range(s.nblocks))
    off = collections.Counter()
    for b in range(s.nblocks):
        bb = s.block_bounds(b)
        zc = 0.5 * (bb[4] + bb[5]); rc = 0.5 * (bb[0] + bb[1]); pc = 0.5 * (bb[2] + bb[3])
        near = abs(rc - 1) < 0.25 and abs(pc) < 0.4
        off[(s.block_level(b), "planet" if near else ("high" if abs(zc) > 0.1 else "mid"))] += 1
    print("   where", sorted(off.items()))
    print("n", n, "nz", nz, "thr", thr, "planet", planet, "dc", dc, "h0", h0, rlim, "zp", zp, "blocks", hist[0][0], "->", s.nblocks, "levels", dict(lv), "remeshes", s.remeshes - r0,
          "trace", [(i, h[0]) for i, h in enumerate(hist) if i == 0 or h[0] != hist[i - 1][0] or h[1] != hist[i - 1][1]], flush=True)
    s.close()
