#!/usr/bin/env python3
"""Run-to-run determinism of the HIP driver on an adaptive deck: the same deck N times, a hash of dt + every leaf.
    python scripts/determinism_check.py [case] [runs] [cycles]      case: blast_amr | linear_wave_amr | disk_planet_dust_amr"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import amr_cases
    from artemis_amd.driver import Simulation
    name = sys.argv[1] if len(sys.argv) > 1 else "blast_amr"
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    kw = dict(n=128, derefine_count=5) if name == "blast_amr" else {}
    case = getattr(amr_cases, name)(**kw)
    seen = {}
    for r in range(runs):
        s = Simulation(amr_cases.DECK(*case["deck"]), case["overrides"])
        trail = []
        done = 0
        while done < cycles:
            done += s.evolve(min(20, cycles - done))
            h = hashlib.sha1()
            h.update(np.float64(s.dt).tobytes())
            for b in range(s.nblocks):
                h.update(np.ascontiguousarray(s.field("gas.prim", b)[[0, 1, 2, 3, 5]]).tobytes())
            trail.append(h.hexdigest()[:10])
        print(r, s.stage_kernel, s.nblocks, s.remeshes, repr(s.dt), " ".join(trail), flush=True)
        seen[tuple(trail)] = seen.get(tuple(trail), 0) + 1
        s.close()
    print("distinct outcomes:", len(seen), list(seen.values()))


if __name__ == "__main__":
    main()
