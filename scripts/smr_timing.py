#!/usr/bin/env python3
"""Throughput of a statically refined disk deck on one MI355X, zone-cycles/s over a few cycles:
    python scripts/smr_timing.py [cycles] [overrides ...]         inputs/disk/disk_cart.in as shipped (808 coarse +
                                                                  1728 fine blocks of 16 x 16 x 8, nghost 4)
    python scripts/smr_timing.py [cycles] sph [overrides ...]     inputs/disk/disk_sph.in x 2 (256 x 128 x 128 root in
        32^3 blocks) with a level-1 region around the midplane between r = 0.7 and 1.9 -- BASELINE configs[3]'s
        combination (spherical-polar disk + static refinement)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from artemis_amd.driver import Simulation
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    t0 = time.perf_counter()
    extra = sys.argv[2:]
    deck = "disk_cart.in"
    if extra and extra[0] == "sph":
        deck, extra = "disk_sph.in", extra[1:]
        extra = ["parthenon/mesh/nx1=256", "parthenon/mesh/nx2=128", "parthenon/mesh/nx3=128", "parthenon/mesh/refinement=static",
                 "parthenon/static_refinement1/level=1", "parthenon/static_refinement1/x1min=0.7",
                 "parthenon/static_refinement1/x1max=1.9", "parthenon/static_refinement1/x2min=1.3",
                 "parthenon/static_refinement1/x2max=1.85", "parthenon/static_refinement1/x3min=-3.0",
                 "parthenon/static_refinement1/x3max=3.0"] + extra
    s = Simulation(os.path.join(ROOT, "inputs", "disk", deck), ["parthenon/time/nlim=100000"] + extra)
    t1 = time.perf_counter()
    s.evolve(3)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    k = s.evolve(n)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("blocks", s.nblocks, "zones", s.total_zones, "setup %.2f s" % (t1 - t0), "cycles", k,
          "%.2f ms/cycle" % (1e3 * (t3 - t2) / k), "%.3e zone-cycles/s" % (s.total_zones * k / (t3 - t2)), "dt", s.dt, flush=True)
    s.close()


if __name__ == "__main__":
    main()
