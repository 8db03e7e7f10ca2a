#!/usr/bin/env python3
"""Throughput of the statically refined Cartesian disk deck (inputs/disk/disk_cart.in as shipped: 808 coarse +
1728 fine blocks of 16 x 16 x 8, nghost 4, alpha viscosity) on one MI355X: zone-cycles/s over a few cycles."""
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
    s = Simulation(os.path.join(ROOT, "inputs", "disk", "disk_cart.in"), ["parthenon/time/nlim=100000"] + sys.argv[2:])
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
