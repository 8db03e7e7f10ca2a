#!/usr/bin/env python3
"""Thread scaling of the CPU oracle on this host (cpu_baseline context for bench.py): Sedov n^3, a few
cycles per thread count.  Prints one JSON line per count.  Run on the GPU box: the numbers qualify the
`cpu_baseline` object of the bench line (cores actually usable, where the restatement stops scaling)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")


def main():
    from bench import cpu_baseline, host_cores
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    info = host_cores()
    print(json.dumps({"host": info}), flush=True)
    t = 1
    while t <= info["logical"]:
        v, secs, cyc = cpu_baseline(n, 2 if t < 8 else 4, t)
        print(json.dumps({"n": n, "threads": t, "zone_cycles_per_s": v, "per_thread": v / t, "seconds": secs}), flush=True)
        t *= 2


if __name__ == "__main__":
    main()
