#!/usr/bin/env python3
"""How fast is the tuned fused kernel on a 2-D gas problem (one plane per workgroup, no x3 march)?
Reference point for moving config 3's gas fluid from the cell-centred general stage onto it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from artemis_amd.driver import Simulation
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ov = [f"parthenon/mesh/nx1={n}", f"parthenon/mesh/nx2={n}", f"parthenon/meshblock/nx1={n}", f"parthenon/meshblock/nx2={n}",
          "gas/riemann=hllc", "problem/radius=0.1", "problem/samples=0", "parthenon/time/tlim=-1.0", "parthenon/time/nlim=-1"]
    for path in ("fused", "general"):
        if path == "general":
            from artemis_amd import capi
            capi.load().artemis_hip_set_option(b"no_tuned", 1)
        s = Simulation(os.path.join(ROOT, "inputs", "blast", "blast.in"), ov)
        s.evolve(5)
        s.set_kernel_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = s.evolve(40)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, nl = s.kernel_ms()
        print(path, s.stage_kernel, "%.3e zc/s" % (n * n * k / dt), "kernel ms %.3f (%d launches)" % (ms, nl), flush=True)
        s.close()


if __name__ == "__main__":
    main()
