"""Throughput of the disk deck (config 4 without refinement) through the driver's per-task chain:
python scripts/disk_timing.py [sph|cyl|axi] [scale] [fused|unfused]  (scale multiplies every active mesh
dimension; the path defaults to the driver's choice)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artemis_amd.driver import Simulation
g = sys.argv[1] if len(sys.argv) > 1 else "sph"
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 1
NX = {"axi": (128, 64, 1), "cyl": (128, 64, 32), "sph": (128, 64, 64)}[g]
nx = [n * scale if n > 1 else 1 for n in NX]
ov = ["parthenon/time/nlim=70"]
for d, n in enumerate(nx, 1):
    ov += [f"parthenon/mesh/nx{d}={n}", f"parthenon/meshblock/nx{d}={n}"]
s = Simulation(os.path.join(ROOT, "inputs", "disk", f"disk_{g}.in"), ov)
if len(sys.argv) > 3:
    s.set_path(sys.argv[3])
s.evolve(10)
import torch
torch.cuda.synchronize()
t = time.time(); n = s.evolve(50); torch.cuda.synchronize(); w = time.time() - t
cells = nx[0] * nx[1] * nx[2]
print("disk", g, nx, "cycles", n, "wall %.3f" % w, "zc/s %.4e" % (cells * n / w), "fused", s.uses_fused_path, "dt", s.dt)
