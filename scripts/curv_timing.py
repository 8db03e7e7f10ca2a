"""Throughput of the curvilinear decks on one MI355X: python scripts/curv_timing.py [blast_sph|blast_cyl|disk_sph|disk_cyl|disk_axi] [fused|unfused]
blast_*: inputs/blast/blast.in on a 192 x 128 x 128 spherical-polar / cylindrical mesh (pure hydro, hlle + plm);
disk_*: scripts/disk_timing.py's decks (x2 / x8).  Prints zone-cycles/s and which kernel family ran."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artemis_amd.driver import Simulation
import torch
what = sys.argv[1] if len(sys.argv) > 1 else "blast_sph"
nx = (192, 128, 128)
if what.startswith("blast"):
    sph = what.endswith("sph")
    ov = ["artemis/coordinates=%s" % ("spherical" if sph else "cylindrical"),
          "parthenon/mesh/x1min=0.2", "parthenon/mesh/x1max=1.2",
          "parthenon/mesh/x2min=%s" % ("0.6" if sph else "0.0"), "parthenon/mesh/x2max=%s" % ("2.5" if sph else "6.0"),
          "parthenon/mesh/x3min=%s" % ("0.0" if sph else "-0.5"), "parthenon/mesh/x3max=%s" % ("6.0" if sph else "0.5"),
          "gas/riemann=hlle", "gas/reconstruct=plm", "problem/symmetry=spherical", "problem/samples=0",
          "parthenon/time/tlim=100.0", "parthenon/time/nlim=70"]
    deck = os.path.join(ROOT, "inputs", "blast", "blast.in")
else:
    g = what.split("_")[1]
    scale = 8 if g == "axi" else 2
    NX = {"axi": (128, 64, 1), "cyl": (128, 64, 32), "sph": (128, 64, 64)}[g]
    nx = tuple(n * scale if n > 1 else 1 for n in NX)
    ov = ["parthenon/time/nlim=70"]
    deck = os.path.join(ROOT, "inputs", "disk", "disk_%s.in" % g)
for d, n in enumerate(nx, 1):
    ov += ["parthenon/mesh/nx%d=%d" % (d, n), "parthenon/meshblock/nx%d=%d" % (d, n)]
s = Simulation(deck, ov)
if len(sys.argv) > 2:
    s.set_path(sys.argv[2])
s.evolve(10)
torch.cuda.synchronize()
t = time.time(); n = s.evolve(50); torch.cuda.synchronize(); w = time.time() - t
cells = nx[0] * nx[1] * nx[2]
print(what, nx, "cycles", n, "wall %.3f" % w, "zc/s %.4e" % (cells * n / max(w, 1e-9)), "kernel", s.stage_kernel, "dt", s.dt, flush=True)
