"""The headline deck (256^3 Sedov, HLLC + PLM, rk2, tuned kernel) to t = 0.1 on one GPU: conservation and the
shock radius against the Sedov-Taylor similarity solution."""
import sys, time, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artemis_amd.driver import Simulation
big = ["parthenon/mesh/nx1=256", "parthenon/mesh/nx2=256", "parthenon/mesh/nx3=256", "parthenon/mesh/x3min=-1.0",
       "parthenon/mesh/x3max=1.0", "parthenon/meshblock/nx1=256", "parthenon/meshblock/nx2=256",
       "parthenon/meshblock/nx3=256", "gas/riemann=hllc", "problem/symmetry=spherical", "problem/radius=0.03",
       "problem/samples=0", "parthenon/time/tlim=0.1"]
f = Simulation(os.path.join(ROOT, "inputs", "blast", "blast.in"), big)
h0 = f.history()
t = time.time(); f.evolve(); w = time.time() - t
h1 = f.history()
P = f.interior(f.field("gas.prim"))
x = -1 + (np.arange(256) + 0.5) / 128
line = P[4, 128, 128, :]  # pressure along +x through the centre (cells 128.. are x>0)
rs = abs(x[np.argmax(line)])
E = h0[4]
print("cycles", f.ncycle, "wall %.1f" % w, "time", f.time, "E0", E, "dE/E", (h1[4] - h0[4]) / h0[4], "dM/M", (h1[0] - h0[0]) / h0[0])
print("shock radius", rs, "Sedov", 1.033 * (E * f.time ** 2) ** 0.2)
