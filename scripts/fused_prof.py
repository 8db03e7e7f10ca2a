"""Development aid: phase clocks of the tuned Cartesian march (library built with ARTEMIS_HIPFLAGS_KERNELS_FUSED=-DFUSED_PROF)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artemis_amd import capi
from artemis_amd.driver import Simulation
import torch, time
import bench
L = capi.load()
buf = (C.c_ulonglong * 16)()
s = Simulation(os.path.join(ROOT, "inputs", "blast", "blast.in"), bench.overrides(1, (256, 256, 256), 100))
s.evolve(5)
torch.cuda.synchronize()
L.artemis_hip_debug_fused_prof(buf, 1)
t = time.time(); n = s.evolve(30); torch.cuda.synchronize(); w = time.time() - t
L.artemis_hip_debug_fused_prof(buf, 0)
print("zc/s %.4e" % (s.total_zones * n / w))
v = [buf[i] for i in range(10)]
tot = float(sum(v)) or 1.0
names = ["loads/loop top", "P1 work", "barrier 1", "stage next (waits for the halo loads)", "barrier 2", "wait for the prefetch (x3 start)", "x3 sweep", "update", "tail", "P2 work: Riemann + duty"]
for n_, x in zip(names, v):
    print("%-34s %6.2f %%  %.3e" % (n_, 100.0 * x / tot, x))
