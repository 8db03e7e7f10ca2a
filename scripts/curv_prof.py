"""Development aid: phase clocks of the curvilinear march (library built with ARTEMIS_HIPFLAGS_KERNELS_CURV=-DCURV_PROF).
python scripts/curv_prof.py [disk_sph|blast_sph|...]: runs scripts/curv_timing.py's deck and prints the share of wave
cycles per phase of a plane."""
import ctypes as C, os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artemis_amd import capi
L = capi.load()
buf = (C.c_ulonglong * 16)()
sys.argv = ["curv_timing.py"] + sys.argv[1:]
import torch
torch.cuda.synchronize()
which = os.environ.get("PROF_KERNEL", "curv")
f = getattr(L, "artemis_hip_debug_%s_prof" % which)
f(buf, 1)
runpy.run_path(os.path.join(ROOT, "scripts", "curv_timing.py"), run_name="__main__")
torch.cuda.synchronize()
f(buf, 0)
v = [buf[i] for i in range(10)]
tot = float(sum(v)) or 1.0
vs_names = ["loop top + deferred stores", "issue the trip's loads", "(a) stage plane k+1", "barrier 1", "(c) x1 / x2 faces + duty", "(b) divergence, viscosity", "x3 face", "barrier 2", "(d) sums of zone k", "tail"]
names = vs_names if which == "vs" else ["loads/loop top", "P1 work", "barrier 1", "P2 work + stage next", "barrier 2", "fold x1 x2", "x3 sweep", "fold x3 + update", "tail", "qnn + wait + u1/ds issue"]
for n, x in zip(names, v):
    print("%-22s %6.2f %%  %.3e" % (n, 100.0 * x / tot, x))
