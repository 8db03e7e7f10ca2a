import sys, time, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artemis_amd.driver import Simulation
N = sys.argv[2] if len(sys.argv) > 2 else "1024"
ov = ["parthenon/mesh/nx1="+N,"parthenon/mesh/nx2="+N,"parthenon/meshblock/nx1="+N,"parthenon/meshblock/nx2="+N,
      "physics/dust=true","physics/drag=true","dust/nspecies=%s" % sys.argv[1],"dust/cfl=0.3","dust/reconstruct=plm","dust/riemann=hlle",
      "dust/dfloor=1.0e-10","dust/stopping_time/type=constant","dust/stopping_time/tau=" + ",".join(["0.1"]*int(sys.argv[1])),
      "drag/type=simple_dust","parthenon/time/nlim=60"]
if "--sync" not in sys.argv:
    ov.append("parthenon/time/tlim=-1.0")  # no time limit: dt stays on the device, no per-step sync
sys.argv = [a for a in sys.argv if a != "--sync"]
s = Simulation(os.path.join(ROOT, "inputs", "ssheet", "ssheet.in"), ov)
if len(sys.argv) > 3: s.set_path(sys.argv[3])
s.evolve(10)
import torch; torch.cuda.synchronize()
t=time.time(); n=s.evolve(50); torch.cuda.synchronize(); w=time.time()-t
print("ns_dust",sys.argv[1],"cycles",n,"wall",w,"zc/s",int(N)**2*n/w, "fused", s.uses_fused_path)
