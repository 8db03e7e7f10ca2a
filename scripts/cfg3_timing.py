import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from artemis_amd.driver import Simulation
ov = ["parthenon/mesh/nx1=1024","parthenon/mesh/nx2=1024","parthenon/meshblock/nx1=1024","parthenon/meshblock/nx2=1024",
      "physics/dust=true","physics/drag=true","dust/nspecies=%s" % sys.argv[1],"dust/cfl=0.3","dust/reconstruct=plm","dust/riemann=hlle",
      "dust/dfloor=1.0e-10","dust/stopping_time/type=constant","dust/stopping_time/tau=" + ",".join(["0.1"]*int(sys.argv[1])),
      "drag/type=simple_dust","parthenon/time/nlim=60"]
s = Simulation("/root/repo/inputs/ssheet/ssheet.in", ov)
s.evolve(10)
import torch; torch.cuda.synchronize()
t=time.time(); n=s.evolve(50); torch.cuda.synchronize(); w=time.time()-t
print("ns_dust",sys.argv[1],"cycles",n,"wall",w,"zc/s",1024*1024*n/w, "fused", s.uses_fused_path)
