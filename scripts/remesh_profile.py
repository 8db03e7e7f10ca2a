"""Development aid: host-side phases of a batched remesh on the configs[4] bench mesh (SETUP_TIMING prints them)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from artemis_amd import capi
from artemis_amd.driver import Simulation
import torch
deck = os.path.join(ROOT, "inputs", "disk", "disk_nbody_cyl.in")
mb = 16
ov = ["parthenon/mesh/nx1=128", "parthenon/mesh/nx2=128", "parthenon/mesh/nx3=16", "parthenon/mesh/x3min=-0.2", "parthenon/mesh/x3max=0.2",
      "parthenon/meshblock/nx1=%d" % mb, "parthenon/meshblock/nx2=%d" % mb, "parthenon/meshblock/nx3=%d" % mb,
      "parthenon/mesh/refinement=adaptive", "parthenon/mesh/numlevel=4", "parthenon/mesh/derefine_count=5",
      "gas/refine_field=pressure", "gas/refine_type=gradient", "gas/refine_thr=2.0", "physics/rotating_frame=true", "rotating_frame/omega=1.0",
      "physics/dust=true", "dust/nspecies=1", "dust/cfl=0.3", "dust/reconstruct=plm", "dust/riemann=hlle", "dust/dfloor=1e-10",
      "physics/drag=true", "drag/type=simple_dust", "dust/stopping_time/type=constant", "dust/stopping_time/tau=0.1", "dust/sizes=1.0",
      "nbody/particle2/mass=1.0e-2", "nbody/particle2/couple=1", "nbody/particle2/soft/type=plummer", "nbody/particle2/soft/radius=0.03",
      "nbody/particle2/initialize/x=1.0", "nbody/particle2/initialize/vy=1.0", "parthenon/time/nlim=-1"]
s = Simulation(deck, ov)
s.evolve(2)
torch.cuda.synchronize()
capi.load().artemis_hip_set_option(b"setup_timing", 1)
for rep in range(2):
    t = time.time()
    ch = s.inject_refine_tags(bench.batch_of_leaves(s))
    torch.cuda.synchronize()
    print("remesh", rep, ch, "%.1f ms" % (1e3 * (time.time() - t)), s.last_remesh(), "bytes", s.device_bytes(), flush=True)
    s.evolve(1)
