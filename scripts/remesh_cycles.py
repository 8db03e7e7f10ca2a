"""Development aid: wall time of the cycles that follow a batched remesh on the configs[4] bench mesh."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
from artemis_amd.driver import Simulation
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
s.evolve(3)
torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == "bench":  # bench.py --remesh-in-timed-region's pattern: tags every fourth cycle
    s.evolve(2)
    torch.cuda.synchronize()
    t00 = time.time()
    ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    if len(sys.argv) > 3:
        from artemis_amd import capi
        capi.load().artemis_hip_set_option(b"setup_timing", 1)
    for cyc in range(ncyc):
        t = time.time()
        what = ""
        if cyc % 4 == 3:
            s.inject_refine_tags(bench.batch_of_leaves(s))
            torch.cuda.synchronize()
            what = "inject %.1f ms %s; " % (1e3 * (time.time() - t), s.last_remesh()[0])
        t1 = time.time()
        r0 = s.remeshes
        s.evolve(1)
        torch.cuda.synchronize()
        print("cycle %2d: %scycle %.1f ms%s  blocks %d bytes %s" % (cyc, what, 1e3 * (time.time() - t1), "*" if s.remeshes != r0 else "", s.nblocks, s.device_bytes()), flush=True)
    print("total %.1f ms" % (1e3 * (time.time() - t00)))
    sys.exit(0)
for rep in range(3):
    t = time.time()
    leaves = bench.batch_of_leaves(s)
    t1 = time.time()
    ch = s.inject_refine_tags(leaves)
    torch.cuda.synchronize()
    t2 = time.time()
    line = "remesh %d: pick %.1f ms, remesh %.1f ms (%s) |" % (rep, 1e3 * (t1 - t), 1e3 * (t2 - t1), s.last_remesh()[0])
    for c in range(5):
        t3 = time.time()
        r0 = s.remeshes
        s.evolve(1)
        torch.cuda.synchronize()
        line += " cycle %.1f ms%s" % (1e3 * (time.time() - t3), "*" if s.remeshes != r0 else "")
    print(line, "bytes", s.device_bytes(), flush=True)
