"""Worker of tests/test_multirank_cpu.py: one rank of a gloo world running the product's host
driver against the CPU test double.  Writes its blocks' interiors to an .npz for the parent."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    spec = json.loads(sys.argv[1])
    from artemis_amd.driver import Simulation, TorchComm
    lib = C.CDLL(os.path.join(ROOT, "tests", "_build", "libartemis_cpudouble.so"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    comm = None
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"],
                                rank=rank, world_size=world)
        comm = TorchComm(torch.device("cpu"))
    sim = Simulation(os.path.join(ROOT, "inputs", *spec["deck"]), spec["overrides"], comm=comm, lib=lib)
    if spec.get("path"):
        sim.set_path(spec["path"])
    if spec.get("overlap"):
        sim.set_overlap(True)
    n = sim.evolve(spec.get("cycles", -1))
    out = {"ncycle": sim.ncycle, "time": sim.time, "dt": sim.dt, "n": n, "nblocks": sim.nblocks,
           "fused": sim.uses_fused_path, "tuned": sim.uses_tuned_kernel, "remeshes": sim.remeshes,
           "levels": [sim.block_level(b) for b in range(sim.nblocks)], "load_balance": sim.load_balance}
    out["nblocks"] = sim.nblocks  # (an adaptive mesh: the count after the run)
    hist = sim.history()
    errs = sim.errors()
    arrays = {}
    for b in range(sim.nblocks):
        arrays["prim%d" % b] = sim.interior(sim.field("gas.prim", b))
        arrays["bounds%d" % b] = np.array(sim.block_bounds(b))
        if spec.get("dust"):
            arrays["dust%d" % b] = sim.interior(sim.field("dust.prim", b))
    np.savez(spec["out"] + ".rank%d.npz" % rank, meta=json.dumps(out), hist=hist, errs=errs, **arrays)
    sim.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
