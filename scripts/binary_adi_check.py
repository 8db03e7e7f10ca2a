"""tst/scripts/binary_adi/binary_adi.py on the GPU: inputs/disk/binary_cyl.in with gamma = 1.4, the three
Riemann solvers and de_switch = 0.2 / 1.0, one orbit; prints the wake-position errors (bound 0.03)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artemis_amd.driver import Simulation


def wake_errors(f):
    d = np.zeros((512, 256))
    for b in range(f.nblocks):
        x1a, x1b, x2a, x2b, _, _ = f.block_bounds(b)
        i0, j0 = int(round((x1a - 0.3) / 2.7 * 256)), int(round(x2a / (2 * np.pi) * 512))
        P = f.interior(f.field("gas.prim", b))
        d[j0:j0 + P.shape[2], i0:i0 + P.shape[3]] = P[0, 0]
    rc = 0.3 + (np.arange(256) + 0.5) * 2.7 / 256
    pc = (np.arange(512) + 0.5) * 2 * np.pi / 512
    sig = d - d.mean(axis=0)[None, :]

    def spiral_pos(r, h=0.05):
        v = (2.0 / (3 * h) * (r ** 1.5 - 1.5 * np.log(r) - 1.0)) % (2 * np.pi)
        return (np.pi - v) % (2 * np.pi) if r > 1.0 else (np.pi + v) % (2 * np.pi)
    ii, io = np.argwhere(rc >= 0.9)[0][0], np.argwhere(rc >= 1.1)[0][0]
    p_i, p_o = pc[np.argmax(sig[:, ii])], pc[np.argmax(sig[:, io])]
    return abs(p_i - spiral_pos(0.9)) / spiral_pos(0.9), abs(p_o - spiral_pos(1.1)) / spiral_pos(1.1)


if __name__ == "__main__":
    for fv in ("llf", "hlle", "hllc"):
        for dv in (0.2, 1.0):
            t = time.time()
            f = Simulation(os.path.join(ROOT, "inputs", "disk", "binary_cyl.in"),
                           ["parthenon/time/tlim={:.16f}".format(2.0 * np.pi), "gas/de_switch={:.1e}".format(dv),
                            "gas/gamma=1.4", "gas/riemann=" + fv])
            f.evolve()
            print(fv, dv, "cycles", f.ncycle, "errs %.4f %.4f" % wake_errors(f), "%.0fs" % (time.time() - t), flush=True)
