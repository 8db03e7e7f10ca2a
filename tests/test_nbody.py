"""N-body gravity coupling: Gravity::NBodyGravity<GEOM> (gravity/nbody_gravity.hpp:28-221) with the
particle functions of nbody/particle_base.hpp, through the reference's own deck inputs/disk/disk_nbody_cyl.in
(<nbody> integrator = none: the particle never moves, so no REBOUND is involved) and the bounds its regression test
holds (tst/scripts/disk_nbody/disk_nbody.py; numbers in tests/golden/reference_pins.json).  The per-task kernel parity
in three geometries with sinks and softening is tests/test_parity_sources.py::test_nbody_gravity."""
import os

import numpy as np
import pytest

from oracle.oracle import Oracle
from pins import PINS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECK = os.path.join(ROOT, "inputs", "disk", "disk_nbody_cyl.in")
NB = PINS["disk_nbody"]
_PI = 3.141592653589793


def nbody_disk_oracle(nx, gam, b="ic"):
    bcn = "ic" if b == "ic" else "disk_extrap"
    o = Oracle(nx, (0.3, -_PI, -1.0), (4.3, _PI, 1.0), ng=2, reconstruct="plm", riemann="hllc", gamma=1.4, dfloor=1e-10,
               siefloor=1e-10, cfl=0.3, integrator="rk2", coordinates="cylindrical",
               bc=(bcn, bcn, "periodic", "periodic", bcn, bcn))
    o.set_gravity_nbody([dict(GM=1.0)])  # <nbody/particle1> mass = 1, soft type none, at the origin
    o.set_viscosity("alpha", alpha=1e-3, r0=1.0, Omega0=1.0)
    o.pgen_disk(r0=1.0, rho0=1.0, dslope=-2.25, flare=0.25, h0=0.05, dens_min=1e-10, pres_min=1e-15, polytropic_index=gam)
    return o


def disk_checks(d0, P, dt):
    d, T = P[0], P[4] / P[0]
    assert not np.isnan(P).any() and d.min() > 0.0 and T.min() > 0.0
    assert NB["dt_low"] < dt < NB["dt_high"], dt
    err = np.sqrt((d0 * (d - d0) ** 2).sum()) / d0.sum()
    assert err <= NB["density_err_max"], err
    return err


@pytest.mark.parametrize("gam,b", [(1.0, "ic"), (1.4, "extrap")])
def test_disk_nbody_reference_test_pin(gam, b):
    """disk_nbody.py:36-140 on the shipped deck (128 x 64 x 32, 10 cycles): no NaN, positive density and temperature,
    1e-4 < dt < 3e-2, density error <= 5e-3.  A single unsoftened particle at the origin is a point mass: the run must
    also track the <gravity/point> oracle run to round-off (different expression trees, same physics)."""
    o = nbody_disk_oracle((128, 64, 32), gam, b)
    d0 = o.interior(o.gprim)[0].copy()
    o.evolve(62.8, NB["cycles"])
    err = disk_checks(d0, o.interior(o.gprim), o.dt)
    assert o.ncycle == NB["cycles"] and err > 1e-5  # (not vacuous: the disk does relax a little)
    f = o.nbody_force()
    assert f.shape == (1, 7) and np.all(f[0, [0, 4, 5, 6]] == 0.0)  # no sink: nothing accreted
    assert np.abs(f[0, 1:4]).max() < 1e-10 * 10.0  # axisymmetric disk: no net pull on the central body


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["fused", "unfused"])
@pytest.mark.parametrize("b", NB["bc"])
def test_disk_nbody_deck_hip_equals_oracle(hiplib, b, path):
    """path = fused (the default since round 4): NBodyGravity inside the curvilinear tile march, its seven sums per
    particle from artemis_hip_nbody_force_sums in device accumulators; unfused: the task with its host-side reduction."""
    from artemis_amd.driver import Simulation
    nx = (64, 32, 16)
    ov = ["parthenon/time/nlim=6"] + [f"parthenon/mesh/nx{d + 1}={n}" for d, n in enumerate(nx)] + \
         [f"parthenon/meshblock/nx{d + 1}={n}" for d, n in enumerate(nx)]
    for d in ("x1", "x3"):
        ov += [f"parthenon/mesh/i{d}_bc={b}", f"parthenon/mesh/o{d}_bc={b}"]
    s = Simulation(DECK, ov)
    assert s.uses_fused_path
    s.set_path(path)
    assert s.uses_fused_path == (path == "fused")
    s.evolve()
    o = nbody_disk_oracle(nx, 1.0, b)
    o.evolve(62.8, 6)
    assert s.ncycle == o.ncycle == 6
    got, want = s.field("gas.prim"), o.gprim
    if b == "ic":
        assert s.dt == o.dt and np.array_equal(got, want)
    else:  # the extrap condition takes log / exp of the state on the device: to rounding
        assert abs(s.dt - o.dt) <= 1e-12 * o.dt
        assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))
    fs, fo = s.nbody_force(), o.nbody_force()
    assert fs.shape == fo.shape == (1, 7)
    # sums of ~3e4 signed O(1e-3) terms that cancel to ~1e-15 on an axisymmetric disk: absolute round-off tolerance
    assert np.max(np.abs(fs - fo)) <= 1e-12
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("gam", NB["gamma"])
def test_disk_nbody_deck_reference_checks(hiplib, gam):
    """The deck as shipped (128 x 64 x 32 in 32^3 blocks), disk_nbody.py's overrides, through the HIP driver."""
    from artemis_amd.driver import Simulation
    s = Simulation(DECK, ["parthenon/time/nlim=%d" % NB["cycles"], "problem/polytropic_index=%.2f" % gam])
    assert s.nblocks == 8
    d0 = [s.interior(s.field("gas.prim", blk))[0].copy() for blk in range(s.nblocks)]
    s.evolve()
    assert s.ncycle == NB["cycles"]
    P = np.concatenate([s.interior(s.field("gas.prim", blk)).reshape(6, -1) for blk in range(s.nblocks)], axis=1)
    disk_checks(np.concatenate([d.ravel() for d in d0]), P, s.dt)
    s.close()
