"""The three adaptive-mesh cases the HIP driver (GPU) and the host logic on the CPU double are checked on against the
independent adaptive oracle (oracle/adaptive.py): the reference's two AMR decks and a BASELINE configs[4] deck
(inputs/disk + N-body planet + four-level AMR + gas and dust).  Each case = the deck with overrides for the product
and the same problem stated for the oracle."""
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECK = lambda *p: os.path.join(ROOT, "inputs", *p)
PI = 3.141592653589793


def blast_amr(n=128, derefine_count=10):
    """inputs/blast/blast_amr.in (cylindrical 2-D blast, pressure-gradient criterion, numlevel 3); n = root zones per
    dimension (the deck ships 128), 8 x 8 blocks."""
    ov = ["parthenon/mesh/nx1=%d" % n, "parthenon/mesh/nx2=%d" % n, "parthenon/mesh/derefine_count=%d" % derefine_count,
          "parthenon/time/nlim=-1"]
    pg = dict(radius=0.1, internal_energy=1.0, p0=1e-5, d0=1.0, samples=0, symmetry="cylindrical",
              x0=(2.5, 0.7853981633974485, 0.0))

    def oracle():
        from oracle.adaptive import AdaptiveOracle
        return AdaptiveOracle((n, n, 1), (8, 8, 1), (1.0, 0.0, -0.5), (5.0, 1.570796326794897, 0.5), ("outflow",) * 6,
                              numlevel=3, refine_field="pressure", refine_type="gradient", refine_thr=10.0,
                              derefine_count=derefine_count, pgen=lambda o: o.pgen_blast(post_init=False, **pg),
                              ng=2, integrator="rk2", reconstruct="plm", riemann="hlle", gamma=1.4, dfloor=1e-10,
                              siefloor=1e-10, cfl=0.3, coordinates="cylindrical").initialize()
    return dict(deck=("blast", "blast_amr.in"), overrides=ov, oracle=oracle, tlim=0.5, dust=False)


def linear_wave_amr(derefine_count=10):
    """inputs/linwave/linear_wave_amr.in as shipped (Cartesian 128 x 64 in 16 x 16 blocks, density-magnitude criterion,
    numlevel 2, periodic): the refined band follows the crests, so blocks are created AND merged."""
    ov = ["parthenon/mesh/derefine_count=%d" % derefine_count, "problem/nperiod=1", "parthenon/time/nlim=-1"]

    def oracle():
        from oracle.adaptive import AdaptiveOracle
        m = AdaptiveOracle((128, 64, 1), (16, 16, 1), (0.0, 0.0, -0.5), (2.236068, 1.118034, 0.5), ("periodic",) * 6,
                           numlevel=2, refine_field="density", refine_type="magnitude", refine_thr=1.0008, deref_thr=1.0008,
                           derefine_count=derefine_count,
                           pgen=lambda o: o.pgen_linear_wave(0, 1.0e-3, 0.0, along=(False, False, False), nperiod=1.0,
                                                             post_init=False),
                           ng=2, integrator="rk2", reconstruct="plm", riemann="hllc", gamma=1.66666666667, cfl=0.9)
        return m.initialize()
    return dict(deck=("linwave", "linear_wave_amr.in"), overrides=ov, oracle=oracle, tlim=-1.0, dust=False)


def disk_planet_dust_amr(n=32, planet=1.0e-2, thr=0.8, derefine_count=3, nz=1, zlim=1.0, h0=0.05, rlim=(0.3, 4.3), nphi=None, zp=0.0):
    """BASELINE configs[4]: inputs/disk/disk_nbody_cyl.in (cylindrical disk, `ic` conditions, alpha viscosity, N-body
    gravity) in 2-D with a planet on a circular orbit at r = 1 (static in the frame rotating with it: <nbody>
    integrator = none, the REBOUND integration is outside this build), one dust species with simple_dust drag, and
    FOUR refinement levels (numlevel = 4) on the pressure-gradient criterion -- the criterion and numlevel
    inputs/disk/binary_nbody_cyl.in:40-41,77-79 carries.  The planet's growing wake drives refinement to level 3.
    nz > 1: the same in THREE dimensions (8^3 blocks, nz root zones over |z| < zlim -- a slab thin against the scale
    height, so that the criterion follows the planet's wake as in 2-D rather than the vertical stratification, which
    would refine the whole midplane to the finest level: thousands of 8^3 blocks for a Python-driven oracle)."""
    nphi = nphi or n
    ov = ["parthenon/mesh/nx1=%d" % n, "parthenon/mesh/nx2=%d" % nphi, "parthenon/mesh/nx3=%d" % nz, "parthenon/meshblock/nx1=8",
          "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=%d" % (8 if nz > 1 else 1),
          "parthenon/mesh/x3min=%r" % -zlim, "parthenon/mesh/x3max=%r" % zlim, "parthenon/mesh/x1min=%r" % rlim[0],
          "parthenon/mesh/x1max=%r" % rlim[1], "problem/h0=%r" % h0, "parthenon/mesh/refinement=adaptive",
          "parthenon/mesh/numlevel=4", "parthenon/mesh/derefine_count=%d" % derefine_count, "gas/refine_field=pressure",
          "gas/refine_type=gradient", "gas/refine_thr=%r" % thr,
          "physics/rotating_frame=true", "rotating_frame/omega=1.0",
          "physics/dust=true", "dust/nspecies=1", "dust/cfl=0.3", "dust/reconstruct=plm", "dust/riemann=hlle",
          "dust/dfloor=1e-10", "physics/drag=true", "drag/type=simple_dust", "dust/stopping_time/type=constant",
          "dust/stopping_time/tau=0.1", "dust/sizes=1.0",
          "nbody/particle2/mass=%r" % planet, "nbody/particle2/couple=1", "nbody/particle2/soft/type=plummer",
          "nbody/particle2/soft/radius=0.03", "nbody/particle2/initialize/x=1.0", "nbody/particle2/initialize/vy=1.0",
          "nbody/particle2/initialize/z=%r" % zp,
          "parthenon/time/nlim=-1"]
    # nbody/nbody_setup.cpp:690-714 restated: total mass rescaled to <nbody> mtot (absent: the sum), positions and
    # velocities shifted by the mass-weighted sums as written there (not divided by the total mass)
    raw = [dict(m=1.0, x=0.0, y=0.0, z=0.0, vx=0.0, vy=0.0, vz=0.0, rs=0.0),
           dict(m=planet, x=1.0, y=0.0, z=zp, vx=0.0, vy=1.0, vz=0.0, rs=0.03)]
    mtot, R, V = 0.0, [0.0] * 3, [0.0] * 3
    for p in raw:
        mtot += p["m"]
        for d, (q, v) in enumerate((("x", "vx"), ("y", "vy"), ("z", "vz"))):
            R[d] += p["m"] * p[q]
            V[d] += p["m"] * p[v]
    mresc = mtot
    parts = [dict(GM=1.0 * (p["m"] * mresc / mtot), pos=(p["x"] - R[0], p["y"] - R[1], p["z"] - R[2]),
                  vel=(p["vx"] - V[0], p["vy"] - V[1], p["vz"] - V[2]), rs=p["rs"], spline=0, couple=1) for p in raw]
    gm = 1.0 * mresc  # nbody.cpp:109

    def setup(o):
        o.set_gravity_nbody(parts, frame_correction=True, gm=gm)
        o.set_rotating_frame(1.0, 0.0)
        o.set_viscosity("alpha", alpha=1e-3, r0=1.0, Omega0=math.sqrt(gm / (1.0 * 1.0 * 1.0)))
        o.set_drag("simple_dust", "constant", tau=[0.1])

    def oracle():
        from oracle.adaptive import AdaptiveOracle
        m = AdaptiveOracle((n, nphi, nz), (8, 8, 8 if nz > 1 else 1), (rlim[0], -PI, -zlim), (rlim[1], PI, zlim),
                           ("ic", "ic", "periodic", "periodic", "ic", "ic"), numlevel=4, refine_field="pressure",
                           refine_type="gradient", refine_thr=thr, derefine_count=derefine_count, setup=setup,
                           pgen=lambda o: o.pgen_disk(r0=1.0, rho0=1.0, dslope=-2.25, flare=0.25, h0=h0, dens_min=1e-10,
                                                      pres_min=1e-15, polytropic_index=1.0, post_init=False),
                           ng=2, integrator="rk2", reconstruct="plm", riemann="hllc", gamma=1.4, dfloor=1e-10,
                           siefloor=1e-10, cfl=0.3, ns_dust=1, dust_reconstruct="plm", dust_riemann="hlle",
                           dust_dfloor=1e-10, dust_cfl=0.3, coordinates="cylindrical")
        m.diffusion = m.gravity = m.rframe = m.drag = True
        return m.initialize()
    return dict(deck=("disk", "disk_nbody_cyl.in"), overrides=ov, oracle=oracle, tlim=62.8, dust=True)


# configs[4] with real vertical extent: |z| <= 0.2 over 0.5 < r < 2.5 with h0 = 0.2 (one scale height at the planet, 2.4 at
# the inner edge), 16 x 32 x 8 root zones in 8^3 blocks, the planet 0.08 above the midplane.  thr = 1.5 leaves the
# midplane of the inner disk at level 2 and takes the layers at |z| > 0.1 there to level 3 (the vertical pressure gradient
# is what fires), and the planet's envelope refines around z = 0.08 over the first dozen cycles.
THICK_DISK = dict(n=16, nphi=32, planet=6e-2, thr=1.5, derefine_count=2, nz=8, zlim=0.2, h0=0.2, rlim=(0.5, 2.5), zp=0.08)


def compare(sim, m, dust, ghosts=True):
    """The product's mesh and state against the adaptive oracle's: tree shape (levels and bounds of the Z-ordered
    leaves), dt, time, and every leaf bit for bit (FillGhost variables, ghost zones included; the stored pressure is
    compared on interior zones)."""
    assert sim.nblocks == len(m.blocks), (sim.nblocks, len(m.blocks))
    assert [sim.block_level(b) for b in range(sim.nblocks)] == [l for l, _ in m.leaves]
    assert sim.time == m.time and sim.dt == m.dt, (sim.time, m.time, sim.dt, m.dt)
    keep = [0, 1, 2, 3, 5]
    for b, blk in enumerate(m.blocks):
        assert list(sim.block_bounds(b)) == m.block_bounds(b), b
        got = sim.field("gas.prim", b)
        if ghosts:
            assert np.array_equal(got[keep], blk.gprim[keep]), (b, m.leaves[b])
        assert np.array_equal(sim.interior(got), blk.interior(blk.gprim)), (b, m.leaves[b])
        if dust:
            gd = sim.field("dust.prim", b)
            assert np.array_equal(gd if ghosts else sim.interior(gd), blk.dprim if ghosts else blk.interior(blk.dprim)), b
