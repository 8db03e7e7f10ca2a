"""Adaptive mesh refinement (<parthenon/mesh> refinement = adaptive) through the HIP driver, on the reference's own
AMR decks (inputs/blast/blast_amr.in, inputs/linwave/linear_wave_amr.in; the reference has no regression test on
them).  The remeshing logic restates Parthenon's (absent submodule) from its published behaviour: tag with the gas
package's criterion after every cycle, split tagged leaves up to numlevel - 1, merge sibling groups that asked for it
derefine_count cycles in a row (and no finer neighbour), keep 2:1 balance, prolongate / restrict the conserved
variables with Artemis' own operators, then ConsToPrim -> boundary exchange -> PrimToCons.  Parity against Parthenon
itself is unpinned (absent submodule); the product is pinned against an INDEPENDENT second implementation of the same
algorithm (oracle/adaptive.py) -- tree shape, dt and every leaf bit for bit across the remeshes of the reference's
two AMR decks and a BASELINE configs[4] deck (test_hip_driver_equals_adaptive_oracle) -- and on properties:
conservation across remeshes, that the fine levels follow the feature, symmetry, derefinement, accuracy against
uniform meshes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECK = lambda *p: os.path.join(ROOT, "inputs", *p)


def levels(s):
    lv = [s.block_level(b) for b in range(s.nblocks)]
    return {l: lv.count(l) for l in sorted(set(lv))}


def cover(s):
    """total interior volume fraction check: sum over blocks of (zones x 4^-level) == root zones"""
    return sum(4.0 ** (-s.block_level(b)) for b in range(s.nblocks))


def test_blast_amr_deck_initial_mesh_and_tracking(hiplib):
    """blast_amr.in as shipped: Mesh::Initialize's refinement loop puts three levels around the blast (the problem
    generator fills every level: no prolongation of the initial condition); the level-2 blocks stay on the shock as it
    expands, mass is conserved to round-off across ~10 remeshes, the tree stays symmetric about theta = pi/4 and derefines behind the shock, and the
    leaves always tile the root mesh."""
    from artemis_amd.driver import Simulation
    s = Simulation(DECK("blast", "blast_amr.in"), [])
    assert s.uses_fused_path and s.remeshes >= 2
    lv0 = levels(s)
    assert set(lv0) == {0, 1, 2} and cover(s) == 256.0
    # the finest blocks sit at the blast centre (r = 2.5, theta = pi/4)
    fine = [s.block_bounds(b) for b in range(s.nblocks) if s.block_level(b) == 2]
    assert any(bb[0] <= 2.5 <= bb[1] and bb[2] <= np.pi / 4 <= bb[3] for bb in fine)
    h0 = s.history()
    n_fine0, r0 = lv0[2], s.remeshes
    s.evolve(300)
    h1 = s.history()
    lv1 = levels(s)
    assert s.remeshes > r0 + 3 and lv1[2] > n_fine0 and cover(s) == 256.0
    assert abs(h1[0] - h0[0]) < 1e-13 * h0[0]
    # the tree mirrors about theta = pi/4 (the blast centre): level map on the finest block grid (64 x 64)
    m = np.zeros((64, 64), int)
    for b in range(s.nblocks):
        x = s.block_bounds(b)
        i0, i1 = int(round((x[0] - 1.0) / 4.0 * 64)), int(round((x[1] - 1.0) / 4.0 * 64))
        j0, j1 = int(round(x[2] / (np.pi / 2) * 64)), int(round(x[3] / (np.pi / 2) * 64))
        m[j0:j1, i0:i1] = s.block_level(b)
    assert np.array_equal(m, m[::-1, :])
    # ... and has derefined behind the shock: the blast centre is no longer on the finest level
    jc, ic = 32, int((2.5 - 1.0) / 4.0 * 64)
    assert m[jc, ic] < 2 and m.max() == 2
    # the shock is inside level-2 blocks: the largest pressure jump of the mesh lies in one of them
    best = (-1.0, None)
    for b in range(s.nblocks):
        P = s.interior(s.field("gas.prim", b))[4, 0]
        g = max(np.abs(np.diff(P, axis=0)).max(), np.abs(np.diff(P, axis=1)).max())
        if g > best[0]:
            best = (g, s.block_level(b))
    assert best[1] == 2
    for b in range(s.nblocks):
        assert np.isfinite(s.field("gas.prim", b)[[0, 1, 2, 3, 5]]).all()
    s.close()


def test_linear_wave_amr_deck_conserves_and_converges(hiplib):
    """linear_wave_amr.in as shipped for one period: the refined band rides on the crests (blocks are created ahead of
    it and merged behind it after derefine_count cycles), mass / momentum / energy are conserved to round-off across
    every remesh (smooth flow: the prolongation is conservative and no floor acts), and the error after one period
    lies between the uniform fine and uniform coarse meshes'."""
    from artemis_amd.driver import Simulation
    ov = ["problem/nperiod=1"]
    s = Simulation(DECK("linwave", "linear_wave_amr.in"), ov)
    lv0 = levels(s)
    assert set(lv0) == {0, 1} and cover(s) == 32.0
    h0 = s.history()
    seen_merge, prev = False, lv0[1]
    while s.time < s.tlim:
        if s.evolve(20) == 0:
            break
        cur = levels(s).get(1, 0)
        seen_merge = seen_merge or cur < prev
        prev = cur
        assert cover(s) == 32.0
    h1 = s.history()
    assert s.remeshes > 10 and seen_merge
    scale = np.abs(h0).max()  # mass, three momenta: round-off
    assert np.allclose(h1[:4], h0[:4], rtol=0, atol=2e-13 * scale), (h1[:4] - h0[:4])
    # total energy: the reference's remesh does not run SetAuxillaryFields (fill_derived.cpp:28): E is rebuilt from the
    # prolongated internal energy, i.e. conserved to O(dx^2) of the wave's kinetic energy per remesh, not to round-off
    assert abs(h1[4] - h0[4]) < 2e-8 * scale, h1[4] - h0[4]
    e_amr = s.errors()[0]
    s.close()
    base = ["problem/nperiod=1", "parthenon/mesh/refinement=none", "gas/refine_field=none"]
    c = Simulation(DECK("linwave", "linear_wave_amr.in"), base)
    c.evolve()
    f = Simulation(DECK("linwave", "linear_wave_amr.in"), base + ["parthenon/mesh/nx1=256", "parthenon/mesh/nx2=128"])
    f.evolve()
    e_c, e_f = c.errors()[0], f.errors()[0]
    c.close(), f.close()
    assert e_f < e_amr < 1.05 * e_c, (e_f, e_amr, e_c)


def test_adaptive_mesh_with_viscosity_distance_table_follows_the_blocks(hiplib, monkeypatch, option):
    """The static Coords::Distance table of the viscous fluxes (artemis_hip_viscous_distance_fill) is rebuilt with
    every remesh: an adaptive viscous run with the table equals the same run evaluating the distances on the fly,
    bit for bit, blocks and levels included."""
    from artemis_amd.driver import Simulation
    ov = ["problem/nperiod=1", "physics/viscosity=true", "gas/viscosity/type=constant", "gas/viscosity/nu=2.0e-3",
          "gas/viscosity/averaging=arithmetic"]

    def run():
        s = Simulation(DECK("linwave", "linear_wave_amr.in"), ov)
        s.evolve(60)
        out = (s.remeshes, [s.block_level(b) for b in range(s.nblocks)],
               [s.interior(s.field("gas.prim", b)).copy() for b in range(s.nblocks)], s.dt)
        s.close()
        return out

    a = run()
    option("no_distance_table", 1)
    b = run()
    assert a[0] == b[0] and a[0] >= 2 and a[1] == b[1] and a[3] == b[3]
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)


def test_adaptive_mesh_without_a_trigger_equals_the_static_root_mesh(hiplib):
    """refinement = adaptive with a criterion that never fires: no remesh, and the run equals refinement = static with
    no region (the same multilevel code path on a one-level tree), bit for bit."""
    from artemis_amd.driver import Simulation
    ov = ["problem/nperiod=1", "parthenon/time/nlim=30", "gas/refine_thr=2.0", "gas/deref_thr=0.0"]
    a = Simulation(DECK("linwave", "linear_wave_amr.in"), ov)
    b = Simulation(DECK("linwave", "linear_wave_amr.in"), ov + ["parthenon/mesh/refinement=none", "gas/refine_field=none"])
    b.set_path("unfused")
    a.evolve(), b.evolve()
    assert a.remeshes == 0 and a.nblocks == b.nblocks == 32 and a.dt == b.dt
    for q in range(32):
        assert np.array_equal(a.interior(a.field("gas.prim", q)), b.interior(b.field("gas.prim", q))), q
    a.close(), b.close()


def test_disk_with_planet_dust_and_adaptive_mesh(hiplib):
    """BASELINE configs[4]'s ingredients in one run (minus REBOUND: the planet sits still in the frame that rotates
    with it): inputs/disk/disk_nbody_cyl.in + a 1e-3 planet at r = 1 with Plummer softening, the rotating frame, one
    dust species with simple_dust drag, `ic` conditions, and a three-level adaptive mesh refining on the gas density
    (the midplane of the inner disk).  The run must build its levels from the initial condition, stay finite and
    positive, keep dt in the disk decks' window, tile the root mesh, and feel the planet (a net force on the star)."""
    from artemis_amd.driver import Simulation
    ov = ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "parthenon/mesh/nx3=32", "parthenon/meshblock/nx1=8",
          "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8", "parthenon/mesh/refinement=adaptive",
          "parthenon/mesh/numlevel=3", "parthenon/mesh/derefine_count=5", "gas/refine_field=density",
          "gas/refine_type=magnitude", "gas/refine_thr=0.5", "gas/deref_thr=0.2",
          "physics/rotating_frame=true", "rotating_frame/omega=1.0",
          "physics/dust=true", "dust/nspecies=1", "dust/cfl=0.3", "dust/reconstruct=plm", "dust/riemann=hlle",
          "dust/dfloor=1e-10", "physics/drag=true", "drag/type=simple_dust", "dust/stopping_time/type=constant",
          "dust/stopping_time/tau=0.1", "dust/sizes=1.0",
          "nbody/particle2/mass=1.0e-3", "nbody/particle2/couple=1", "nbody/particle2/soft/type=plummer",
          "nbody/particle2/soft/radius=0.03", "nbody/particle2/initialize/x=1.0", "nbody/particle2/initialize/vy=1.0"]
    s = Simulation(DECK("disk", "disk_nbody_cyl.in"), ov)
    lv = levels(s)
    assert set(lv) == {0, 1, 2} and s.remeshes >= 2 and s.uses_fused_path  # (round 4: drag + n-body inside the one-kernel stages)
    assert sum(8.0 ** (-s.block_level(b)) for b in range(s.nblocks)) == 64.0
    s.evolve(20)
    assert s.ncycle == 20 and 1e-4 < s.dt < 3e-2
    for b in range(s.nblocks):
        g, d = s.field("gas.prim", b)[[0, 1, 2, 3, 5]], s.field("dust.prim", b)
        assert np.isfinite(g).all() and np.isfinite(d).all() and g[0].min() > 0 and g[4].min() > 0 and d[0].min() > 0
    f = s.nbody_force()
    assert f.shape == (2, 7) and np.isfinite(f).all() and np.abs(f[0, :3]).max() > 0.0
    s.close()


def amr_cases_thick():
    import amr_cases
    return amr_cases.THICK_DISK


# ---- HIP driver == the independent adaptive oracle -----------------------------------------------------------------
@pytest.mark.parametrize("name,kw,cycles,batch,min_remeshes,levels", [
    ("blast_amr", dict(n=128, derefine_count=5), 120, 20, 6, {0, 1, 2}),        # the deck's own 128^2 root mesh
    ("linear_wave_amr", dict(derefine_count=3), 100, 20, 5, {0, 1}),           # as shipped
    ("disk_planet_dust_amr", dict(), 40, 10, 8, {1, 2, 3}),                    # BASELINE configs[4], numlevel = 4
    ("disk_planet_dust_amr", dict(n=64, thr=0.5), 20, 10, 2, {1, 2, 3}),       # the same on a 64^2 root (300+ blocks)
    # ... and in THREE dimensions: 16 x 16 x 8 root in 8^3 blocks, four levels, 312 -> 560 blocks of both fluids
    ("disk_planet_dust_amr", dict(n=16, planet=3e-2, thr=2.5, nz=8, zlim=0.01), 18, 6, 4, {1, 2, 3}),
    # ... and with REAL vertical extent (round-4 verdict 5 i): |z| <= 0.2 = one scale height at the planet (h0 = 0.2),
    # 16 x 32 x 8 root over 0.5 < r < 2.5.  The stratification holds the layers above and below the midplane of the inner
    # disk at level 3 from the start (256 of the finest blocks lie at |z| > 0.1), the planet sits at z = 0.08 -- in the
    # second finest-level block layer above the midplane -- and its envelope refines there: 400 -> 484 blocks, 3 remeshes
    ("disk_planet_dust_amr", dict(**amr_cases_thick()), 15, 5, 3, {1, 2, 3}),
])
def test_hip_driver_equals_adaptive_oracle(hiplib, name, kw, cycles, batch, min_remeshes, levels):
    """The HIP driver against oracle/adaptive.py (an independent restatement of the remeshing: tests/amr_cases.py,
    tests/test_adaptive_oracle.py) after every batch of cycles: same leaves in the same Z-order, same levels and bounds,
    same dt and time, and every leaf equal bit for bit, ghost zones included -- across >= min_remeshes remeshes with
    refinement and derefinement.  The configs[4] rows run inputs/disk/disk_nbody_cyl.in with a planet, a dust species with
    drag, alpha viscosity, the rotating frame, `ic` conditions and FOUR levels; the last one in 3-D (8^3 blocks: octant
    hand-over, x3 restriction / prolongation / flux correction of both fluids, the N-body task and the viscous cross
    terms in all three planes; a 16 x 16 x 8 root rather than 32 x 32 x 8 because the oracle is a Python loop over
    per-block C oracles: 560 blocks at the end)."""
    import amr_cases
    from artemis_amd.driver import Simulation
    case = getattr(amr_cases, name)(**kw)
    s = Simulation(amr_cases.DECK(*case["deck"]), case["overrides"])
    m = case["oracle"]()
    assert s.remeshes == m.remeshes and s.uses_fused_path  # (configs[4] too since round 4: drag + n-body inside the stages)
    r0, done, seen = s.remeshes, 0, set()
    while done < cycles:
        done += s.evolve(min(batch, cycles - done))
        m.evolve(case["tlim"], done)
        amr_cases.compare(s, m, case["dust"])
        assert s.remeshes == m.remeshes
        seen |= set(m.level_counts())
    assert s.remeshes - r0 >= min_remeshes and seen >= levels, (s.remeshes, r0, seen)
    if name == "disk_planet_dust_amr":
        f = s.nbody_force()
        want = m.nbody_force()  # per-block partial sums of the oracle, blocks that left the mesh included
        assert f.shape == want.shape == (2, 7) and np.abs(f - want).max() <= 1e-11 * max(1.0, np.abs(want).max()), (f, want)
    s.close()


@pytest.mark.gpu
def test_config4_at_bench_size_one_kernel_stages_equal_the_task_chain(hiplib):
    """BASELINE configs[4]'s combination at the size bench.py --workload disk_amr runs (128 x 128 x 16 root over |z| < 0.2 in
    16^3 blocks, four adaptive levels: ~7000 blocks, 29 M zones of gas and dust): the one-kernel stages (curvilinear tile
    marches, viscous source, N-body and drag inside, flux correction as a fix-up) against the per-task chain -- same mesh,
    same dt, every leaf of both fluids equal bit for bit after three cycles; the particle forces to round-off (their sums
    are formed in a different order).  (The small forms of the deck run against the adaptive oracle above.)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "amr_paths_check.py"), "3"], capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert " 0 of " in r.stdout and "leaf arrays differ" in r.stdout, r.stdout[-500:]
