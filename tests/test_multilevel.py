"""Static mesh refinement (BASELINE configs[3]/[4] data path): block tree, ghost exchange
across levels (RestrictAverage / ProlongateSharedMinMod on coarse buffers), flux correction.

The checker is oracle/multilevel.py -- a numpy + per-block-oracle restatement written independently of the
product's C++ driver (different language, different decomposition of the exchange).  Parthenon is absent
from the reference checkout, so parity against a Parthenon build is UNPINNED; what is pinned:
  * the HIP path == the multilevel oracle, bit for bit, ghosts included (GPU tests),
  * the host logic == the oracle on the CPU double, 1 rank and 2 ranks over gloo (CPU tests),
  * conservation across level boundaries to round-off WITH flux correction (and its violation without),
  * exactness on a uniform state, reference-held disk.py bounds on inputs/disk/disk_cart.in (tests/golden).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle.multilevel import BlockTree, MultiLevelOracle
from pins import DISK

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECK = lambda *p: os.path.join(ROOT, "inputs", *p)

KW = dict(reconstruct="plm", gamma=1.4, dfloor=1e-10, siefloor=1e-10, cfl=0.3)


def region_overrides(n, lo, hi, level=1):
    blk = "parthenon/static_refinement%d" % n
    out = ["parthenon/mesh/refinement=static", f"{blk}/level={level}"]
    for d in range(3):
        out += [f"{blk}/x{d + 1}min={lo[d]}", f"{blk}/x{d + 1}max={hi[d]}"]
    return out


CASES = {
    # 2-D cylindrical blast, outflow, refined centre: 12 coarse + 16 fine blocks of 8x8
    "blast2d": dict(
        deck=("blast", "blast.in"),
        ov=["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8",
            "gas/riemann=hllc", "problem/radius=0.45", "problem/p0=0.1", "problem/samples=0",
            "problem/symmetry=cylindrical", "parthenon/time/nlim=20"] + region_overrides(1, (-0.3, -0.3, -0.5), (0.3, 0.3, 0.5)),
        oracle=dict(mesh=(32, 32, 1), block=(8, 8, 1), lo=(-1, -1, -0.5), hi=(1, 1, 0.5), bc=("outflow",) * 6,
                    regions=[(1, (-0.3, 0.3), (-0.3, 0.3), (-0.5, 0.5))], riemann="hllc"),
        pgen=dict(radius=0.45, internal_energy=1.0, p0=0.1, d0=1.0, samples=0, symmetry="cylindrical"), nlim=20,
        blocks=(12, 16)),
    # 3-D, periodic x1 / x3 with the refined region AT the periodic boundary, reflecting x2, viscosity (edge and
    # corner ghost zones and diffusion-flux correction matter), rk3
    "visc3d": dict(
        deck=("blast", "blast.in"),
        ov=["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=16", "parthenon/mesh/nx3=16", "parthenon/mesh/x3min=-1.0",
            "parthenon/mesh/x3max=1.0", "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8",
            "parthenon/mesh/ix1_bc=periodic", "parthenon/mesh/ox1_bc=periodic", "parthenon/mesh/ix2_bc=reflecting",
            "parthenon/mesh/ox2_bc=reflecting", "parthenon/mesh/ix3_bc=periodic", "parthenon/mesh/ox3_bc=periodic",
            "physics/viscosity=true", "gas/viscosity/type=constant", "gas/viscosity/nu=0.02", "gas/riemann=hlle",
            "problem/radius=0.45", "problem/samples=0", "problem/symmetry=spherical", "problem/p0=0.1", "problem/x1=-0.7",
            "parthenon/time/integrator=rk3", "parthenon/time/nlim=6"]
        + region_overrides(1, (-1.0, -0.2, -0.2), (-0.6, 0.2, 0.2)),
        oracle=dict(mesh=(32, 16, 16), block=(8, 8, 8), lo=(-1, -1, -1), hi=(1, 1, 1),
                    bc=("periodic", "periodic", "reflecting", "reflecting", "periodic", "periodic"),
                    regions=[(1, (-1.0, -0.6), (-0.2, 0.2), (-0.2, 0.2))], riemann="hlle", integrator="rk3"),
        visc=0.02, pgen=dict(radius=0.45, internal_energy=1.0, p0=0.1, d0=1.0, samples=0, x0=(-0.7, 0, 0)), nlim=6,
        blocks=(12, 32)),
    # spherical-polar wedge (BASELINE configs[3]'s geometry): curvilinear RestrictAverage / ProlongateSharedMinMod on
    # the coarse buffers with their own metric tables, area-weighted flux correction with spherical face areas,
    # reflecting r / theta, periodic phi with the refined region next to the periodic seam
    "sph3d": dict(
        deck=("blast", "blast.in"),
        ov=["artemis/coordinates=spherical", "parthenon/mesh/nx1=32", "parthenon/mesh/nx2=16", "parthenon/mesh/nx3=16",
            "parthenon/mesh/x1min=0.4", "parthenon/mesh/x1max=2.0", "parthenon/mesh/x2min=0.6", "parthenon/mesh/x2max=2.5",
            "parthenon/mesh/x3min=0.0", "parthenon/mesh/x3max=6.283185307179586",
            "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8",
            "parthenon/mesh/ix1_bc=reflecting", "parthenon/mesh/ox1_bc=reflecting", "parthenon/mesh/ix2_bc=reflecting",
            "parthenon/mesh/ox2_bc=reflecting", "parthenon/mesh/ix3_bc=periodic", "parthenon/mesh/ox3_bc=periodic",
            "gas/riemann=hlle", "problem/radius=1.0", "problem/samples=0", "problem/symmetry=spherical", "problem/p0=0.1",
            "parthenon/time/nlim=6"] + region_overrides(1, (0.9, 1.2, 0.0), (1.4, 1.9, 1.5)),
        oracle=dict(mesh=(32, 16, 16), block=(8, 8, 8), lo=(0.4, 0.6, 0.0), hi=(2.0, 2.5, 6.283185307179586),
                    bc=("reflecting", "reflecting", "reflecting", "reflecting", "periodic", "periodic"),
                    regions=[(1, (0.9, 1.4), (1.2, 1.9), (0.0, 1.5))], riemann="hlle", coordinates="spherical"),
        pgen=dict(radius=1.0, internal_energy=1.0, p0=0.1, d0=1.0, samples=0), nlim=6, blocks=None),
    # two refinement levels (a level-2 region forces level 1 around it through 2:1 balance), 2-D, vl2
    "twolevel2d": dict(
        deck=("blast", "blast.in"),
        ov=["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=64", "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8",
            "gas/riemann=hllc", "problem/radius=0.2", "problem/samples=0", "problem/symmetry=cylindrical", "problem/p0=0.01",
            "parthenon/time/integrator=vl2", "parthenon/time/nlim=10"] + region_overrides(1, (-0.05, -0.05, -0.5), (0.05, 0.05, 0.5), 2),
        oracle=dict(mesh=(64, 64, 1), block=(8, 8, 1), lo=(-1, -1, -0.5), hi=(1, 1, 0.5), bc=("outflow",) * 6,
                    regions=[(2, (-0.05, 0.05), (-0.05, 0.05), (-0.5, 0.5))], riemann="hllc", integrator="vl2"),
        pgen=dict(radius=0.2, internal_energy=1.0, p0=0.01, d0=1.0, samples=0, symmetry="cylindrical"), nlim=10,
        blocks=None),
}


def run_oracle(case):
    c = CASES[case]
    o = c["oracle"]
    kw = dict(KW, riemann=o["riemann"])
    if o.get("coordinates"):
        kw["coordinates"] = o["coordinates"]
    m = MultiLevelOracle(o["mesh"], o["block"], o["lo"], o["hi"], o["bc"], regions=o["regions"], ng=2,
                         integrator=o.get("integrator", "rk2"), **kw)
    if c.get("visc"):
        m.diffusion = True
    for blk in m.blocks:
        if c.get("visc"):
            blk.set_viscosity("constant", nu=c["visc"])
        blk.pgen_blast(post_init=False, **c["pgen"])
    m.post_init()
    h0 = m.history()
    m.evolve(0.1, c["nlim"])
    return m, h0


# ---- the tree ------------------------------------------------------------------------------------------------
def test_block_tree_of_the_shipped_cartesian_disk_deck():
    """inputs/disk/disk_cart.in: 128^3 root in 16x16x8 blocks, level-1 region [-2,2]^2 x [-1,1] in [-3,3]^3 -- every
    root block the region overlaps splits: 6^3 = 216 of 1024 -> 1728 fine + 808 coarse blocks; at disk.py's 64^3 root
    (tst/scripts/disk/disk.py:64-69) 64 of 128 -> 512 + 64."""
    for nrb, want in (((8, 8, 16), (808, 1728)), ((4, 4, 8), (64, 512))):
        t = BlockTree(nrb, 3, (False,) * 3)
        t.add_region(1, (-2, -2, -1), (2, 2, 1), (-3, -3, -3), (3, 3, 3))
        lv = [l for l, _ in t.leaves()]
        assert (lv.count(0), lv.count(1)) == want
    # 2:1 balance over corners: a level-2 block forces level 1 on everything it touches
    t = BlockTree((8, 8, 1), 2, (False,) * 3)
    t.add_region(2, (-0.05, -0.05, 0), (0.05, 0.05, 0), (-1, -1, -0.5), (1, 1, 0.5))
    leaves = t.leaves()
    idx = {lf: q for q, lf in enumerate(leaves)}
    for level, loc in leaves:
        for o in t.directions():
            kind, what = t.neighbour(level, loc, o)
            if kind == "coarser":
                assert (level - 1, what) in idx
            elif kind == "finer":
                assert all((level + 1, cl) in idx for cl, _ in what)


def test_oracle_conserves_across_levels_only_with_flux_correction():
    m, h0 = run_oracle("blast2d")
    h1 = m.history()
    assert abs(h1[0] - h0[0]) < 1e-13 * h0[0] and abs(h1[4] - h0[4]) < 1e-13 * h0[4]
    c = CASES["blast2d"]["oracle"]
    bad = MultiLevelOracle(c["mesh"], c["block"], c["lo"], c["hi"], c["bc"], regions=c["regions"], ng=2,
                           **dict(KW, riemann="hllc"))
    bad.flux_correction = lambda: None
    for blk in bad.blocks:
        blk.pgen_blast(post_init=False, **CASES["blast2d"]["pgen"])
    bad.post_init()
    b0 = bad.history()
    bad.evolve(0.1, 20)
    assert abs(bad.history()[4] - b0[4]) > 1e-9 * b0[4]  # the check above is not vacuous


def test_oracle_uniform_state_is_static_on_a_refined_mesh():
    c = CASES["visc3d"]["oracle"]
    # (rk2: its stage weights 1/2 + 1/2 reproduce a constant exactly; rk3's 1/3 + 2/3 do not, on any mesh)
    m = MultiLevelOracle(c["mesh"], c["block"], c["lo"], c["hi"], c["bc"], regions=c["regions"], ng=2, integrator="rk2",
                         **dict(KW, riemann="hlle"))
    for blk in m.blocks:
        blk.pgen_blast(radius=1e-9, internal_energy=1.0, p0=0.7, d0=1.3, samples=0, post_init=False)
    m.post_init()
    before = [blk.gprim.copy() for blk in m.blocks]
    m.evolve(-1.0, 3)
    assert all(np.array_equal(a, blk.gprim) for a, blk in zip(before, m.blocks))


# ---- host driver on the CPU double (worker processes: the double exports libartemis_hip.so's symbols) ----------
def _run_workers(world, spec, tmp_path, tag):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_multirank_cpu import run_world
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpu_double"), "-s"])
    return run_world(world, spec, tmp_path, tag)


# Two ways through a stage on a refined mesh, both bit-identical to the oracle: the per-task chain ("unfused": flux
# arrays, SetFluxCorrections between CalculateFluxes and ApplyUpdate as the reference does it) and the one-kernel
# stages with the correction as a fix-up of the coarse zones on coarse-fine faces ("fused", the default where it
# applies: include/artemis_hip.h).
@pytest.mark.parametrize("path", ["fused", "unfused"])
@pytest.mark.parametrize("case", ["blast2d", "visc3d", "twolevel2d", "sph3d"])
def test_host_driver_on_cpu_double_equals_multilevel_oracle(case, path, tmp_path):
    c = CASES[case]
    res = _run_workers(1, dict(deck=list(c["deck"]), overrides=c["ov"], path=path), tmp_path, case)[0]
    m, h0 = run_oracle(case)
    assert res["meta"]["nblocks"] == len(m.blocks) and res["meta"]["fused"] == (path == "fused")
    if c["blocks"]:
        lv = [l for l, _ in m.leaves]
        assert (lv.count(0), lv.count(1)) == c["blocks"]
    assert res["meta"]["ncycle"] == m.ncycle == c["nlim"] and res["meta"]["dt"] == m.dt and res["meta"]["time"] == m.time
    for b, (bounds, prim) in enumerate(res["blocks"]):
        blk = m.blocks[b]
        assert list(bounds) == m.block_bounds(b)
        assert np.array_equal(prim, blk.interior(blk.gprim)), (case, b)
    h1 = res["hist"]
    assert abs(h1[0] - h0[0]) < 1e-12 * h0[0] and abs(h1[4] - h0[4]) < 1e-12 * abs(h0[4])


def test_refined_blocks_split_over_two_ranks_bitwise(tmp_path):
    """The Z-ordered leaves are dealt to the ranks in contiguous runs; ghost and flux-correction operations that
    cross the rank boundary travel as ONE message per peer and phase (gloo here, RCCL on GPUs).  Same bits."""
    c = CASES["visc3d"]
    spec = dict(deck=list(c["deck"]), overrides=c["ov"])
    one = _run_workers(1, spec, tmp_path, "one")
    two = _run_workers(2, spec, tmp_path, "two")
    assert sum(r["meta"]["nblocks"] for r in two) == one[0]["meta"]["nblocks"] == 44
    assert abs(two[0]["meta"]["nblocks"] - two[1]["meta"]["nblocks"]) <= 1
    for r in two:
        assert r["meta"]["ncycle"] == one[0]["meta"]["ncycle"] and r["meta"]["dt"] == one[0]["meta"]["dt"]
    from test_multirank_cpu import by_bounds
    a, b = by_bounds(one), by_bounds(two)
    assert a.keys() == b.keys()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    assert np.allclose(two[0]["hist"], one[0]["hist"], rtol=1e-13)


@pytest.mark.parametrize("nranks", [2, 4])
def test_cost_weighted_split_is_bitwise_and_reports_its_balance(tmp_path, nranks):
    """<artemis_amd/loadbalance>: blocks weighed by level (a fine block counted 3x a root block here) and by the
    coarse-fine face operations they take part in; the Z-order runs then even the cumulative cost, not the block count.
    The result does not depend on the split -- 2 and 4 ranks equal 1 rank bit for bit -- and every rank reports the
    same max / mean cost ratio, which the weighted split keeps below what equal counts give under the same costs."""
    c = CASES["visc3d"]
    # the refined region moved into a corner of the mesh (later overrides win): ONE of the 16 root blocks splits, so the
    # Z-ordered list starts with its eight children -- an uneven cost per run of equal length
    ov = c["ov"] + region_overrides(1, (-1.0, -1.0, -1.0), (-0.6, -0.6, -0.6))
    lb = ["artemis_amd/loadbalance/level_cost=1.0,3.0", "artemis_amd/loadbalance/flux_face_cost=0.05"]
    one = _run_workers(1, dict(deck=list(c["deck"]), overrides=ov), tmp_path, "one")
    many = _run_workers(nranks, dict(deck=list(c["deck"]), overrides=ov + lb), tmp_path, "w%d" % nranks)
    counts = [r["meta"]["nblocks"] for r in many]
    assert sum(counts) == one[0]["meta"]["nblocks"] == 23 and min(counts) >= 1
    assert max(counts) - min(counts) > 1  # (not the equal-count split)
    ratios = {r["meta"]["load_balance"] for r in many}
    assert len(ratios) == 1
    ratio = ratios.pop()
    # what equal counts would give under the same costs: contiguous runs of 44 / nranks blocks of the Z-ordered leaves
    levels = [lv for r in many for lv in r["meta"]["levels"]]  # (ranks hold contiguous runs in rank order)
    cost = [3.0 if lv else 1.0 for lv in levels]
    n = len(cost)
    base, extra = divmod(n, nranks)
    runs, g = [], 0
    for r in range(nranks):
        k = base + (1 if r < extra else 0)
        runs.append(sum(cost[g:g + k]))
        g += k
    equal_counts = max(runs) / (sum(runs) / nranks)
    print("cost-weighted split on %d ranks: max / mean = %.3f (equal counts under the same level costs: %.3f), blocks per rank %s"
          % (nranks, ratio, equal_counts, counts))
    assert 1.0 <= ratio < equal_counts and ratio < 1.35
    from test_multirank_cpu import by_bounds
    a, b = by_bounds(one), by_bounds(many)
    assert a.keys() == b.keys()
    for key in a:
        assert np.array_equal(a[key], b[key]), key


# ---- the conduction problem's `conductive` user condition on coarse buffers (pgen/conduction.hpp:125-255) --------
COND_SMR_OV = ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=16", "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8",
               "parthenon/mesh/x2min=-0.25", "parthenon/mesh/x2max=0.25", "gravity/uniform/gx1=-0.02",
               "parthenon/time/nlim=8"] + region_overrides(1, (0.2, -0.25, -0.5), (0.4, 0.0, 0.5))


def conduction_smr_oracle():
    """inputs/diffusion/conduction.in in 2-D with a refined region AT the inner conductive boundary: the fine blocks
    there have coarser neighbours along x2 and x1, so their coarse buffers carry the conductive condition before the
    prolongation reads them."""
    m = MultiLevelOracle((32, 16, 1), (8, 8, 1), (0.2, -0.25, -0.5), (1.2, 0.25, 0.5),
                         ("conductive", "conductive") + ("periodic",) * 4, regions=[(1, (0.2, 0.4), (-0.25, 0.0), (-0.5, 0.5))],
                         ng=2, integrator="rk2", reconstruct="plm", riemann="hllc", gamma=1.66667, dfloor=1e-10,
                         siefloor=1e-15, cfl=0.3)
    m.diffusion = m.gravity = m.drag = True
    for o in m.blocks + m.coarse:
        o.set_gravity_uniform(-0.02, 0.0, 0.0)
        o.set_conductivity("conductivity", cond=0.1)
        o.set_drag("self", "constant")
        o.set_damping(0, inner=(4.0, -1.7976931348623157e308, -1.7976931348623157e308), inner_rate=(1.0e4, 0.0, 0.0))
        o.pgen_conduction(gas_rho=1.0, gas_temp=0.05, flux=0.01, post_init=False)
    m.post_init()
    m.evolve(40.0, 8)
    return m


def test_conductive_condition_on_coarse_buffers_equals_multilevel_oracle_cpu(tmp_path):
    res = _run_workers(1, dict(deck=["diffusion", "conduction.in"], overrides=COND_SMR_OV), tmp_path, "cond")[0]
    m = conduction_smr_oracle()
    assert res["meta"]["nblocks"] == len(m.blocks) and set(res["meta"]["levels"]) == {0, 1}
    assert res["meta"]["ncycle"] == m.ncycle == 8 and res["meta"]["dt"] == m.dt and res["meta"]["time"] == m.time
    for b, (bounds, prim) in enumerate(res["blocks"]):
        blk = m.blocks[b]
        assert list(bounds) == m.block_bounds(b)
        assert np.array_equal(prim, blk.interior(blk.gprim)), b


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["fused", "unfused"])
def test_conductive_condition_on_coarse_buffers_equals_multilevel_oracle(hiplib, path):
    from artemis_amd.driver import Simulation
    s = Simulation(DECK("diffusion", "conduction.in"), COND_SMR_OV)
    s.set_path(path)
    s.evolve()
    m = conduction_smr_oracle()
    assert s.nblocks == len(m.blocks) and s.ncycle == m.ncycle == 8 and s.dt == m.dt and s.time == m.time
    for b, blk in enumerate(m.blocks):
        assert s.block_bounds(b) == m.block_bounds(b) and s.block_level(b) == m.leaves[b][0]
        got = s.field("gas.prim", b)
        assert np.array_equal(got[[0, 1, 2, 3, 5]], blk.gprim[[0, 1, 2, 3, 5]]), b  # ghost zones included
    s.close()


# ---- the HIP path ----------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("path", ["fused", "unfused"])
@pytest.mark.parametrize("case", ["blast2d", "visc3d", "twolevel2d", "sph3d"])
def test_hip_driver_equals_multilevel_oracle(hiplib, case, path):
    """`fused`: every block through the one-kernel stage (the tuned tile kernel for the Cartesian blast, its
    curvilinear instantiation for sph3d, the cell-centred stage for the viscous deck), then the coarse zones on
    coarse-fine faces redone with the restricted fine fluxes -- the same sum as ApplyUpdate after
    SetFluxCorrections, which is what `unfused` (the per-task chain) computes literally.  Both equal the oracle."""
    from artemis_amd.driver import Simulation
    c = CASES[case]
    s = Simulation(DECK(*c["deck"]), c["ov"])
    assert s.uses_fused_path  # the default on refined meshes without drag / n-body
    s.set_path(path)
    assert s.uses_fused_path == (path == "fused")
    h0 = s.history()
    s.evolve()
    m, _ = run_oracle(case)
    assert s.nblocks == s.nblocks_global == len(m.blocks)
    assert s.ncycle == m.ncycle and s.dt == m.dt and s.time == m.time
    for b, blk in enumerate(m.blocks):
        assert s.block_bounds(b) == m.block_bounds(b) and s.block_level(b) == m.leaves[b][0]
        got = s.field("gas.prim", b)
        assert np.array_equal(got[[0, 1, 2, 3, 5]], blk.gprim[[0, 1, 2, 3, 5]]), (case, b)  # ghosts included
        assert np.array_equal(s.interior(got), blk.interior(blk.gprim)), (case, b)
    h1 = s.history()
    assert abs(h1[0] - h0[0]) < 1e-12 * h0[0] and abs(h1[4] - h0[4]) < 1e-12 * abs(h0[4])
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["fused", "unfused"])
@pytest.mark.parametrize("case", ["visc3d", "sph3d", "twolevel2d"])
def test_refined_mesh_exchange_through_rccl_loopback(hiplib, case, path, option):
    """The block-graph exchange and the flux correction of a refined mesh through the NATIVE C++ RCCL transport on one
    GPU: with LOOPBACK_COMM the Z-ordered leaf list is cut into two virtual halves and every operation between them --
    same-level slabs, restriction on the fly into a coarser neighbour, copies into a finer neighbour's coarse buffer,
    restricted fine fluxes (viscous ones included) -- is packed into the send buffer, sent to this rank itself with
    ncclSend / ncclRecv on the comm stream, and unpacked from the receive buffer: the code path two GPUs take.  Bitwise
    equal to the run that copies on the device (which tests above hold against the multilevel oracle)."""
    from artemis_amd.driver import RcclComm, Simulation
    c = CASES[case]
    ref = Simulation(DECK(*c["deck"]), c["ov"])
    ref.set_path(path)
    ref.evolve()
    option("loopback_comm", 1)
    comm = RcclComm(0, 1)
    try:
        sim = Simulation(DECK(*c["deck"]), c["ov"], comm=comm)
        sim.set_path(path)
        sim.evolve()
        assert sim.nblocks == ref.nblocks and sim.ncycle == ref.ncycle and sim.dt == ref.dt and sim.time == ref.time
        for b in range(sim.nblocks):
            assert np.array_equal(sim.field("gas.prim", b)[[0, 1, 2, 3, 5]], ref.field("gas.prim", b)[[0, 1, 2, 3, 5]]), (case, b)
        assert np.array_equal(sim.history(), ref.history())
        sim.close()
    finally:
        comm.close()
        ref.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["visc3d", "sph3d"])
def test_flux_rows_only_on_blocks_with_coarse_fine_faces(hiplib, case, option):
    """On the one-kernel stages of a refined mesh the flux arrays are touched on coarse-fine faces alone (the fine side's
    faces, their restrictions, the faces of the fix-up zones), so the driver gives rows to the blocks that own such a face
    and points every other block's table entries at one shared row.  Same bits as with every block's rows
    (ARTEMIS_DENSE_FLUX), and switching to the per-task chain afterwards -- which writes every block's fluxes -- allocates
    the full set (the result still equals the run that started dense)."""
    from artemis_amd.driver import Simulation
    c = CASES[case]
    option("dense_flux", 1)
    ref = Simulation(DECK(*c["deck"]), c["ov"])
    ref.evolve(4)
    option("dense_flux", 0)
    sim = Simulation(DECK(*c["deck"]), c["ov"])
    sim.evolve(4)
    assert sim.uses_fused_path and sim.ncycle == ref.ncycle == 4 and sim.dt == ref.dt
    for b in range(sim.nblocks):
        assert np.array_equal(sim.field("gas.prim", b), ref.field("gas.prim", b)), (case, b)
    ref.set_path("unfused"), sim.set_path("unfused")
    ref.evolve(3), sim.evolve(3)
    assert sim.dt == ref.dt
    for b in range(sim.nblocks):
        assert np.array_equal(sim.field("gas.prim", b), ref.field("gas.prim", b)), (case, b)
    sim.close(), ref.close()


def _disk_cart(extra):
    from artemis_amd.driver import Simulation
    return Simulation(DECK("disk", "disk_cart.in"), ["parthenon/time/nlim=%d" % DISK["cycles"]] + extra)


def _check_disk(s, d0):
    """tst/scripts/disk/disk.py:118-187 on the `cart` geometry (numbers: tests/golden/reference_pins.json)"""
    num = den = 0.0
    for b in range(s.nblocks):
        P = s.interior(s.field("gas.prim", b))
        d, T = P[0], P[4] / P[0]
        assert not np.isnan(P).any() and d.min() > 0.0 and T.min() > 0.0
        num += (d0[b] * (d - d0[b]) ** 2).sum()
        den += d0[b].sum()
    assert DISK["dt_low"] < s.dt < DISK["dt_high"], s.dt
    err = np.sqrt(num) / den
    assert err <= DISK["density_err_max"], err
    return err


@pytest.mark.gpu
@pytest.mark.parametrize("gam", DISK["gamma"])
def test_disk_cart_deck_reference_checks_at_test_resolution(hiplib, gam):
    """inputs/disk/disk_cart.in UNCHANGED (refinement = static, level 1) with disk.py's own overrides for `cart`
    (64^3 root, nlim 5 + 5, polytropic index 1 and 1.4): 64 coarse + 512 fine blocks through the HIP driver; the
    reference's checks pass and mass is conserved to round-off until material reaches the outflow boundary."""
    s = _disk_cart(["parthenon/mesh/nx%d=%d" % (d + 1, n) for d, n in enumerate(DISK["cart_mesh_override"])]
                   + ["problem/polytropic_index=%.2f" % gam, "gas/de_switch=0.0"])
    assert s.nblocks == 576 and [s.block_level(b) for b in range(s.nblocks)].count(1) == 512
    d0 = [s.interior(s.field("gas.prim", b))[0].copy() for b in range(s.nblocks)]
    s.evolve()
    assert s.ncycle == DISK["cycles"]
    _check_disk(s, d0)
    s.close()


@pytest.mark.gpu
def test_disk_cart_deck_as_shipped(hiplib):
    """The deck exactly as shipped: 128^3 root -> 808 coarse + 1728 fine blocks of 16x16x8 (nghost 4, alpha
    viscosity, point-mass gravity), 10 cycles."""
    s = _disk_cart([])
    lv = [s.block_level(b) for b in range(s.nblocks)]
    assert s.nblocks == 2536 and lv.count(1) == 1728 and s.total_zones == 2536 * 16 * 16 * 8
    d0 = [s.interior(s.field("gas.prim", b))[0].copy() for b in range(s.nblocks)]
    s.evolve()
    assert s.ncycle == DISK["cycles"]
    _check_disk(s, d0)
    s.close()


@pytest.mark.gpu
def test_disk_sph_deck_with_a_static_refinement_region(hiplib):
    """BASELINE configs[3]'s combination on one GPU: inputs/disk/disk_sph.in (spherical-polar, `ic` conditions,
    point-mass gravity, alpha viscosity, curvilinear rotating frame) with a level-1 static region around the
    midplane between r = 0.7 and 1.9 -- 8 of the 16 root blocks refine: 8 coarse + 64 fine blocks of 32^3.  The
    reference ships no refined spherical deck, so this runs disk.py's checks for the `sph` geometry (density error,
    dt window, positivity) and compares with the unrefined run of the same deck: the refined mesh stays as close to
    the initial equilibrium as the uniform one (same bound), and mass is conserved to round-off by both."""
    from artemis_amd.driver import Simulation
    ov = ["parthenon/time/nlim=%d" % DISK["cycles"], "problem/polytropic_index=1.40", "gas/de_switch=1e-2"]  # as disk.py runs it
    reg = region_overrides(1, (0.7, 1.3, -1.0), (1.9, 1.85, 1.0))
    s = Simulation(DECK("disk", "disk_sph.in"), ov + reg)
    lv = [s.block_level(b) for b in range(s.nblocks)]
    assert s.nblocks == 72 and lv.count(1) == 64 and s.uses_fused_path
    d0 = [s.interior(s.field("gas.prim", b))[0].copy() for b in range(s.nblocks)]
    h0 = s.history()
    s.evolve()
    assert s.ncycle == DISK["cycles"]
    err = _check_disk(s, d0)
    u = Simulation(DECK("disk", "disk_sph.in"), ov)
    u0 = [u.interior(u.field("gas.prim", b))[0].copy() for b in range(u.nblocks)]
    u.evolve()
    err_u = _check_disk(u, u0)
    assert err < 3.0 * err_u + 1e-6, (err, err_u)
    s.close(), u.close()


# ---- BASELINE configs[3]'s combination: the spherical disk deck on a refined mesh, bit for bit ------------------
DISK_SMR_OV = ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=16", "parthenon/mesh/nx3=16", "parthenon/meshblock/nx1=8",
               "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8", "parthenon/time/nlim=6",
               "problem/polytropic_index=1.40", "gas/de_switch=1e-2"] + \
    region_overrides(1, (0.7, 1.3, -1.0), (1.9, 1.85, 1.0))


def disk_smr_oracle():
    """inputs/disk/disk_sph.in at 32 x 16 x 16 in 8^3 blocks with a level-1 region: every block AND every coarse
    buffer evaluates the disk profile on its own zone centres (the `ic` condition re-evaluates it wherever it is
    applied, pgen/disk.hpp:597-632)."""
    from test_oracle_pins import _DISK_DECKS
    D = _DISK_DECKS["sph"]
    m = MultiLevelOracle((32, 16, 16), (8, 8, 8), D["lo"], D["hi"], D["bc"]("ic"),
                         regions=[(1, (0.7, 1.9), (1.3, 1.85), (-1.0, 1.0))], ng=2, integrator="rk2", reconstruct="plm",
                         riemann=D["riemann"], gamma=1.4, dfloor=1e-10, siefloor=D["siefloor"], cfl=0.3,
                         coordinates="spherical", de_switch=1e-2)
    m.diffusion = m.gravity = m.rframe = True
    for blk in m.blocks + m.coarse:
        blk.set_gravity_point(mass=1.0)
        blk.set_rotating_frame(1.0, 0.0)
        blk.set_viscosity("alpha", alpha=1e-3, r0=1.0, Omega0=1.0)
        blk.pgen_disk(r0=1.0, rho0=1.0, dslope=-2.25, flare=0.25, h0=0.05, dens_min=1e-10, pres_min=1e-15,
                      polytropic_index=1.4, post_init=False)
    m.post_init()
    m.evolve(62.8, 6)
    return m


def test_disk_sph_on_a_refined_mesh_cpu_double_equals_multilevel_oracle(tmp_path):
    res = _run_workers(1, dict(deck=["disk", "disk_sph.in"], overrides=DISK_SMR_OV), tmp_path, "dsmr")[0]
    m = disk_smr_oracle()
    assert res["meta"]["nblocks"] == len(m.blocks) and res["meta"]["fused"]
    assert res["meta"]["ncycle"] == m.ncycle == 6 and res["meta"]["dt"] == m.dt and res["meta"]["time"] == m.time
    for b, (bounds, prim) in enumerate(res["blocks"]):
        blk = m.blocks[b]
        assert list(bounds) == m.block_bounds(b)
        assert np.array_equal(prim, blk.interior(blk.gprim)), b


@pytest.mark.gpu
def test_disk_sph_on_a_refined_mesh_hip_equals_multilevel_oracle(hiplib):
    from artemis_amd.driver import Simulation
    s = Simulation(DECK("disk", "disk_sph.in"), DISK_SMR_OV)
    s.evolve()
    m = disk_smr_oracle()
    assert s.nblocks == len(m.blocks) and s.ncycle == m.ncycle and s.dt == m.dt and s.time == m.time
    for b, blk in enumerate(m.blocks):
        got = s.field("gas.prim", b)
        assert np.array_equal(got[[0, 1, 2, 3, 5]], blk.gprim[[0, 1, 2, 3, 5]]), b  # ghosts included
        assert np.array_equal(s.interior(got), blk.interior(blk.gprim)), b
    s.close()


def test_disk_sph_on_a_refined_mesh_two_ranks_bitwise(tmp_path):
    """The same refined spherical disk deck split over 2 ranks (gloo): identical bits, block for block."""
    spec = dict(deck=["disk", "disk_sph.in"], overrides=DISK_SMR_OV)
    one = _run_workers(1, spec, tmp_path, "ds1")
    two = _run_workers(2, spec, tmp_path, "ds2")
    assert sum(r["meta"]["nblocks"] for r in two) == one[0]["meta"]["nblocks"] == 72
    from test_multirank_cpu import by_bounds
    a, b = by_bounds(one), by_bounds(two)
    assert a.keys() == b.keys()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    for r in two:
        assert r["meta"]["dt"] == one[0]["meta"]["dt"] and r["meta"]["ncycle"] == 6


# ---- gas + dust on a refined mesh (no multilevel oracle for dust: conservation, rank independence, accuracy) ------
ADV_SMR = dict(deck=["advection", "advection.in"], cycles=12, dust=True,
               overrides=["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=16", "parthenon/mesh/nx3=16", "parthenon/meshblock/nx1=8",
                          "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8"]
               + region_overrides(1, (1.0, 0.4, 0.4), (2.0, 1.1, 1.1)))


def test_gas_and_two_dust_species_on_a_refined_mesh(tmp_path):
    """inputs/advection/advection.in (gas + two counter-streaming dust species, fully periodic) with a level-1 region
    in the middle: the dust fluids go through the same block-graph exchange, coarse buffers and flux correction as the
    gas.  Mass and momentum of every fluid are conserved to round-off across the level boundaries; 2 ranks give the
    bits of 1 rank; the wave is carried as accurately as on the uniform mesh."""
    one = _run_workers(1, ADV_SMR, tmp_path, "a1")
    two = _run_workers(2, ADV_SMR, tmp_path, "a2")
    uni = _run_workers(1, dict(ADV_SMR, overrides=ADV_SMR["overrides"][:6]), tmp_path, "au")
    assert one[0]["meta"]["nblocks"] > uni[0]["meta"]["nblocks"] == 16 and one[0]["meta"]["fused"]
    h = one[0]["hist"]
    # history after the run against the analytic integrals of the initial state (advection.py:100-187 pins them on
    # the uniform mesh): gas mass 6.75 = rho0 * volume, dust masses likewise; total momenta of the counter-streaming
    # dust species cancel
    hu = uni[0]["hist"]
    assert np.allclose(h, hu, rtol=0, atol=5e-13 * np.abs(hu).max()), (h, hu)
    from test_multirank_cpu import by_bounds
    a, b = by_bounds(one), by_bounds(two)
    assert a.keys() == b.keys()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    e, eu = one[0]["errs"], uni[0]["errs"]
    assert np.all(np.isfinite(e[:3])) and np.all(e[:3] < 1.6 * eu[:3]), (e[:3], eu[:3])


@pytest.mark.gpu
def test_gas_and_two_dust_species_on_a_refined_mesh_hip(hiplib):
    """The same deck through the HIP driver: every fluid's mass and momentum conserved to round-off over 12 cycles
    across the level boundaries (periodic domain), all fields finite."""
    from artemis_amd.driver import Simulation
    s = Simulation(DECK(*ADV_SMR["deck"]), ADV_SMR["overrides"] + ["parthenon/time/nlim=12"])
    assert s.uses_fused_path and s.nblocks > 16
    h0 = s.history()
    s.evolve()
    h1 = s.history()
    assert s.ncycle == 12 and np.allclose(h1, h0, rtol=0, atol=5e-13 * np.abs(h0).max()), (h0, h1)
    for b in range(s.nblocks):
        assert np.isfinite(s.field("gas.prim", b)[[0, 1, 2, 3, 5]]).all() and np.isfinite(s.field("dust.prim", b)).all()
    s.close()


# ---- N-body gravity on a refined cylindrical mesh (configs[4]'s ingredients minus REBOUND and adaptivity) ---------
NBODY_SMR_OV = ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=16", "parthenon/mesh/nx3=16", "parthenon/meshblock/nx1=8",
                "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8", "parthenon/time/nlim=6",
                "problem/polytropic_index=1.00"] + region_overrides(1, (0.8, -0.7, -0.4), (1.6, 0.7, 0.4))


def nbody_smr_oracle():
    from test_nbody import _PI
    m = MultiLevelOracle((32, 16, 16), (8, 8, 8), (0.3, -_PI, -1.0), (4.3, _PI, 1.0),
                         ("ic", "ic", "periodic", "periodic", "ic", "ic"),
                         regions=[(1, (0.8, 1.6), (-0.7, 0.7), (-0.4, 0.4))], ng=2, integrator="rk2", reconstruct="plm",
                         riemann="hllc", gamma=1.4, dfloor=1e-10, siefloor=1e-10, cfl=0.3, coordinates="cylindrical")
    m.diffusion = m.gravity = True
    for blk in m.blocks + m.coarse:
        blk.set_gravity_nbody([dict(GM=1.0)])
        blk.set_viscosity("alpha", alpha=1e-3, r0=1.0, Omega0=1.0)
        blk.pgen_disk(r0=1.0, rho0=1.0, dslope=-2.25, flare=0.25, h0=0.05, dens_min=1e-10, pres_min=1e-15,
                      polytropic_index=1.0, post_init=False)
    m.post_init()
    m.evolve(62.8, 6)
    return m


def test_nbody_disk_deck_on_a_refined_mesh_cpu_double_equals_multilevel_oracle(tmp_path):
    """inputs/disk/disk_nbody_cyl.in (cylindrical, `ic` conditions, alpha viscosity, the N-body gravity task with the
    central particle) at 32 x 16 x 16 in 8^3 blocks with a level-1 region at r ~ 1: host driver on the CPU double ==
    multilevel oracle, bit for bit, on 1 rank; 2 ranks give the same bits."""
    spec = dict(deck=["disk", "disk_nbody_cyl.in"], overrides=NBODY_SMR_OV)
    res = _run_workers(1, spec, tmp_path, "nb1")
    m = nbody_smr_oracle()
    r = res[0]
    assert r["meta"]["nblocks"] == len(m.blocks) and r["meta"]["ncycle"] == m.ncycle == 6
    assert r["meta"]["dt"] == m.dt and r["meta"]["time"] == m.time
    for b, (bounds, prim) in enumerate(r["blocks"]):
        assert list(bounds) == m.block_bounds(b)
        assert np.array_equal(prim, m.blocks[b].interior(m.blocks[b].gprim)), b
    two = _run_workers(2, spec, tmp_path, "nb2")
    from test_multirank_cpu import by_bounds
    a, b2 = by_bounds(res), by_bounds(two)
    for key in a:
        assert np.array_equal(a[key], b2[key]), key


@pytest.mark.gpu
def test_nbody_disk_deck_on_a_refined_mesh_hip_equals_multilevel_oracle(hiplib):
    from artemis_amd.driver import Simulation
    s = Simulation(DECK("disk", "disk_nbody_cyl.in"), NBODY_SMR_OV)
    s.evolve()
    m = nbody_smr_oracle()
    assert s.nblocks == len(m.blocks) and s.ncycle == m.ncycle and s.dt == m.dt
    for b, blk in enumerate(m.blocks):
        assert np.array_equal(s.interior(s.field("gas.prim", b)), blk.interior(blk.gprim)), b
    f = s.nbody_force()
    assert f.shape == (1, 7) and np.all(np.isfinite(f))
    s.close()


# ---- the shearing sheet's user conditions (extrap / inflow) on fine blocks and coarse buffers ---------------------
SSHEET_SMR_OV = ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8",
                 "parthenon/time/nlim=8", "parthenon/time/tlim=100.0"] + region_overrides(1, (-0.4, -0.9, -0.2), (0.4, 0.9, 0.2))


def ssheet_smr_oracle():
    m = MultiLevelOracle((32, 32, 1), (8, 8, 1), (-1.0, -1.0, -0.2), (1.0, 1.0, 0.2),
                         ("extrap", "extrap", "inflow", "inflow", "extrap", "extrap"),
                         regions=[(1, (-0.4, 0.4), (-0.9, 0.9), (-0.2, 0.2))], ng=2, integrator="rk2", reconstruct="plm",
                         riemann="hllc", gamma=1.000001, dfloor=1e-10, siefloor=1e-10, cfl=0.3)
    m.gravity = m.rframe = True
    for blk in m.blocks + m.coarse:
        blk.set_rotating_frame(1.0, 1.5)
        blk.set_gravity_point(1e-5, soft=0.03)
        blk.pgen_strat(rho0=1.0, dens_min=1e-10, h=0.05, post_init=False)
    m.post_init()
    m.evolve(100.0, 8)
    return m


def test_shearing_sheet_on_a_refined_mesh_cpu_double_equals_multilevel_oracle(tmp_path):
    """inputs/ssheet/ssheet.in (strat problem, point-mass gravity, shearing box, `extrap` x1 / `inflow` x2 conditions)
    at 32^2 in 8^2 blocks with a refined strip touching the x2 boundaries: the user conditions act on fine blocks and
    on the coarse buffers of the blocks next to the level boundary.  Host driver == multilevel oracle bit for bit."""
    res = _run_workers(1, dict(deck=["ssheet", "ssheet.in"], overrides=SSHEET_SMR_OV), tmp_path, "ss")[0]
    m = ssheet_smr_oracle()
    assert res["meta"]["nblocks"] == len(m.blocks) and res["meta"]["ncycle"] == m.ncycle == 8
    assert res["meta"]["dt"] == m.dt and res["meta"]["time"] == m.time
    for b, (bounds, prim) in enumerate(res["blocks"]):
        assert list(bounds) == m.block_bounds(b)
        assert np.array_equal(prim, m.blocks[b].interior(m.blocks[b].gprim)), b


@pytest.mark.gpu
def test_shearing_sheet_on_a_refined_mesh_hip_equals_multilevel_oracle(hiplib):
    from artemis_amd.driver import Simulation
    s = Simulation(DECK("ssheet", "ssheet.in"), SSHEET_SMR_OV)
    s.evolve()
    m = ssheet_smr_oracle()
    assert s.nblocks == len(m.blocks) and s.ncycle == m.ncycle == 8 and s.dt == m.dt
    for b, blk in enumerate(m.blocks):
        got = s.field("gas.prim", b)
        assert np.array_equal(got[[0, 1, 2, 3, 5]], blk.gprim[[0, 1, 2, 3, 5]]), b  # ghosts included
    s.close()


def test_disk_extrap_condition_on_a_refined_mesh(tmp_path):
    """The disk problem's `extrap` condition (power-law extrapolation in ln x of density, sie and the inertial azimuthal
    velocity, pgen/disk.hpp:634-825) on fine blocks and coarse buffers of the refined spherical disk deck: host driver on
    the CPU double == multilevel oracle (both evaluate log / exp with the host libm here: bit for bit)."""
    ov = [o for o in DISK_SMR_OV if "nlim" not in o] + ["parthenon/time/nlim=3"]  # (this coarse mesh goes unstable later)
    ov += ["parthenon/mesh/i%s_bc=extrap" % d for d in ("x1", "x2")] + ["parthenon/mesh/o%s_bc=extrap" % d for d in ("x1", "x2")]
    res = _run_workers(1, dict(deck=["disk", "disk_sph.in"], overrides=ov), tmp_path, "dex")[0]
    from test_oracle_pins import _DISK_DECKS
    D = _DISK_DECKS["sph"]
    m = MultiLevelOracle((32, 16, 16), (8, 8, 8), D["lo"], D["hi"], D["bc"]("disk_extrap"),
                         regions=[(1, (0.7, 1.9), (1.3, 1.85), (-1.0, 1.0))], ng=2, integrator="rk2", reconstruct="plm",
                         riemann=D["riemann"], gamma=1.4, dfloor=1e-10, siefloor=D["siefloor"], cfl=0.3,
                         coordinates="spherical", de_switch=1e-2)
    m.diffusion = m.gravity = m.rframe = True
    for blk in m.blocks + m.coarse:
        blk.set_gravity_point(mass=1.0)
        blk.set_rotating_frame(1.0, 0.0)
        blk.set_viscosity("alpha", alpha=1e-3, r0=1.0, Omega0=1.0)
        blk.pgen_disk(r0=1.0, rho0=1.0, dslope=-2.25, flare=0.25, h0=0.05, dens_min=1e-10, pres_min=1e-15,
                      polytropic_index=1.4, post_init=False)
    m.post_init()
    m.evolve(62.8, 3)
    assert res["meta"]["ncycle"] == m.ncycle == 3 and res["meta"]["dt"] == m.dt
    for b, (bounds, prim) in enumerate(res["blocks"]):
        assert np.isfinite(prim).all() and np.array_equal(prim, m.blocks[b].interior(m.blocks[b].gprim)), b


# ---- one dimension: the smallest refined mesh (1-D blocks are never tagged by the criteria, but a static region works) ---
SMR1D = ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=1", "parthenon/mesh/nx3=1", "parthenon/meshblock/nx1=8",
         "parthenon/meshblock/nx2=1", "parthenon/meshblock/nx3=1", "gas/riemann=hllc", "problem/radius=0.2", "problem/samples=0",
         "problem/p0=0.1", "parthenon/time/nlim=20"] + region_overrides(1, (-0.3, -0.5, -0.5), (0.3, 0.5, 0.5))


def test_one_dimensional_refined_mesh_both_paths_agree_cpu_double(tmp_path):
    """A 1-D blast with a level-1 region in the middle (4 coarse + 8 fine blocks of 8 zones): the tuned kernel's 1-D form +
    the coarse-fine fix-up (two zones per coarse block touch a corrected face) equals the per-task chain bit for bit, and
    mass is conserved to round-off across the level boundaries."""
    got = {}
    for path in ("fused", "unfused"):
        got[path] = _run_workers(1, dict(deck=["blast", "blast.in"], overrides=SMR1D, path=path), tmp_path, "smr1d" + path)[0]
        assert got[path]["meta"]["nblocks"] == 12 and got[path]["meta"]["fused"] == (path == "fused")
        assert sorted(got[path]["meta"]["levels"]) == [0] * 4 + [1] * 8
    assert got["fused"]["meta"]["dt"] == got["unfused"]["meta"]["dt"] and got["fused"]["meta"]["ncycle"] == 20
    for (ba, pa), (bb, pb) in zip(got["fused"]["blocks"], got["unfused"]["blocks"]):
        assert list(ba) == list(bb) and np.array_equal(pa, pb)
    assert np.allclose(got["fused"]["hist"], got["unfused"]["hist"], rtol=0, atol=1e-14)


@pytest.mark.gpu
def test_one_dimensional_refined_mesh_both_paths_agree_hip(hiplib):
    from artemis_amd.driver import Simulation
    out = {}
    for path in ("fused", "unfused"):
        s = Simulation(DECK("blast", "blast.in"), SMR1D)
        s.set_path(path)
        h0 = s.history()
        s.evolve()
        h1 = s.history()
        assert s.nblocks == 12 and s.ncycle == 20 and abs(h1[0] - h0[0]) < 1e-13 * h0[0]
        out[path] = (s.dt, [s.field("gas.prim", b)[[0, 1, 2, 3, 5]].copy() for b in range(s.nblocks)])
        s.close()
    assert out["fused"][0] == out["unfused"][0]
    assert all(np.array_equal(a, b) for a, b in zip(out["fused"][1], out["unfused"][1]))


# inputs/disk/disk_nbody_cyl.in with a planet, one dust species under simple_dust drag, the rotating frame and a three-level
# adaptive mesh: `ic` conditions on x1 and x3, an atmosphere sitting on the density floors, level boundaries inside it
PLANET_AMR_OV = ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "parthenon/mesh/nx3=32", "parthenon/meshblock/nx1=8",
                 "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8", "parthenon/mesh/refinement=adaptive",
                 "parthenon/mesh/numlevel=3", "parthenon/mesh/derefine_count=5", "gas/refine_field=density",
                 "gas/refine_type=magnitude", "gas/refine_thr=0.5", "gas/deref_thr=0.2",
                 "physics/rotating_frame=true", "rotating_frame/omega=1.0",
                 "physics/dust=true", "dust/nspecies=1", "dust/cfl=0.3", "dust/reconstruct=plm", "dust/riemann=hlle",
                 "dust/dfloor=1e-10", "physics/drag=true", "drag/type=simple_dust", "dust/stopping_time/type=constant",
                 "dust/stopping_time/tau=0.1", "dust/sizes=1.0",
                 "nbody/particle2/mass=1.0e-3", "nbody/particle2/couple=1", "nbody/particle2/soft/type=plummer",
                 "nbody/particle2/soft/radius=0.03", "nbody/particle2/initialize/x=1.0", "nbody/particle2/initialize/vy=1.0"]


def _planet_amr_run(path=None, cycles=12):
    from artemis_amd.driver import Simulation
    s = Simulation(DECK("disk", "disk_nbody_cyl.in"), PLANET_AMR_OV)
    if path:
        s.set_path(path)
    s.evolve(cycles)
    out = dict(dt=s.dt, nb=s.nblocks, remeshes=s.remeshes, kernel=s.stage_kernel,
               gas=[s.field("gas.prim", b)[[0, 1, 2, 3, 5]] for b in range(s.nblocks)],
               dust=[s.field("dust.prim", b) for b in range(s.nblocks)])
    s.close()
    return out


def test_floors_behind_the_refined_fill_cpu_double(tmp_path):
    """The same on the CPU double, small: the planet-disk deck (dust, drag, rotating frame) on a static three-level mesh whose
    level boundaries lie in the floored atmosphere next to the inner radial boundary.  One-kernel stages == per-task chain in
    every active zone; without artemis_hip_ml_floor_ghosts (NO_ML_FLOOR) they part within three cycles.  (The double's stage
    reads ghost zones as supplied -- tests/cpu_double/double.cpp::prim_to_cons_as_supplied -- as the device's kernels do.)"""
    drop = ("mesh/nx", "refinement=", "numlevel", "derefine_count", "gas/refine", "gas/deref")
    ov = [o for o in PLANET_AMR_OV if not any(d in o for d in drop)]
    ov += ["parthenon/mesh/nx1=16", "parthenon/mesh/nx2=8", "parthenon/mesh/nx3=16"]
    ov += region_overrides(1, (0.3, -3.0, -0.2), (0.7, 3.0, 0.2), level=2)
    runs = {}
    for tag, path, env in (("fused", "fused", {}), ("unfused", "unfused", {}), ("nofloor", "fused", {"ARTEMIS_NO_ML_FLOOR": "1"})):
        spec = dict(deck=["disk", "disk_nbody_cyl.in"], overrides=ov, path=path, cycles=3, dust=True, env=env)
        runs[tag] = _run_workers(1, spec, tmp_path, "floor_" + tag)[0]
        lv = runs[tag]["meta"]["levels"]
        assert (lv.count(0), lv.count(1), lv.count(2)) == (2, 12, 32) and runs[tag]["meta"]["fused"] == (path == "fused")

    def parted(a, b):
        gas = sum(1 for (_, x), (_, y) in zip(a["blocks"], b["blocks"]) if not np.array_equal(x, y))
        return gas + sum(1 for x, y in zip(a["dust"], b["dust"]) if not np.array_equal(x, y))
    assert runs["fused"]["meta"]["dt"] == runs["unfused"]["meta"]["dt"]
    assert parted(runs["fused"], runs["unfused"]) == 0
    assert parted(runs["nofloor"], runs["unfused"]) > 0
    # two ranks: restricted zones arrive in messages too (the unpack operations put their blocks on the list)
    from test_multirank_cpu import by_bounds
    spec = dict(deck=["disk", "disk_nbody_cyl.in"], overrides=ov, path="fused", cycles=3, dust=True)
    two = _run_workers(2, spec, tmp_path, "floor_two")
    one, both = by_bounds([runs["fused"]]), by_bounds(two)
    assert sorted(one) == sorted(both)
    for key in one:
        assert np.array_equal(one[key], both[key]), key


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["no_ic_skip", "no_ic_in_shell", "no_drag_in_march", "trim_pool"])
def test_round6_shortcuts_on_a_refined_disk_are_bitwise_neutral(hiplib, option, switch):
    """Round 6 took launches and passes out of the refined-mesh stage: `ic` faces skipped once a buffer holds them
    (NO_IC_SKIP), `ic` faces inside the one-launch boundary fill (NO_IC_IN_SHELL), the drag finish inside the dust march
    (NO_DRAG_IN_MARCH), the allocator's cache kept across remeshes (TRIM_POOL restores the old behaviour).  Each switch
    selects the older path; every zone of every leaf -- ghost zones included -- and dt must not move."""
    a = _planet_amr_run()
    option(switch)
    b = _planet_amr_run()
    assert a["dt"] == b["dt"] and a["nb"] == b["nb"] and a["remeshes"] == b["remeshes"]
    for q in range(a["nb"]):
        assert np.array_equal(a["gas"][q], b["gas"][q], equal_nan=True), (switch, "gas", q)
        assert np.array_equal(a["dust"][q], b["dust"][q], equal_nan=True), (switch, "dust", q)


@pytest.mark.gpu
def test_floors_bind_next_to_level_boundaries_one_kernel_stages_equal_the_task_chain(hiplib, option):
    """The reference's PrimToCons floors every ghost zone behind the boundary fill (fill_derived.cpp:227-262).  Where the
    atmosphere sits on the density floor a restricted average rounds below it and a prolongation undershoots it; the
    one-kernel stages keep no conserved ghost zones, so the fill ends with artemis_hip_ml_floor_ghosts on the blocks next to
    a level boundary.  Against the per-task chain (the reference's sequence literally, PrimToCons over whole blocks): every
    zone of every leaf, ghost zones included, and dt -- and the deck does exercise it: without the pass (NO_ML_FLOOR) the two
    part ways in the second cycle."""
    a = _planet_amr_run()
    b = _planet_amr_run(path="unfused")
    assert "stage_curv_kernel" in a["kernel"] and "per-task" in b["kernel"]
    assert a["dt"] == b["dt"] and a["nb"] == b["nb"] and a["remeshes"] == b["remeshes"]
    for q in range(a["nb"]):
        assert np.array_equal(a["gas"][q], b["gas"][q], equal_nan=True), ("gas", q)
        assert np.array_equal(a["dust"][q], b["dust"][q], equal_nan=True), ("dust", q)
    option("no_ml_floor")
    c = _planet_amr_run(cycles=2)
    d = _planet_amr_run(path="unfused", cycles=2)
    assert any(not np.array_equal(x, y, equal_nan=True) for x, y in zip(c["gas"], d["gas"]))


@pytest.mark.gpu
def test_cartesian_tile_march_equals_the_cell_centred_stage_on_the_shipped_disk_deck(hiplib, option):
    """inputs/disk/disk_cart.in (statically refined Cartesian disk, gravity + alpha viscosity, nghost 4) runs the SYS =
    cartesian instantiation of the tile march since round 6; NO_CART_MARCH sends it back to the cell-centred stage: same
    bits in every zone of every leaf, same dt."""
    def run():
        s = _disk_cart(["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=64", "parthenon/mesh/nx3=64"])
        s.evolve(6)
        kern = s.stage_kernel
        out = (s.dt, s.nblocks, kern, [s.field("gas.prim", b)[[0, 1, 2, 3, 5]] for b in range(s.nblocks)])
        s.close()
        return out
    a = run()
    assert "stage_curv_kernel" in a[2]
    option("no_cart_march")
    b = run()
    assert "stage_curv_kernel" not in b[2]
    assert a[0] == b[0] and a[1] == b[1]
    for q in range(a[1]):
        assert np.array_equal(a[3][q], b[3][q], equal_nan=True), q
