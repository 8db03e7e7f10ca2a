"""Adaptive mesh refinement: the host logic (tagging, tree update, hand-over of the conserved state, the rebuilt
block-graph exchange) on the CPU test double, 1 rank.  The GPU tests are tests/test_adaptive.py."""
import os

import numpy as np

from test_multilevel import _run_workers

LINWAVE = dict(deck=["linwave", "linear_wave_amr.in"], cycles=45,
               overrides=["problem/nperiod=1", "parthenon/mesh/nx1=64", "parthenon/mesh/nx2=32", "parthenon/meshblock/nx1=8",
                          "parthenon/meshblock/nx2=8"])


def test_linear_wave_amr_on_cpu_double_conserves_across_remeshes(tmp_path):
    """linear_wave_amr.in at half resolution, 45 cycles: the refined band follows the crest (blocks are created and,
    after derefine_count cycles, merged), the leaves tile the root mesh, and mass / momentum / energy after the run
    equal those of the initial (adaptively refined) state to round-off; total energy to the truncation level of the
    reference's remesh."""
    a = _run_workers(1, LINWAVE, tmp_path, "amr")[0]
    i = _run_workers(1, dict(LINWAVE, cycles=0), tmp_path, "ini")[0]
    lv = a["meta"]["levels"]
    assert a["meta"]["remeshes"] >= i["meta"]["remeshes"] + 2 and set(lv) == {0, 1}
    assert sum(4.0 ** (-l) for l in lv) == 32.0 == sum(4.0 ** (-l) for l in i["meta"]["levels"])
    assert a["meta"]["ncycle"] == 45 and i["meta"]["ncycle"] == 0
    scale = np.abs(i["hist"]).max()
    assert np.allclose(a["hist"][:4], i["hist"][:4], rtol=0, atol=2e-13 * scale), a["hist"][:4] - i["hist"][:4]
    # total energy: a remesh of the reference does not run SetAuxillaryFields (fill_derived.cpp:28) -- ConsToPrim takes
    # the internal energy as it was prolongated and PrimToCons rebuilds E from it, so E moves by O(dx^2) of the kinetic
    # energy of the wave per remesh (measured 2e-10 relative over this run), not by round-off
    assert abs(a["hist"][4] - i["hist"][4]) < 5e-9 * scale, a["hist"][4] - i["hist"][4]
    for _, prim in a["blocks"]:
        assert np.isfinite(prim).all()


def test_adaptive_mesh_on_two_ranks_equals_one_rank_bitwise(tmp_path):
    """Blocks migrate between ranks when the tree changes (the Z-ordered leaf list is re-dealt): whole blocks of
    conserved variables travel through the transport, the tags are all-reduced, and 2 ranks reproduce 1 rank bit for
    bit -- same tree, same cycle count, same dt, same zones."""
    from test_multirank_cpu import by_bounds
    one = _run_workers(1, LINWAVE, tmp_path, "r1")
    two = _run_workers(2, LINWAVE, tmp_path, "r2")
    assert sorted(one[0]["meta"]["levels"]) == sorted(two[0]["meta"]["levels"] + two[1]["meta"]["levels"])
    assert one[0]["meta"]["remeshes"] == two[0]["meta"]["remeshes"] == two[1]["meta"]["remeshes"] >= 3
    for r in two:
        assert r["meta"]["ncycle"] == one[0]["meta"]["ncycle"] == 45 and r["meta"]["dt"] == one[0]["meta"]["dt"]
    a, b = by_bounds(one), by_bounds(two)
    assert a.keys() == b.keys()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    assert np.allclose(two[0]["hist"], one[0]["hist"], rtol=1e-13)


def test_cylindrical_blast_amr_on_two_ranks_equals_one_rank_bitwise(tmp_path):
    """blast_amr.in at a quarter of its root resolution (cylindrical-polar: migrating blocks bring their metric tables
    along), three levels, 150 cycles with several remeshes: 2 ranks == 1 rank, bit for bit; mass conserved."""
    from test_multirank_cpu import by_bounds
    spec = dict(deck=["blast", "blast_amr.in"], cycles=150, overrides=["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=64"])
    one = _run_workers(1, spec, tmp_path, "b1")
    two = _run_workers(2, spec, tmp_path, "b2")
    ini = _run_workers(1, dict(spec, cycles=0), tmp_path, "b0")
    assert set(one[0]["meta"]["levels"]) == {0, 1, 2} and one[0]["meta"]["remeshes"] > ini[0]["meta"]["remeshes"]
    assert one[0]["meta"]["remeshes"] == two[0]["meta"]["remeshes"]
    a, b = by_bounds(one), by_bounds(two)
    assert a.keys() == b.keys()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    assert abs(one[0]["hist"][0] - ini[0]["hist"][0]) < 1e-13 * ini[0]["hist"][0]


def test_three_dimensional_adaptive_blast_two_ranks_equal_one_rank(tmp_path):
    """3-D Cartesian Sedov blast (inputs/blast/blast.in) on an adaptive mesh, 16^3 root in 8^3 blocks, two levels on the
    pressure gradient: octant bookkeeping of the hand-over in three dimensions.  2 ranks == 1 rank bit for bit, mass and
    total energy conserved (periodic box: the refined region also crosses the periodic seam)."""
    from test_multirank_cpu import by_bounds
    spec = dict(deck=["blast", "blast.in"], cycles=25,
                overrides=["parthenon/mesh/nx1=16", "parthenon/mesh/nx2=16", "parthenon/mesh/nx3=16", "parthenon/mesh/x3min=-1.0",
                           "parthenon/mesh/x3max=1.0", "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8",
                           "parthenon/meshblock/nx3=8", "parthenon/mesh/refinement=adaptive", "parthenon/mesh/numlevel=2",
                           "parthenon/mesh/derefine_count=3", "gas/refine_field=pressure", "gas/refine_type=gradient",
                           "gas/refine_thr=3.0", "problem/symmetry=spherical", "problem/radius=0.35", "problem/samples=0",
                           "problem/p0=0.05", "problem/x1=-0.4", "problem/x2=-0.4", "problem/x3=-0.4"]
                + ["parthenon/mesh/%sx%d_bc=periodic" % (io, d) for io in "io" for d in (1, 2, 3)])
    one = _run_workers(1, spec, tmp_path, "t1")
    two = _run_workers(2, spec, tmp_path, "t2")
    ini = _run_workers(1, dict(spec, cycles=0), tmp_path, "t0")
    lv = one[0]["meta"]["levels"]
    assert set(lv) == {0, 1} and sum(8.0 ** (-l) for l in lv) == 8.0
    assert one[0]["meta"]["remeshes"] > ini[0]["meta"]["remeshes"] and one[0]["meta"]["remeshes"] == two[0]["meta"]["remeshes"]
    a, b = by_bounds(one), by_bounds(two)
    assert a.keys() == b.keys()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    h0, h1 = ini[0]["hist"], one[0]["hist"]
    assert abs(h1[0] - h0[0]) < 1e-13 * h0[0] and abs(h1[4] - h0[4]) < 1e-3 * h0[4]  # (energy: truncation level, above)


def test_config4_disk_planet_dust_two_ranks_equal_one_rank(tmp_path):
    """BASELINE configs[4]'s deck (tests/amr_cases.py: cylindrical disk + planet + dust with drag + viscosity, four
    levels) split over 2 ranks: blocks of BOTH fluids migrate when the tree changes, the N-body force rows are summed
    over ranks, and the run reproduces 1 rank bit for bit.  (1 rank == the independent adaptive oracle:
    tests/test_adaptive_oracle.py.)"""
    import amr_cases
    from test_multirank_cpu import by_bounds
    case = amr_cases.disk_planet_dust_amr()
    spec = dict(deck=list(case["deck"]), cycles=20, dust=True, overrides=case["overrides"])
    one = _run_workers(1, spec, tmp_path, "c1")
    two = _run_workers(2, spec, tmp_path, "c2")
    assert one[0]["meta"]["remeshes"] == two[0]["meta"]["remeshes"] == two[1]["meta"]["remeshes"] >= 5
    assert max(one[0]["meta"]["levels"]) == 3
    assert sorted(one[0]["meta"]["levels"]) == sorted(two[0]["meta"]["levels"] + two[1]["meta"]["levels"])
    for r in two:
        assert r["meta"]["ncycle"] == one[0]["meta"]["ncycle"] == 20 and r["meta"]["dt"] == one[0]["meta"]["dt"]
    a, b = by_bounds(one), by_bounds(two)
    assert a.keys() == b.keys()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    da, db = by_bounds(one, "dust"), by_bounds(two, "dust")
    for key in da:
        assert np.array_equal(da[key], db[key]), key


def test_config4_in_three_dimensions_two_ranks_equal_one_rank(tmp_path):
    """The same deck in THREE dimensions (tests/amr_cases.py with nz = 8: 16 x 16 x 8 root in 8^3 blocks, four levels,
    312 -> 368 blocks of gas and dust over the first remesh of the run): octants of both fluids migrate between the
    ranks, x3 restriction / prolongation / flux correction run across the rank boundary, and 2 ranks reproduce 1 rank
    bit for bit.  (The GPU driver == the adaptive oracle on this case over 18 cycles: tests/test_adaptive.py.)"""
    import amr_cases
    from test_multirank_cpu import by_bounds
    case = amr_cases.disk_planet_dust_amr(n=16, planet=3e-2, thr=2.5, nz=8, zlim=0.01)
    spec = dict(deck=list(case["deck"]), cycles=5, dust=True, overrides=case["overrides"])
    one = _run_workers(1, spec, tmp_path, "d1")
    two = _run_workers(2, spec, tmp_path, "d2")
    assert one[0]["meta"]["remeshes"] == two[0]["meta"]["remeshes"] == two[1]["meta"]["remeshes"] >= 4
    assert max(one[0]["meta"]["levels"]) == 3 and len(one[0]["meta"]["levels"]) > 312
    assert sorted(one[0]["meta"]["levels"]) == sorted(two[0]["meta"]["levels"] + two[1]["meta"]["levels"])
    for r in two:
        assert r["meta"]["ncycle"] == one[0]["meta"]["ncycle"] == 5 and r["meta"]["dt"] == one[0]["meta"]["dt"]
    a, b = by_bounds(one), by_bounds(two)
    assert a.keys() == b.keys()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    da, db = by_bounds(one, "dust"), by_bounds(two, "dust")
    for key in da:
        assert np.array_equal(da[key], db[key]), key


def test_cost_weighted_split_on_an_adaptive_mesh_equals_one_rank(tmp_path):
    """<artemis_amd/loadbalance> level_cost / flux_face_cost deal the Z-ordered leaves to the ranks in runs of UNEQUAL
    block counts.  A run-time remesh asks both the old and the new state who owns a leaf and which local slot it has
    (adopt_state_from): with the equal-count formula there, blocks would be copied from wrong slots and the migration
    messages of two ranks would not match (round-4 advisor finding).  Each state now carries its own split: 2 and 3
    ranks with weighted costs reproduce 1 rank bit for bit across the remeshes of linear_wave_amr, with block counts
    that differ between the ranks."""
    from test_multirank_cpu import by_bounds
    lb = ["artemis_amd/loadbalance/level_cost=1.0,3.0", "artemis_amd/loadbalance/flux_face_cost=0.25"]
    spec = dict(LINWAVE, overrides=LINWAVE["overrides"] + lb)
    one = _run_workers(1, LINWAVE, tmp_path, "w1")
    a = by_bounds(one)
    uneven = []
    for nr in (2, 3):
        many = _run_workers(nr, spec, tmp_path, "w%d" % nr)
        counts = [r["meta"]["nblocks"] for r in many]
        uneven.append(len(set(counts)) > 1)
        assert sum(counts) == one[0]["meta"]["nblocks"]
        for r in many:
            assert r["meta"]["remeshes"] == one[0]["meta"]["remeshes"] >= 3
            assert r["meta"]["ncycle"] == 45 and r["meta"]["dt"] == one[0]["meta"]["dt"]
        b = by_bounds(many)
        assert a.keys() == b.keys()
        for key in a:
            assert np.array_equal(a[key], b[key]), (nr, key)
    assert any(uneven)  # (the weighted split really dealt unequal counts in at least one of the worlds)
