"""Adaptive mesh refinement against an INDEPENDENT oracle (oracle/adaptive.py: Python sets + per-block C oracles, no
code shared with the product's block tree): the product's host logic -- tagging, derefinement counters, the finer-
neighbour rule, 2:1 balance, Z-ordering, the hand-over of the conserved state by copy / ProlongateSharedMinMod /
RestrictAverage, ConsToPrim -> exchange -> PrimToCons, the dt rule -- on the CPU test double, compared after EVERY
batch of cycles: tree shape, dt, time, and every leaf bit for bit (ghost zones included).  The GPU versions of the
same cases (the HIP driver) are in tests/test_adaptive.py.

Reference anchors: gas.cpp:305-380 (criterion selection), utils/refinement/amr_criteria.hpp:29-168,
prolongation.hpp:83-184, restriction.hpp:42-114, artemis_driver.cpp:291-293 (tagging task), fill_derived.cpp:28
(SetAuxillaryFields is not part of a remesh).  Parity with Parthenon's remesher itself stays unpinned (absent
submodule; the reference holds no regression answers for its AMR decks)."""
import ctypes as C
import os
import subprocess

import pytest

import amr_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def double():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpu_double"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    return C.CDLL(os.path.join(ROOT, "tests", "_build", "libartemis_cpudouble.so"))


def run_case(lib, case, cycles, batch, min_remeshes, levels):
    from artemis_amd.driver import Simulation
    s = Simulation(amr_cases.DECK(*case["deck"]), case["overrides"], lib=lib)
    m = case["oracle"]()
    assert s.remeshes == m.remeshes  # Mesh::Initialize's refinement loop took the same number of passes
    r0, done, seen = s.remeshes, 0, set()
    while done < cycles:
        done += s.evolve(min(batch, cycles - done))
        m.evolve(case["tlim"], done)
        amr_cases.compare(s, m, case["dust"])
        assert s.remeshes == m.remeshes
        seen |= set(m.level_counts())
    assert s.remeshes - r0 >= min_remeshes, (s.remeshes, r0)
    assert seen >= set(levels), seen
    s.close()
    return m


def test_blast_amr_deck_host_logic_equals_adaptive_oracle(double):
    """inputs/blast/blast_amr.in at half its root resolution (64^2 in 8^2 blocks, three levels, cylindrical): 110 cycles,
    >= 8 remeshes in which the finest level follows the shock outwards and the blocks behind it merge again."""
    run_case(double, amr_cases.blast_amr(n=64, derefine_count=5), 110, 10, 8, {0, 1, 2})


def test_linear_wave_amr_deck_host_logic_equals_adaptive_oracle(double):
    """inputs/linwave/linear_wave_amr.in as shipped (128 x 64 in 16^2 blocks, two levels, periodic): the refined band
    rides on the crests, blocks are created ahead of it and merged behind it."""
    m = run_case(double, amr_cases.linear_wave_amr(derefine_count=3), 60, 10, 4, {0, 1})
    assert m.remeshes >= 5


def test_config4_disk_planet_dust_four_levels_host_logic_equals_adaptive_oracle(double):
    """BASELINE configs[4] (amr_cases.disk_planet_dust_amr): cylindrical disk + planet (N-body gravity task) + one
    dust species with drag + alpha viscosity + rotating frame + `ic` conditions, numlevel = 4 on the pressure-gradient
    criterion.  30 cycles, >= 8 remeshes, levels 1..3 present (the root level refines completely at start-up)."""
    m = run_case(double, amr_cases.disk_planet_dust_amr(), 30, 5, 8, {1, 2, 3})
    assert max(l for l, _ in m.leaves) == 3  # four levels: 0 (root) .. 3


def test_config4_with_real_vertical_extent_host_logic_equals_adaptive_oracle(double):
    """configs[4] in three dimensions over |z| <= 0.2 (amr_cases.THICK_DISK: one scale height at the planet, 2.4 at the
    inner edge; 16 x 32 x 8 root in 8^3 blocks, four levels).  The vertical pressure gradient holds the layers at
    |z| > 0.1 of the inner disk at level 3 from the start while the midplane there stays at level 2, and the planet --
    0.08 above the midplane -- refines its surroundings over three remeshes (400 -> 456 -> 470 -> 484 blocks of gas and
    dust in 13 cycles).  (The thin-slab 3-D row is tests/test_adaptive.py's, on the GPU; this one also runs there.)"""
    m = run_case(double, amr_cases.disk_planet_dust_amr(**amr_cases.THICK_DISK), 13, 5, 3, {1, 2, 3})
    high = [b for b, (lv, _) in enumerate(m.leaves) if lv == 3 and min(abs(m.block_bounds(b)[4]), abs(m.block_bounds(b)[5])) >= 0.1]
    assert len(high) >= 200 and len(m.blocks) == 484
