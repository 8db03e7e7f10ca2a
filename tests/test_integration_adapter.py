"""INTEGRATION.md's adapter is a real file (integration/artemis_hip_adapter.hpp).  Two checks:

* it compiles, -Wall -Werror, against the C ABI of include/artemis_hip.h and a stand-in for the handful of
  Parthenon / Artemis names it touches (tests/mock_parthenon/: not Parthenon -- host arrays behind a SparsePack,
  a Params map, par_for as loops);
* it RUNS: tests/adapter_live/run_stage.cpp executes the reference's task list for RK2 steps
  (artemis_driver.cpp:145-273, in its order; two MeshData partitions, a u0 and a u1 register each) through the
  adapter's forwarders on the CPU test double of the library, and the result equals the oracle's bit for bit.
  This is the test that catches a forwarder handing the wrong register to a task (the round-2 adapter cached the
  cons1 tables from u0: gam0*u0 + gam1*u0 in stage 2) -- see test_a_stale_u1_register_is_caught."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = ["-I", os.path.join(ROOT, "tests", "mock_parthenon"), "-I", os.path.join(ROOT, "include"),
       "-I", os.path.join(ROOT, "integration")]


def test_adapter_compiles_against_the_c_abi(tmp_path):
    tu = tmp_path / "tu.cpp"
    tu.write_text('#include "artemis_hip_adapter.hpp"\nint parthenon::Globals::nghost = 2;\nint main() { return 0; }\n')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror"] + INC + [str(tu)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def test_adapter_fills_every_field_of_the_pack():
    """Every member of artemis_pack_t / artemis_fluid_pack_t must be assigned somewhere in the adapter."""
    hdr = open(os.path.join(ROOT, "include", "artemis_hip.h")).read()
    txt = open(os.path.join(ROOT, "integration", "artemis_hip_adapter.hpp")).read()
    fluid = re.search(r"typedef struct artemis_fluid_pack \{(.*?)\} artemis_fluid_pack_t;", hdr, re.S).group(1)
    pack = re.search(r"typedef struct artemis_pack \{(.*?)\} artemis_pack_t;", hdr, re.S).group(1)
    strip = lambda s: re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    names = lambda s: re.findall(r"[\*\s](\w+)(?:\[\d\])?\s*[;,]", strip(s))
    for n in names(fluid):
        if n in ("siefloor", "de_switch", "pflux", "vface", "diff_flux"):  # gas only
            assert re.search(r"p\.gas\.%s\b" % n, txt), n
        else:
            assert re.search(r"p\.gas\.%s\b" % n, txt) and re.search(r"p\.dust\.%s\b" % n, txt), n
    for n in names(pack):
        if n not in ("gas", "dust"):
            assert re.search(r"p\.%s\b" % n, txt), n


def test_adapter_forwards_every_task_of_the_stage():
    """One forwarder per task of ArtemisDriver::StepTasks that is on the path (artemis_driver.cpp:157-255) plus the
    package callbacks (artemis.cpp:122-123, gas.cpp:288-299, dust.cpp:216-227)."""
    txt = open(os.path.join(ROOT, "integration", "artemis_hip_adapter.hpp")).read()
    for fn in ("DeepCopyConservedData", "GasCalculateFluxes", "DustCalculateFluxes", "GasZeroDiffusionFlux", "GasViscousFlux",
               "GasThermalFlux", "ApplyUpdate", "GasFluxSource", "DustFluxSource", "GasDiffusionUpdate", "ExternalGravity",
               "RotatingFrameForce", "DragSource", "SetAuxillaryFields", "ConsToPrim", "PrimToCons",
               "GasEstimateTimestepMesh", "DustEstimateTimestepMesh", "StageFused", "StageFusedFillDerived"):
        assert re.search(r"inline \w+ %s\(" % fn, txt), fn


# ---- the live run ----------------------------------------------------------------------------------------------
NX, NG = (32, 8, 8), 2
LO, HI = (-1.0, -0.25, -0.25), (1.0, 0.25, 0.25)
SHAPE = (NX[2] + 2 * NG, NX[1] + 2 * NG, NX[0] + 2 * NG)


@pytest.fixture(scope="module")
def harness():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpu_double"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    out = os.path.join(ROOT, "tests", "_build", "adapter_run_stage")
    build = os.path.join(ROOT, "tests", "_build")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror"] + INC +
                          [os.path.join(ROOT, "tests", "adapter_live", "run_stage.cpp"), "-o", out, "-L", build,
                           "-lartemis_cpudouble", "-Wl,-rpath," + build, "-fopenmp"])
    return out


def initial_state(seed, dust):
    """Smooth random primitives over the ENTIRE block (ghost zones included: PostInitialization's PrimToCons covers
    them), a pressure jump in the middle so that limiters and the HLLC branches are exercised."""
    rng = np.random.default_rng(seed)
    k, j, i = np.meshgrid(*[np.arange(n, dtype=float) for n in SHAPE], indexing="ij")
    wave = lambda: 0.2 * np.sin(0.37 * i + rng.uniform(0, 6)) * np.cos(0.61 * j + rng.uniform(0, 6)) * np.cos(0.83 * k + rng.uniform(0, 6))
    g = np.empty((6,) + SHAPE)
    g[0] = 1.0 + wave() + 0.5 * (i > SHAPE[2] / 2)
    for c in range(3):
        g[1 + c] = 0.3 * wave() + 0.05 * rng.standard_normal(SHAPE)
    g[5] = 1.5 + wave() + 1.0 * (i > SHAPE[2] / 2)
    g[4] = 0.4 * g[0] * g[5]
    d = None
    if dust:
        d = np.empty((4,) + SHAPE)
        d[0] = 0.1 + 0.02 * wave()
        for c in range(3):
            d[1 + c] = 0.3 * wave()
    return g, d


def oracle_run(mode, g, d, dt, nsteps):
    from oracle.oracle import Oracle
    full = mode in ("full", "stage_full")
    o = Oracle(NX, LO, HI, ng=NG, ns_gas=1, ns_dust=1 if full else 0, reconstruct="plm", riemann="hllc",
               dust_reconstruct="plm", dust_riemann="hlle", gamma=1.4, dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10,
               cfl=0.3, dust_cfl=0.3, bc=("outflow",) * 6, integrator="rk2")
    if full:
        o.set_gravity_uniform(0.1, -0.2, 0.05)
        o.set_rotating_frame(1.0, 1.5)
        o.set_drag("simple_dust", "constant", tau=[0.1])
        o.set_viscosity("constant", nu=0.01)
        o.dprim[:] = d
    o.gprim[:] = g
    o.PrimToCons()
    for _ in range(nsteps):
        o.dt = dt
        o.step()
    return o


def run_harness(harness, tmp_path, mode, states, dt, nsteps, realloc):
    src, dst = tmp_path / ("%s.in" % mode), tmp_path / ("%s.out" % mode)
    with open(src, "wb") as f:
        for g, d in states:
            f.write(np.ascontiguousarray(g).tobytes())
            if d is not None:
                f.write(np.ascontiguousarray(d).tobytes())
    r = subprocess.run([harness, mode, str(src), str(dst), repr(dt), str(nsteps), str(int(realloc))],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    raw = np.fromfile(dst)
    n = int(np.prod(SHAPE))
    per = 12 + (4 if states[0][1] is not None else 0)
    assert raw.size == 2 * per * n + 1
    out = []
    for q in range(2):
        a = raw[q * per * n:(q + 1) * per * n].reshape((per,) + SHAPE)
        out.append((a[:6], a[6:12], a[12:] if per > 12 else None))
    return out, raw[-1]


@pytest.mark.parametrize("mode,realloc", [("gas", 0), ("gas", 1), ("full", 0), ("full", 1), ("fused", 1), ("gas", 2),
                                          ("full", 2), ("fused", 2), ("stage_gas", 1), ("stage_full", 0),
                                          ("stage_full", 1), ("stage_full", 2)])
def test_rk2_steps_through_the_adapter_equal_the_oracle(harness, tmp_path, mode, realloc):
    """Two RK2 steps of the reference's task list through the forwarders == the oracle, every bit of the primitives AND
    of the conserved state u0 (ghost zones included), on both partitions; with `realloc` every variable moves to a new
    allocation between the steps (a remesh / restart changes addresses like that) and the adapter must rebuild its
    tables by itself.  `full`: gas + dust + gravity + shearing box + drag + viscosity through the widened forwarders;
    `fused`: the opt-in StageFused / StageFusedFillDerived pair (cons current after every stage); `stage_gas` /
    `stage_full`: the DEFAULT wiring of INTEGRATION.md section 3 -- StageCovered -> Stage -> conditions ->
    StageFillDerived, one task per stage (the tuned kernel for gas alone; artemis_hip_stage_general with the
    diffusion-flux tasks inside for gas + dust + gravity + shearing box + drag + viscosity).  realloc = 2: both
    blocks in ONE partition and, between the steps, the SECOND block replaced by a new block object with another gid /
    logical location and fresh allocations while the first stays put -- what a remesh does to a refined or migrated
    block next to an unchanged one; a cache keyed on the partition's first block alone would keep the stale tables."""
    dust = mode in ("full", "stage_full")
    states = [initial_state(11, dust), initial_state(29, dust)]
    dt, nsteps = 2.0e-3, 2
    got, dt_est = run_harness(harness, tmp_path, mode, states, dt, nsteps, realloc)
    dts = []
    for q, (g, d) in enumerate(states):
        o = oracle_run(mode, g, d, dt, nsteps)
        dts.append(o.new_dt())
        keep = [0, 1, 2, 3, 5] if mode in ("fused", "stage_gas") else list(range(6))  # (the fused stage keeps P on interior zones only)
        assert np.array_equal(got[q][0][keep], o.gprim[keep]), (mode, q, "gas prim")
        assert np.array_equal(got[q][1], o.gu0), (mode, q, "gas cons")
        if dust:
            assert np.array_equal(got[q][2], o.dprim), (mode, q, "dust prim")
    # EstimateTimestepMesh of the new state (gas, dust, viscous limit) over the blocks of partition 0
    assert dt_est == (min(dts) if realloc == 2 else dts[0])


def test_default_wiring_leaves_refined_meshes_to_the_task_list(harness, tmp_path):
    """pmesh->multilevel: the one-kernel stage stores no flux arrays, so the flux correction the replaced task block
    contains (artemis_driver.cpp:196-202) would silently have nothing to send.  StageCovered must be false there and
    Stage itself must fail with PARTHENON_REQUIRE's exception (round-4 advisor finding)."""
    states = [initial_state(11, False), initial_state(29, False)]
    src = tmp_path / "ml.in"
    with open(src, "wb") as f:
        for g, _ in states:
            f.write(np.ascontiguousarray(g).tobytes())
    r = subprocess.run([harness, "covered_multilevel", str(src), str(tmp_path / "ml.out"), "1e-3", "1", "0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])


def test_a_stale_u1_register_is_caught(harness, tmp_path):
    """Sanity of the check itself: the oracle run with the round-2 defect emulated (stage 2 combining u0 with ITSELF
    instead of with the start-of-step copy) differs from the adapter's result at O(dt) -- i.e. the comparison above
    would have failed on the old adapter."""
    states = [initial_state(11, False), initial_state(29, False)]
    dt = 2.0e-3
    got, _ = run_harness(harness, tmp_path, "gas", states, dt, 1, 0)
    from oracle.oracle import Oracle
    o = Oracle(NX, LO, HI, ng=NG, reconstruct="plm", riemann="hllc", gamma=1.4, dfloor=1e-10, siefloor=1e-10, cfl=0.3,
               bc=("outflow",) * 6, integrator="rk2")
    o.gprim[:] = states[0][0]
    o.PrimToCons()
    for stage, (g0, g1, be) in enumerate(((0.0, 1.0, 1.0), (0.5, 0.5, 0.5))):
        o.DeepCopyConservedData()  # u1 <- u0 before EVERY stage: what a cons1 table aliased to u0 amounts to
        o.CalculateFluxes(0, False)
        o.ApplyUpdate(g0, g1, be * dt)
        o.FluxSource(be * dt, 0)
        o.SetAuxillaryFields(), o.ConsToPrim(), o.ApplyBoundaryConditions(), o.PrimToCons()
    diff = np.abs(got[0][0][0] - o.gprim[0]).max()
    assert diff > 1e-6, diff
