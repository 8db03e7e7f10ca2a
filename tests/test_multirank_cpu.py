"""Host-logic tests without a GPU: the product's C++ driver (unmodified sources) linked against
the CPU test double of the C ABI (tests/cpu_double, oracle-backed), single process and under
torch.distributed/gloo with world_size 2 and 4.  What is under test is everything ABOVE the
kernels: deck parsing, the rank grid / mesh-block bricks, ghost-slab links and tags (including
periodic wrap-around between ranks), TorchComm's batched isend/irecv, the dt all-reduce and the
EvolutionDriver loop.  The N-rank result must equal the 1-rank run of the same block layout
BIT FOR BIT (same per-block arithmetic, only the transport differs)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def double_lib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpu_double"), "-s"])
    return os.path.join(ROOT, "tests", "_build", "libartemis_cpudouble.so")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_RESULTS = {}  # (world, spec) -> results: several tests compare against the same reference run


def run_world(world, spec, tmp_path, tag):
    key = (world, json.dumps(spec, sort_keys=True))
    if key in _RESULTS:
        return _RESULTS[key]
    spec = dict(spec, out=str(tmp_path / tag))
    # the checker's OpenMP team per rank: the machine's CPUs shared by the ranks (the updates have no cross-thread
    # floating-point reductions, so the fields do not depend on the team size)
    threads = str(max(1, min(4, (os.cpu_count() or 1) // world)))
    for attempt in range(2):  # (the probed port can be taken between the probe and the rendezvous: one retry)
        port = str(free_port())
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=port, OMP_NUM_THREADS=threads)
            env.update(spec.get("env", {}))  # (option switches of the run: ARTEMIS_<NAME>)
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mr_worker.py"),
                                           json.dumps(spec)], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=600)[0].decode() for p in procs]
        # (also retried once: gloo's worker threads racing the interpreter's teardown on a loaded machine --
        # "terminate called without an active exception", after the rank has written its results)
        rendezvous = any(p.returncode != 0 and ("Address already in use" in o or "Connection re" in o or "timed out" in o.lower()
                                                or "terminate called without an active exception" in o)
                         for p, o in zip(procs, outs))
        if not (rendezvous and attempt == 0):
            break
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    res = []
    for r in range(world):
        z = np.load(spec["out"] + ".rank%d.npz" % r)
        nblk = json.loads(str(z["meta"]))["nblocks"]
        res.append(dict(meta=json.loads(str(z["meta"])), hist=z["hist"], errs=z["errs"],
                        blocks=[(z["bounds%d" % b], z["prim%d" % b]) for b in range(nblk)],
                        dust=[z["dust%d" % b] for b in range(nblk)] if spec.get("dust") else []))
    _RESULTS[key] = res
    return res


def by_bounds(results, which="gas"):
    d = {}
    for r in results:
        for q, (bounds, prim) in enumerate(r["blocks"]):
            d[tuple(np.round(bounds, 12))] = prim if which == "gas" else r["dust"][q]
    return d


BLAST = dict(deck=["blast", "blast.in"], cycles=6, overrides=[
    "parthenon/mesh/nx1=32", "parthenon/mesh/nx2=16", "parthenon/mesh/nx3=16",
    "parthenon/mesh/x3min=-1.0", "parthenon/mesh/x3max=1.0", "parthenon/meshblock/nx1=16",
    "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8", "gas/riemann=hllc",
    "problem/symmetry=spherical", "problem/radius=0.3", "problem/samples=0"])

LINWAVE = dict(deck=["linwave", "linear_wave.in"], overrides=[
    "parthenon/mesh/nghost=2", "parthenon/mesh/nx1=16", "parthenon/mesh/nx2=8", "parthenon/mesh/nx3=8",
    "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=4", "parthenon/meshblock/nx3=4",
    "problem/amp=1.0e-6", "problem/wave_flag=0", "problem/vflow=0.0", "parthenon/time/nlim=1000"])


@pytest.mark.parametrize("world", [2, 4])
def test_blast_ranks_equal_single_process_bitwise(double_lib, tmp_path, world):
    one = run_world(1, BLAST, tmp_path, "one")
    many = run_world(world, BLAST, tmp_path, "w%d" % world)
    assert one[0]["meta"]["nblocks"] == 8 and sum(r["meta"]["nblocks"] for r in many) == 8
    assert one[0]["meta"]["fused"]
    for r in many:
        for k in ("ncycle", "time", "dt"):
            assert r["meta"][k] == one[0]["meta"][k], k
        assert np.allclose(r["hist"], one[0]["hist"], rtol=1e-13)
    a, b = by_bounds(one), by_bounds(many)
    assert a.keys() == b.keys()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    if world == 2:  # no time limit: dt stays in "device" memory, reduced in place by the communicator
        spec = dict(BLAST, overrides=BLAST["overrides"] + ["parthenon/time/tlim=-1.0"])
        fa, fb = run_world(1, spec, tmp_path, "nolim1"), run_world(2, spec, tmp_path, "nolim2")
        assert fb[0]["meta"]["dt"] == fb[1]["meta"]["dt"] == fa[0]["meta"]["dt"]
        assert fb[0]["meta"]["time"] == fa[0]["meta"]["time"]
        xa, xb = by_bounds(fa), by_bounds(fb)
        for key in xa:
            assert np.array_equal(xa[key], xb[key]), key
    if world == 2:  # shell-first ordering with the exchange on the comm stream
        ovl = by_bounds(run_world(world, dict(BLAST, overlap=True), tmp_path, "ovl"))
        for key in a:
            assert np.array_equal(a[key], ovl[key]), key


def test_linwave_periodic_wraparound_two_ranks(double_lib, tmp_path):
    """Periodic images cross the rank boundary (2x2x2 blocks, 2 ranks): bitwise equal to the
    single-process run, full period, reference error threshold shape (not value: N = 16)."""
    one = run_world(1, LINWAVE, tmp_path, "one")
    two = run_world(2, LINWAVE, tmp_path, "two")
    assert one[0]["meta"]["ncycle"] == two[0]["meta"]["ncycle"] == two[1]["meta"]["ncycle"] > 10
    a, b = by_bounds(one), by_bounds(two)
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    # error norms are all-reduced sums: identical on both ranks, equal to 1 rank to round-off
    assert np.array_equal(two[0]["errs"], two[1]["errs"])
    assert np.allclose(two[0]["errs"], one[0]["errs"], rtol=1e-10)
    assert 1e-8 < one[0]["errs"][0] < 2e-6


def test_unfused_path_and_block_layouts_single_process(double_lib, tmp_path):
    """Per-task path == fused path through the double, and the 8-block layout agrees with the
    1-block layout to round-off (block-local cell edges differ in the last bit)."""
    spec1 = dict(BLAST, overrides=[o for o in BLAST["overrides"] if "meshblock" not in o] +
                 ["parthenon/meshblock/nx1=32", "parthenon/meshblock/nx2=16", "parthenon/meshblock/nx3=16"])
    f8 = run_world(1, BLAST, tmp_path, "f8")
    u8 = run_world(1, dict(BLAST, path="unfused"), tmp_path, "u8")
    f1 = run_world(1, spec1, tmp_path, "f1")
    assert not u8[0]["meta"]["fused"]
    a, b = by_bounds(f8), by_bounds(u8)
    for key in a:
        assert np.array_equal(a[key], b[key])
    full = f1[0]["blocks"][0][1]
    for bounds, prim in f8[0]["blocks"]:
        i0 = int(round((bounds[0] + 1.0) / (2.0 / 32)))
        j0 = int(round((bounds[2] + 1.0) / (2.0 / 16)))
        k0 = int(round((bounds[4] + 1.0) / (2.0 / 16)))
        ref = full[:, k0:k0 + 8, j0:j0 + 8, i0:i0 + 16]
        assert np.max(np.abs(prim - ref) / (np.abs(ref) + 1e-30)) < 1e-11


SPH3D = dict(deck=["blast", "blast.in"], cycles=5, overrides=[
    "artemis/coordinates=spherical", "parthenon/mesh/nx1=16", "parthenon/mesh/nx2=8", "parthenon/mesh/nx3=8",
    "parthenon/mesh/x1min=0.2", "parthenon/mesh/x1max=1.4", "parthenon/mesh/x2min=0.7",
    "parthenon/mesh/x2max=2.4", "parthenon/mesh/x3min=0.0", "parthenon/mesh/x3max=6.283185307179586",
    "parthenon/mesh/ix1_bc=reflecting", "parthenon/mesh/ix2_bc=reflecting", "parthenon/mesh/ox2_bc=reflecting",
    "parthenon/mesh/ix3_bc=periodic", "parthenon/mesh/ox3_bc=periodic",
    "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=4", "parthenon/meshblock/nx3=4",
    "problem/symmetry=spherical", "problem/radius=0.6", "problem/samples=0", "problem/p0=1.0e-2"])


def test_spherical3d_two_ranks_equal_single_process_bitwise(double_lib, tmp_path):
    """artemis/coordinates = spherical on a 3-D wedge, 2x2x2 blocks with reflecting radial /
    polar and periodic azimuthal boundaries: the per-task chain with the metric tables of each rank's own
    blocks gives the same bits on 2 ranks as on 1 and as the fused general stage (the default for one gas
    species on a curvilinear mesh); mass is conserved with the spherical cell volumes."""
    one = run_world(1, dict(SPH3D, path="unfused"), tmp_path, "s1")
    two = run_world(2, dict(SPH3D, path="unfused"), tmp_path, "s2")
    assert not one[0]["meta"]["fused"] and one[0]["meta"]["ncycle"] == 5
    fs = run_world(1, SPH3D, tmp_path, "s1f")
    assert fs[0]["meta"]["fused"] and not fs[0]["meta"]["tuned"]
    per_task = by_bounds(fs)
    fs2 = by_bounds(run_world(2, SPH3D, tmp_path, "s2f"))  # the default (fused) path split over two ranks
    assert fs2.keys() == per_task.keys()
    for key in fs2:
        assert np.array_equal(fs2[key], per_task[key]), key
    for r in two:
        for k in ("ncycle", "time", "dt"):
            assert r["meta"][k] == one[0]["meta"][k], k
    a, b = by_bounds(one), by_bounds(two)
    assert a.keys() == b.keys() and len(a) == 8
    for key in a:
        assert np.array_equal(a[key], b[key]), key
        assert np.array_equal(a[key], per_task[key]), key  # general fused stage == per-task chain
        assert np.isfinite(a[key]).all()
    # volume of the wedge x d0 = 1: (r1^3 - r0^3)/3 * (cos t0 - cos t1) * 2 pi
    vol = (1.4 ** 3 - 0.2 ** 3) / 3.0 * (np.cos(0.7) - np.cos(2.4)) * 2 * np.pi
    assert abs(one[0]["hist"][0] - vol) < 1e-12 * vol
    assert np.allclose(two[0]["hist"], one[0]["hist"], rtol=1e-13)


# ---- source packages through the driver: drag (simple_drag.in) and the shearing sheet (ssheet.in) --
def test_drag_deck_driver_equals_oracle_and_ranks_agree(double_lib, tmp_path):
    """inputs/drag/simple_drag.in (constant pgen, gas + 4 dust species, simple_dust drag,
    periodic): the driver's task order and deck parsing against oracle.step() bit for bit on one
    block, and 4 blocks on 2 ranks against 1 rank."""
    from oracle.oracle import Oracle
    one_blk = dict(deck=["drag", "simple_drag.in"], cycles=25, dust=True,
                   overrides=["parthenon/meshblock/nx1=128"])
    r = run_world(1, one_blk, tmp_path, "d1")[0]
    assert r["meta"]["fused"] and not r["meta"]["tuned"] and r["meta"]["nblocks"] == 1
    u = run_world(1, dict(one_blk, path="unfused"), tmp_path, "d1u")[0]
    assert not u["meta"]["fused"] and np.array_equal(u["blocks"][0][1], r["blocks"][0][1])
    assert np.array_equal(u["dust"][0], r["dust"][0]) and u["meta"]["dt"] == r["meta"]["dt"]
    o = Oracle((128, 1, 1), (0.0, -0.5, -0.5), (1.0, 0.5, 0.5), ng=2, ns_gas=1, ns_dust=4,
               reconstruct="plm", riemann="hlle", dust_reconstruct="plm", dust_riemann="hlle", gamma=1.4,
               dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10, cfl=0.3, dust_cfl=0.3,
               bc=("periodic",) * 6, integrator="rk2")
    o.set_drag("simple_dust", "constant", tau=[1e-2, 0.1, 1.0, 1e1])
    o.pgen_constant(gas_rho=10.0, gas_v=(1.0, 0, 0), gas_temp=1.0, dust_rho=0.01, dust_v=(0, 0, 0))
    o.evolve(10.0, 25)
    assert r["meta"]["time"] == o.time and r["meta"]["dt"] == o.dt
    assert np.array_equal(r["blocks"][0][1], o.interior(o.gprim))
    assert np.array_equal(r["dust"][0], o.interior(o.dprim))
    assert np.max(np.abs(o.interior(o.dprim)[4] - 0.0)) > 0.5  # the tau = 0.01 grains have caught up
    four = dict(one_blk, overrides=[])
    a, b = run_world(1, four, tmp_path, "d4"), run_world(2, four, tmp_path, "d4r2")
    assert a[0]["meta"]["nblocks"] == 4 and [x["meta"]["nblocks"] for x in b] == [2, 2]
    xa, xb = by_bounds(a), by_bounds(b)
    for key in xa:
        assert np.array_equal(xa[key], xb[key]), key
    assert a[0]["meta"]["dt"] == b[0]["meta"]["dt"] == b[1]["meta"]["dt"]


def test_shearing_sheet_deck_driver_equals_oracle_and_ranks_agree(double_lib, tmp_path):
    """inputs/ssheet/ssheet.in (strat pgen, point-mass gravity, shearing-box sources, extrap /
    inflow user boundary conditions) at 32^2: driver == oracle on one block; 4 blocks on 2 ranks
    == 1 rank (the user conditions act on the outer blocks only, interior faces exchange)."""
    from oracle.oracle import Oracle
    small = ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "gravity/point/mass=1.0e-3"]
    one_blk = dict(deck=["ssheet", "ssheet.in"], cycles=20,
                   overrides=small + ["parthenon/meshblock/nx1=32", "parthenon/meshblock/nx2=32"])
    r = run_world(1, one_blk, tmp_path, "s1")[0]
    assert r["meta"]["fused"] and not r["meta"]["tuned"] and r["meta"]["nblocks"] == 1
    u = run_world(1, dict(one_blk, path="unfused"), tmp_path, "s1u")[0]
    assert not u["meta"]["fused"] and np.array_equal(u["blocks"][0][1], r["blocks"][0][1])
    o = Oracle((32, 32, 1), (-1.0, -1.0, -0.2), (1.0, 1.0, 0.2), ng=2, reconstruct="plm", riemann="hllc",
               gamma=1.000001, dfloor=1e-10, siefloor=1e-10, cfl=0.3,
               bc=("extrap", "extrap", "inflow", "inflow", "extrap", "extrap"), integrator="rk2")
    o.set_rotating_frame(1.0, 1.5)
    o.set_gravity_point(1e-3, soft=0.03)
    o.pgen_strat(rho0=1.0, dens_min=1e-10, h=0.05)
    o.evolve(100.0, 20)
    assert r["meta"]["time"] == o.time and r["meta"]["dt"] == o.dt
    assert np.array_equal(r["blocks"][0][1], o.interior(o.gprim))
    assert np.ptp(o.interior(o.gprim)[0]) > 1e-6  # the planet has perturbed the sheet
    four = dict(one_blk, overrides=small + ["parthenon/meshblock/nx1=16", "parthenon/meshblock/nx2=16"])
    a, b = run_world(1, four, tmp_path, "s4"), run_world(2, four, tmp_path, "s4r2")
    assert a[0]["meta"]["nblocks"] == 4 and [x["meta"]["nblocks"] for x in b] == [2, 2]
    xa, xb = by_bounds(a), by_bounds(b)
    for key in xa:
        assert np.array_equal(xa[key], xb[key]), key


# ---- gas diffusion through the driver (viscous_diffusion.py configuration) --------------------------
VISC = ["physics/viscosity=true", "physics/conduction=false", "gas/viscosity/nu=2.5e-01",
        "problem/temperature_bump=0.0", "problem/sigma=0.5", "parthenon/time/tlim=2.0"]


def test_viscous_diffusion_deck_driver_equals_oracle_and_block_edges(double_lib, tmp_path):
    """inputs/diffusion/gaussian_bump.in as tst/scripts/diffusion/viscous_diffusion.py runs it (2-D,
    nu = 0.25, a v3 bump plus -- to make the cross-derivative terms and hence the edge / corner
    ghost zones matter -- v1 and v2 bumps): one block == oracle bit for bit; 2x2 blocks on 1 and 2
    ranks agree bit for bit with each other and with the one-block run to round-off, which only
    holds if the sequential extended-slab exchange delivers the diagonal neighbours' zones."""
    from oracle.oracle import Oracle
    bump = ["problem/vx3_bump=1.0e-2", "problem/vx1_bump=2.0e-2", "problem/vx2_bump=-1.5e-2", "problem/x1c=0.4",
            "problem/x2c=-0.3"]
    one_blk = dict(deck=["diffusion", "gaussian_bump.in"], cycles=30,
                   overrides=VISC + bump + ["parthenon/meshblock/nx1=64", "parthenon/meshblock/nx2=64"])
    r = run_world(1, one_blk, tmp_path, "v1")[0]
    assert r["meta"]["fused"] and not r["meta"]["tuned"] and r["meta"]["nblocks"] == 1
    o = Oracle((64, 64, 1), (-6.0, -6.0, -0.5), (6.0, 6.0, 0.5), ng=2, reconstruct="plm", riemann="hllc",
               gamma=1.000001, dfloor=1e-10, siefloor=1e-10, cfl=0.3,
               bc=("outflow",) * 4 + ("periodic",) * 2, integrator="rk2")
    o.set_viscosity("constant", nu=0.25)
    o.pgen_gaussian_bump(sigma=0.5, centre=(0.4, -0.3, 0.0), v_bump=(2.0e-2, -1.5e-2, 1.0e-2))
    o.evolve(2.0, 30)
    assert r["meta"]["time"] == o.time and r["meta"]["dt"] == o.dt
    assert np.array_equal(r["blocks"][0][1], o.interior(o.gprim))
    four = dict(one_blk, overrides=VISC + bump)  # the deck's own 32^2 blocks
    a, b = run_world(1, four, tmp_path, "v4"), run_world(2, four, tmp_path, "v4r2")
    assert a[0]["meta"]["nblocks"] == 4 and [x["meta"]["nblocks"] for x in b] == [2, 2]
    xa, xb = by_bounds(a), by_bounds(b)
    full = o.interior(o.gprim)
    for key in xa:
        assert np.array_equal(xa[key], xb[key]), key
        i0, j0 = int(round((key[0] + 6.0) / 12.0 * 64)), int(round((key[2] + 6.0) / 12.0 * 64))
        ref = full[:, :, j0:j0 + 32, i0:i0 + 32]
        assert np.max(np.abs(xa[key] - ref)) < 1e-13, key


def test_conduction_deck_driver_equals_oracle(double_lib, tmp_path):
    """The shipped deck (heat conduction of a temperature bump, 2x2 blocks): driver == oracle on one
    block; the temperature peak decays and heat is conserved."""
    from oracle.oracle import Oracle
    spec = dict(deck=["diffusion", "gaussian_bump.in"], cycles=40,
                overrides=["parthenon/meshblock/nx1=64", "parthenon/meshblock/nx2=64", "gas/conductivity/cond=0.05",
                           "problem/sigma=0.5"])
    r = run_world(1, spec, tmp_path, "c1")[0]
    o = Oracle((64, 64, 1), (-6.0, -6.0, -0.5), (6.0, 6.0, 0.5), ng=2, reconstruct="plm", riemann="hllc",
               gamma=1.000001, dfloor=1e-10, siefloor=1e-10, cfl=0.3,
               bc=("outflow",) * 4 + ("periodic",) * 2, integrator="rk2")
    o.set_conductivity("conductivity", cond=0.05)
    o.pgen_gaussian_bump(sigma=0.5, temperature_bump=5.0)
    e0, s0 = o.history()[4], o.interior(o.gprim)[5].max()
    o.evolve(1.0, 40)
    assert r["meta"]["time"] == o.time and r["meta"]["dt"] == o.dt
    assert np.array_equal(r["blocks"][0][1], o.interior(o.gprim))
    # gamma = 1.000001 makes cv = 1e6: the diffusivity K/(rho cv) is tiny, the peak only just moves
    assert o.interior(o.gprim)[5].max() < s0 and abs(o.history()[4] - e0) < 1e-12 * e0


def test_conduction_problem_deck_driver_equals_oracle(double_lib, tmp_path):
    """inputs/diffusion/conduction.in (conduction pgen, `conductive` user boundary conditions, uniform
    gravity, self damping, heat conduction): driver == oracle for 200 cycles on the deck's block."""
    from oracle.oracle import Oracle
    spec = dict(deck=["diffusion", "conduction.in"], cycles=200, overrides=["gravity/uniform/gx1=-0.02"])
    r = run_world(1, spec, tmp_path, "k1")[0]
    assert r["meta"]["fused"] and not r["meta"]["tuned"] and r["meta"]["nblocks"] == 1
    o = Oracle((128, 1, 1), (0.2, -0.5, -0.5), (1.2, 0.5, 0.5), ng=2, reconstruct="plm", riemann="hllc",
               gamma=1.66667, dfloor=1e-10, siefloor=1e-15, cfl=0.3,
               bc=("conductive", "conductive") + ("periodic",) * 4, integrator="rk2")
    o.set_gravity_uniform(-0.02, 0.0, 0.0)
    o.set_conductivity("conductivity", cond=0.1)
    o.set_drag("self", "constant")
    o.set_damping(0, inner=(4.0, -1.7976931348623157e308, -1.7976931348623157e308), inner_rate=(1.0e4, 0.0, 0.0))
    o.pgen_conduction(gas_rho=1.0, gas_temp=0.05, flux=0.01)
    o.evolve(40.0, 200)
    assert r["meta"]["time"] == o.time and r["meta"]["dt"] == o.dt
    assert np.array_equal(r["blocks"][0][1], o.interior(o.gprim))


def test_disk_deck_driver_equals_oracle_and_ranks_agree(double_lib, tmp_path):
    """inputs/disk/disk_axi.in at half resolution (disk pgen on the host libm, `ic` conditions from
    the stored initial state, point-mass gravity, alpha viscosity with its radial table, the
    curvilinear rotating frame): driver == oracle bit for bit on one block; the 2x2-block run on
    1 and 2 ranks agrees bit for bit with itself."""
    from test_oracle_pins import disk_oracle
    half = ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=32", "problem/polytropic_index=1.40"]
    one = dict(deck=["disk", "disk_axi.in"], cycles=6, overrides=half + ["parthenon/meshblock/nx1=64",
                                                                       "parthenon/meshblock/nx2=32"])
    r = run_world(1, dict(one, path="unfused"), tmp_path, "d1")[0]
    assert not r["meta"]["fused"] and r["meta"]["nblocks"] == 1
    o = disk_oracle("axi", 1.4, "ic", nx=(64, 32, 1))
    o.evolve(62.8, 6)
    assert r["meta"]["time"] == o.time and r["meta"]["dt"] == o.dt
    assert np.array_equal(r["blocks"][0][1], o.interior(o.gprim))
    # the general fused stage (DiffusionUpdate, RotatingFrameImpl from the cell's own mass fluxes) gives
    # the same bits; it is the default for this deck (one gas species on a curvilinear mesh)
    g = run_world(1, one, tmp_path, "d1f")[0]
    assert g["meta"]["fused"] and np.array_equal(g["blocks"][0][1], r["blocks"][0][1])
    four = dict(one, overrides=half)  # the deck's 32-zone blocks: 2 x 1 ... at half resolution 2 x 1
    four["overrides"] = half + ["parthenon/meshblock/nx2=16"]
    a, b = run_world(1, four, tmp_path, "d4"), run_world(2, four, tmp_path, "d4r2")
    assert a[0]["meta"]["nblocks"] == 4 and [x["meta"]["nblocks"] for x in b] == [2, 2]
    xa, xb = by_bounds(a), by_bounds(b)
    for key in xa:
        assert np.array_equal(xa[key], xb[key]), key


def test_disk_deck_damp_to_visc_driver_equals_oracle_and_ranks_agree(double_lib, tmp_path):
    """inputs/disk/disk_axi.in at half resolution with self drag and <gas/damping> damp_to_visc = true (the
    radial damping zones relax the gas towards the alpha-viscosity inflow velocity, drag.cpp:109-121):
    driver == oracle bit for bit on one block; four blocks on 1 and 2 ranks agree bit for bit."""
    from test_oracle_pins import disk_oracle
    big = 1.7976931348623157e308
    ov = ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=32", "problem/polytropic_index=1.40", "physics/drag=true",
          "drag/type=self", "gas/damping/inner_x1=0.6", "gas/damping/inner_x1_rate=30.0", "gas/damping/outer_x1=3.5",
          "gas/damping/outer_x1_rate=30.0", "gas/damping/damp_to_visc=true"]
    one = dict(deck=["disk", "disk_axi.in"], cycles=6, overrides=ov + ["parthenon/meshblock/nx1=64",
                                                                     "parthenon/meshblock/nx2=32"])
    r = run_world(1, one, tmp_path, "v1")[0]
    o = disk_oracle("axi", 1.4, "ic", nx=(64, 32, 1))
    o.set_drag("self", "constant")
    o.set_damping(0, inner=(0.6, -big, -big), inner_rate=(30.0, 0.0, 0.0), outer=(3.5, big, big),
                  outer_rate=(30.0, 0.0, 0.0))
    o.set_damp_to_visc(True)
    o.evolve(62.8, 6)
    assert r["meta"]["time"] == o.time and r["meta"]["dt"] == o.dt
    assert np.array_equal(r["blocks"][0][1], o.interior(o.gprim))
    four = dict(one, overrides=ov + ["parthenon/meshblock/nx2=16"])
    a, b = run_world(1, four, tmp_path, "v4"), run_world(2, four, tmp_path, "v4r2")
    assert a[0]["meta"]["nblocks"] == 4 and [x["meta"]["nblocks"] for x in b] == [2, 2]
    xa, xb = by_bounds(a), by_bounds(b)
    for key in xa:
        assert np.array_equal(xa[key], xb[key]), key


def test_alpha_disk_deck_driver_equals_oracle_and_ranks_agree(double_lib, tmp_path):
    """inputs/diffusion/alpha_disk.in with the overrides of tst/scripts/diffusion/alpha_disk.py (disk pgen with
    mdot, alpha viscosity, beta cooling, `viscous` conditions): driver == oracle bit for bit on one
    block for 400 cycles; two blocks on two ranks agree with the one-block run to round-off."""
    from test_oracle_pins import alpha_disk_oracle
    al, h = 0.1, 0.1
    ov = ["parthenon/mesh/x1max=2.0", "physics/viscosity=true", f"gas/viscosity/alpha={al:.8e}",
          f"cooling/tcyl={h ** 2:.8e}", "cooling/cyl_plaw=-1.0", f"problem/mdot={al * h ** 2 * 3 * np.pi:.8e}",
          "problem/quiet_start=true", f"problem/h0={h:.8e}", "problem/dslope=0.0", "problem/flare=0.0",
          "artemis/coordinates=axisymmetric", "parthenon/mesh/nx1=64", "parthenon/mesh/nx2=1",
          "parthenon/meshblock/nx2=1", "parthenon/mesh/nx3=1", "parthenon/meshblock/nx3=1",
          "parthenon/mesh/x2min=-0.5", "parthenon/mesh/x2max=0.5", "parthenon/time/tlim=8000.0"]
    one = dict(deck=["diffusion", "alpha_disk.in"], cycles=400, overrides=ov + ["parthenon/meshblock/nx1=64"])
    r = run_world(1, one, tmp_path, "a1")[0]
    assert not r["meta"]["fused"] and r["meta"]["nblocks"] == 1
    o = alpha_disk_oracle()
    o.evolve(8e3, 400)
    assert r["meta"]["time"] == o.time and r["meta"]["dt"] == o.dt
    assert np.array_equal(r["blocks"][0][1], o.interior(o.gprim))
    two = run_world(2, dict(one, overrides=ov), tmp_path, "a2")  # the deck's 32-zone blocks, one per rank
    assert [x["meta"]["nblocks"] for x in two] == [1, 1]
    xb = by_bounds(two)
    full = o.interior(o.gprim)
    for key, part in xb.items():
        i0 = int(round((key[0] - 0.3) / 1.7 * 64))
        ref = full[:, :, :, i0:i0 + 32]
        assert np.max(np.abs(part - ref)) < 1e-12, key


def test_binary_deck_driver_equals_oracle_and_ranks_agree(double_lib, tmp_path):
    """inputs/disk/binary_cyl.in at 64 x 128 (binary gravity with the host-evaluated orbit, rotating frame
    incl. the frame velocity in FluxSource, alpha viscosity, self damping, `ic` radial conditions,
    periodic azimuth): driver == oracle bit for bit on one block; 2 x 4 blocks on two ranks (the
    azimuthal wrap-around crosses the rank boundary) agree with the one-block run to round-off."""
    from test_driver_gpu import binary_oracle
    small = ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=128"]
    one = dict(deck=["disk", "binary_cyl.in"], cycles=20,
               overrides=small + ["parthenon/meshblock/nx1=64", "parthenon/meshblock/nx2=128"])
    r = run_world(1, one, tmp_path, "b1")[0]
    assert r["meta"]["fused"] and r["meta"]["nblocks"] == 1  # (since round 4: the stage stops at the conserved state for the damping)
    o = binary_oracle((64, 128, 1))
    o.evolve(2 * np.pi, 20)
    assert r["meta"]["time"] == o.time and r["meta"]["dt"] == o.dt
    assert np.array_equal(r["blocks"][0][1], o.interior(o.gprim))
    two = run_world(2, dict(one, overrides=small), tmp_path, "b2")  # the deck's 32 x 32 blocks: 2 x 4
    assert [x["meta"]["nblocks"] for x in two] == [4, 4]
    full = o.interior(o.gprim)
    scale = np.maximum(np.abs(full).max(axis=(1, 2, 3), keepdims=True), 1e-300)  # v3 is identically zero
    for key, part in by_bounds(two).items():
        i0, j0 = int(round((key[0] - 0.3) / 2.7 * 64)), int(round(key[2] / 6.283185307179586 * 128))
        ref = full[:, :, j0:j0 + 32, i0:i0 + 32]
        assert np.max(np.abs(part - ref) / scale) < 1e-11, key


def test_stratified_box_3d_driver_equals_oracle_and_ranks_agree(double_lib, tmp_path):
    """The strat problem in 3-D (Gaussian vertical stratification, vertical gravity of the shearing
    box, `extrap` on the x3 faces with its power-law density continuation): driver == oracle on one
    block over 12 cycles; 2 x 2 x 2 blocks on two ranks == one rank."""
    from oracle.oracle import Oracle
    box = ["parthenon/mesh/nx1=16", "parthenon/mesh/nx2=16", "parthenon/mesh/nx3=16", "parthenon/mesh/x3min=-0.2",
           "parthenon/mesh/x3max=0.2", "parthenon/mesh/ix3_bc=extrap", "parthenon/mesh/ox3_bc=extrap",
           "gravity/point/mass=1.0e-3"]
    one_blk = dict(deck=["ssheet", "ssheet.in"], cycles=12,
                   overrides=box + ["parthenon/meshblock/nx1=16", "parthenon/meshblock/nx2=16", "parthenon/meshblock/nx3=16"])
    r = run_world(1, one_blk, tmp_path, "z1")[0]
    assert r["meta"]["fused"] and r["meta"]["nblocks"] == 1
    o = Oracle((16, 16, 16), (-1.0, -1.0, -0.2), (1.0, 1.0, 0.2), ng=2, reconstruct="plm", riemann="hllc",
               gamma=1.000001, dfloor=1e-10, siefloor=1e-10, cfl=0.3,
               bc=("extrap", "extrap", "inflow", "inflow", "extrap", "extrap"), integrator="rk2")
    o.set_rotating_frame(1.0, 1.5)
    o.set_gravity_point(1e-3, soft=0.03)
    o.pgen_strat(rho0=1.0, dens_min=1e-10, h=0.05)
    o.evolve(100.0, 12)
    assert r["meta"]["time"] == o.time and r["meta"]["dt"] == o.dt
    assert np.array_equal(r["blocks"][0][1], o.interior(o.gprim))
    rho = o.interior(o.gprim)[0]
    assert rho[8, 8, 8] > 5.0 * rho[0, 8, 8]  # stratified: the midplane is denser than the top layer
    eight = dict(one_blk, overrides=box + ["parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8"])
    a, b = run_world(1, eight, tmp_path, "z8"), run_world(2, eight, tmp_path, "z8r2")
    assert a[0]["meta"]["nblocks"] == 8 and [x["meta"]["nblocks"] for x in b] == [4, 4]
    xa, xb = by_bounds(a), by_bounds(b)
    for key in xa:
        assert np.array_equal(xa[key], xb[key]), key


def test_disk_alpha_deck_driver_equals_oracle(double_lib, tmp_path):
    """inputs/disk/disk_alpha.in (1-D axisymmetric alpha disk, gamma = 1.4, beta cooling with beta0 = 1e-8,
    `viscous` conditions, mdot from the deck) for 300 cycles: driver == oracle bit for bit."""
    from oracle.oracle import Oracle
    spec = dict(deck=["disk", "disk_alpha.in"], cycles=300, overrides=["parthenon/time/nlim=300"])
    r = run_world(1, spec, tmp_path, "da")[0]
    assert r["meta"]["nblocks"] == 1 and not r["meta"]["fused"]
    o = Oracle((128, 1, 1), (0.3, -3.141592653589793, -0.5), (4.3, 3.141592653589793, 0.5), ng=2, reconstruct="plm",
               riemann="hllc", gamma=1.4, dfloor=1e-10, siefloor=1e-10, cfl=0.3, integrator="rk2",
               coordinates="axisymmetric", bc=("viscous", "viscous") + ("periodic",) * 4)
    o.set_gravity_point(mass=1.0)
    o.set_viscosity("alpha", alpha=1e-2, r0=1.0, Omega0=1.0)
    o.set_cooling(beta0=1e-8, tcyl=0.0025, cyl_plaw=-1.0)
    o.pgen_disk(r0=1.0, dslope=-0.5, flare=0.0, h0=0.05, dens_min=1e-10, pres_min=1e-15, polytropic_index=1.0,
                mdot=0.00023561944901923456)
    d0 = o.interior(o.gprim)[0].copy()
    o.evolve(628.0, 300)
    assert r["meta"]["time"] == o.time and r["meta"]["dt"] == o.dt
    assert np.array_equal(r["blocks"][0][1], o.interior(o.gprim))
    assert np.max(np.abs(o.interior(o.gprim)[0] / d0 - 1.0)) < 0.15  # relaxing towards the viscous steady state
