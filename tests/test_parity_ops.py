"""GPU parity: every C-ABI task of libartemis_hip.so against the CPU oracle on the same seeded
inputs.  fp64 with -ffp-contract=off on both sides, IEEE-correct division and sqrt: the bar
is BIT-EXACT (np.array_equal), which is stricter than the L-infinity 1e-12 relative tolerance
BASELINE.md asks for."""
import os

import numpy as np
import pytest
import torch

from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def random_state(o, rng, mach=2.0, contrast=1.0e3, shock=True):
    """Physical random primitives incl. ghosts: large density/pressure contrasts and
    supersonic velocities of both signs so every solver branch (am >= 0 / < 0, bp/bm
    clamps, dq2 <= 0 limiter zeros) is taken somewhere."""
    ns = o.cfg.ns_gas
    shp = (o.nk, o.nj, o.ni)
    shift = int(os.environ.get("ARTEMIS_SEED_SHIFT", "0"))  # sweep other random states: advance the stream
    if shift:
        rng.random(shift)
    if ns:
        p = o.gprim
        for n in range(ns):
            p[n] = np.exp(rng.uniform(-np.log(contrast), np.log(contrast), shp) * 0.5)
            for d in range(3):
                p[ns + 3 * n + d] = rng.normal(0.0, mach, shp)
            p[5 * ns + n] = np.exp(rng.uniform(-np.log(contrast), np.log(contrast), shp) * 0.5)
        if shock:  # a piecewise-constant patch: exercises dq2 == 0 and frho == 0 exactly
            p[:, : o.nk // 2 + 1, : o.nj // 2 + 1, : o.ni // 2] = p[:, :1, :1, :1]
            p[ns:4 * ns, : o.nk // 2 + 1, : o.nj // 2 + 1, : o.ni // 2] = 0.0
    nd = o.cfg.ns_dust
    if nd:
        q = o.dprim
        for n in range(nd):
            q[n] = np.exp(rng.uniform(-3, 3, shp))
            for d in range(3):
                q[nd + 3 * n + d] = rng.normal(0.0, mach, shp)
    o.PrimToCons()


def make_pair(nx, ng=2, ns_gas=1, ns_dust=0, recon="plm", riem="hllc", drecon="plm", driem="hlle",
              gamma=1.4, seed=0, nb=1, dfloor=1e-10, siefloor=1e-10, de_switch=0.0,
              bc=("periodic",) * 6, cfl=0.8):
    from artemis_amd.pack import MeshBlockPack
    rng = np.random.default_rng(seed)
    oracles, xmin, xmax = [], [], []
    for b in range(nb):
        lo = (-1.0 + 0.37 * b, -0.5 - 0.11 * b, 0.25 + b)
        hi = (lo[0] + 2.0, lo[1] + 1.3, lo[2] + 0.7)
        o = Oracle(nx, lo, hi, ng=ng, ns_gas=ns_gas, ns_dust=ns_dust, reconstruct=recon,
                   riemann=riem, dust_reconstruct=drecon, dust_riemann=driem, gamma=gamma,
                   dfloor=dfloor, siefloor=siefloor, de_switch=de_switch, dust_dfloor=dfloor,
                   bc=bc, cfl=cfl, dust_cfl=cfl)
        random_state(o, rng)
        oracles.append(o)
        xmin.append(lo)
        xmax.append(hi)
    mb = MeshBlockPack(nb, nx, xmin, xmax, ng=ng, ns_gas=ns_gas, ns_dust=ns_dust, reconstruct=recon,
                       riemann=riem, dust_reconstruct=drecon, dust_riemann=driem, gamma=gamma,
                       dfloor=dfloor, siefloor=siefloor, de_switch=de_switch, dust_dfloor=dfloor)
    push(oracles, mb)
    return oracles, mb


def push(oracles, mb):
    for b, o in enumerate(oracles):
        if o.cfg.ns_gas:
            mb.gas_prim[b].copy_(torch.from_numpy(o.gprim.copy()))
            mb.gas_u0[b].copy_(torch.from_numpy(o.gu0.copy()))
            mb.gas_u1[b].copy_(torch.from_numpy(o.gu1.copy()))
        if o.cfg.ns_dust:
            mb.dust_prim[b].copy_(torch.from_numpy(o.dprim.copy()))
            mb.dust_u0[b].copy_(torch.from_numpy(o.du0.copy()))
            mb.dust_u1[b].copy_(torch.from_numpy(o.du1.copy()))


def same(a_gpu, b_np, what):
    a = a_gpu.cpu().numpy()
    if not np.array_equal(a, b_np):
        bad = np.argwhere(a != b_np)
        rel = np.max(np.abs(a - b_np) / (np.abs(b_np) + 1e-300))
        raise AssertionError(f"{what}: {len(bad)} mismatches, first at {bad[0]}, max rel {rel:.3e}")


def face_slices(o, d):
    """Cells that hold a face of direction d (fluid_fluxes.hpp:105,130,172)."""
    ks, js, is_ = slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1)
    if d == 0:
        is_ = slice(o.is_, o.ie + 2)
    if d == 1:
        js = slice(o.js, o.je + 2)
    if d == 2:
        ks = slice(o.ks, o.ke + 2)
    return (slice(None), ks, js, is_)


CASES = [  # nx, ng, recon, riemann
    ((24, 12, 10), 2, "plm", "hllc"),
    ((24, 12, 10), 2, "plm", "hlle"),
    ((24, 12, 10), 2, "plm", "llf"),
    ((20, 8, 6), 4, "ppm", "hllc"),
    ((20, 8, 6), 3, "ppm", "hlle"),
    ((20, 8, 6), 2, "pcm", "llf"),
    ((70, 9, 1), 2, "plm", "hllc"),   # 2-D, ragged vs the 64x4 thread tile
    ((131, 1, 1), 2, "plm", "hllc"),  # 1-D
    ((5, 3, 2), 2, "plm", "hllc"),    # tiny block
    # blocks at least a 32 x 8 tile wide: the task runs through the LDS-staged tile march (ragged tiles, chunks)
    ((40, 17, 21), 2, "plm", "hllc"),
    ((40, 17, 21), 2, "plm", "hlle"),
    ((40, 17, 21), 3, "plm", "llf"),
    ((67, 9, 5), 2, "pcm", "hllc"),
    ((64, 16, 1), 2, "plm", "hlle"),
    ((33, 8, 1), 4, "pcm", "llf"),
]


@pytest.mark.parametrize("nx,ng,recon,riem", CASES)
def test_calculate_fluxes_gas(hiplib, nx, ng, recon, riem):
    (o,), mb = make_pair(nx, ng=ng, recon=recon, riem=riem, seed=1)
    o.CalculateFluxes(0, False)
    mb.CalculateFluxes(0, False)
    for d in range(o.ndim):
        sl = face_slices(o, d)
        same(mb.gas_flux[d][0][sl], o.gflux(d)[sl], f"flux x{d+1}")
        same(mb.gas_pflux[d][0][sl], o.gpflux(d)[sl], f"pflux x{d+1}")
        same(mb.gas_vface[d][0][sl], o.gvface(d)[sl], f"vface x{d+1}")


@pytest.mark.parametrize("riem", ["hllc", "hlle", "llf"])
def test_calculate_fluxes_tile_march_equals_per_task_kernel(hiplib, riem, monkeypatch, option):
    """Several blocks wide enough for the tile march: artemis_hip_calculate_fluxes through the march, through the
    one-thread-per-zone kernel (ARTEMIS_NO_TILED_FLUX) and the oracle agree bit for bit on every face output."""
    oracles, mb = make_pair((48, 20, 19), recon="plm", riem=riem, seed=21, nb=3)
    mb.CalculateFluxes(0, False)
    tiled = [[mb.gas_flux[d].clone(), mb.gas_pflux[d].clone(), mb.gas_vface[d].clone()] for d in range(3)]
    option("no_tiled_flux", 1)
    for d in range(3):
        mb.gas_flux[d].zero_(), mb.gas_pflux[d].zero_(), mb.gas_vface[d].zero_()
    mb.CalculateFluxes(0, False)
    for b, o in enumerate(oracles):
        o.CalculateFluxes(0, False)
        for d in range(3):
            sl = face_slices(o, d)
            same(tiled[d][0][b][sl], o.gflux(d)[sl], f"block {b} flux x{d+1}")
            same(tiled[d][1][b][sl], o.gpflux(d)[sl], f"block {b} pflux x{d+1}")
            same(tiled[d][2][b][sl], o.gvface(d)[sl], f"block {b} vface x{d+1}")
            assert torch.equal(tiled[d][0][b][sl], mb.gas_flux[d][b][sl])


@pytest.mark.parametrize("riem", ["hllc", "hlle", "llf"])
def test_calculate_fluxes_tile_march_with_vanishing_velocities(hiplib, riem):
    """The flux task keeps every output bit exact next to velocities of 1e-150 .. 1e-320 (shock precursors): the
    tile march notes tile planes and columns that hold such a velocity and takes the IEEE divisions there."""
    (o,), mb = make_pair((48, 20, 19), recon="plm", riem=riem, seed=31)
    rng = np.random.default_rng(7)
    w = o.gprim
    scale = rng.choice([1.0, 0.0, 1e-300, 1e-306, 1e-250, 1e-160, 1e-150, 1e-100, 1e-40], size=w[1].shape,
                       p=[0.3, 0.1, 0.1, 0.1, 0.08, 0.08, 0.08, 0.08, 0.08])
    for v in (1, 2, 3):
        w[v] *= scale
    o.PrimToCons()
    push([o], mb)
    o.CalculateFluxes(0, False)
    mb.CalculateFluxes(0, False)
    for d in range(3):
        sl = face_slices(o, d)
        same(mb.gas_flux[d][0][sl], o.gflux(d)[sl], f"flux x{d+1}")
        same(mb.gas_pflux[d][0][sl], o.gpflux(d)[sl], f"pflux x{d+1}")
        same(mb.gas_vface[d][0][sl], o.gvface(d)[sl], f"vface x{d+1}")


def test_calculate_fluxes_pcm_override(hiplib):
    # artemis_driver.cpp:182: VL2 stage 1 forces PCM whatever gas/reconstruct says
    (o,), mb = make_pair((16, 8, 8), recon="plm", riem="hllc", seed=2)
    o.CalculateFluxes(0, True)
    mb.CalculateFluxes(0, True)
    for d in range(3):
        sl = face_slices(o, d)
        same(mb.gas_flux[d][0][sl], o.gflux(d)[sl], f"flux x{d+1}")


@pytest.mark.parametrize("driem", ["hlle", "llf"])
@pytest.mark.parametrize("drecon", ["plm", "ppm"])
def test_calculate_fluxes_dust_and_species(hiplib, driem, drecon):
    (o,), mb = make_pair((18, 10, 6), ng=4, ns_gas=2, ns_dust=3, recon="plm", riem="hlle",
                         drecon=drecon, driem=driem, seed=3)
    for fluid in (0, 1):
        o.CalculateFluxes(fluid, False)
        mb.CalculateFluxes(fluid, False)
    for d in range(3):
        sl = face_slices(o, d)
        same(mb.gas_flux[d][0][sl], o.gflux(d)[sl], f"gas flux x{d+1}")
        same(mb.dust_flux[d][0][sl], o.dflux(d)[sl], f"dust flux x{d+1}")


def run_stage_chain(oracles, mb, g0, g1, be, dt, bc, pcm=False):
    for o in oracles:
        for fluid in (0, 1):
            o.CalculateFluxes(fluid, pcm)
        o.ApplyUpdate(g0, g1, be * dt)
        o.FluxSource(be * dt)
        o.SetAuxillaryFields()
        o.ConsToPrim()
        o.ApplyBoundaryConditions()
        o.PrimToCons()
    for fluid in (0, 1):
        mb.CalculateFluxes(fluid, pcm)
    mb.ApplyUpdate(g0, g1, be * dt)
    mb.FluxSource(be * dt)
    mb.SetAuxillaryFields()
    mb.ConsToPrim()
    mb.ApplyBoundaryConditions(bc)
    mb.PrimToCons()


def compare_state(oracles, mb, interior_cons_only=True):
    for b, o in enumerate(oracles):
        if o.cfg.ns_gas:
            same(mb.gas_prim[b], o.gprim, f"gas prim block {b}")
            same(mb.gas_u0[b], o.gu0, f"gas cons block {b}")
        if o.cfg.ns_dust:
            same(mb.dust_prim[b], o.dprim, f"dust prim block {b}")
            same(mb.dust_u0[b], o.du0, f"dust cons block {b}")


@pytest.mark.parametrize("nx,ng,recon,riem", CASES[:7])
def test_individual_tasks(hiplib, nx, ng, recon, riem):
    """ApplyUpdate, FluxSource, SetAuxillaryFields, ConsToPrim, PrimToCons one at a time."""
    (o,), mb = make_pair(nx, ng=ng, recon=recon, riem=riem, seed=4)
    o.DeepCopyConservedData()
    mb.DeepCopyConservedData()
    same(mb.gas_u1[0], o.gu1, "u1 after DeepCopyConservedData")
    o.CalculateFluxes(0, False)
    mb.CalculateFluxes(0, False)
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    dt = 1.0e-4
    o.ApplyUpdate(0.5, 0.5, 0.5 * dt)
    mb.ApplyUpdate(0.5, 0.5, 0.5 * dt)
    same(mb.gas_u0[0][I], o.gu0[I], "ApplyUpdate")
    o.FluxSource(0.5 * dt)
    mb.FluxSource(0.5 * dt)
    same(mb.gas_u0[0][I], o.gu0[I], "FluxSource (interior)")
    o.SetAuxillaryFields()
    mb.SetAuxillaryFields()
    same(mb.gas_u0[0][I], o.gu0[I], "SetAuxillaryFields")
    o.ConsToPrim()
    mb.ConsToPrim()
    same(mb.gas_prim[0][I], o.gprim[I], "ConsToPrim")
    o.PrimToCons()
    mb.PrimToCons()
    same(mb.gas_prim[0], o.gprim, "PrimToCons prim (entire)")
    same(mb.gas_u0[0], o.gu0, "PrimToCons cons (entire)")


@pytest.mark.parametrize("bcname", ["outflow", "periodic", "reflecting"])
def test_boundary_conditions(hiplib, bcname):
    rng = np.random.default_rng(5)
    bc = (bcname,) * 6
    from artemis_amd.pack import MeshBlockPack
    o = Oracle((12, 10, 8), (0, 0, 0), (1, 1, 1), ng=3, ns_gas=1, ns_dust=2, bc=bc, gamma=1.4)
    random_state(o, rng, shock=False)
    mb = MeshBlockPack(1, (12, 10, 8), [(0, 0, 0)], [(1, 1, 1)], ng=3, ns_gas=1, ns_dust=2, gamma=1.4)
    push([o], mb)
    o.ApplyBoundaryConditions()
    mb.ApplyBoundaryConditions([bc])
    same(mb.gas_prim[0], o.gprim, "gas ghosts")
    same(mb.dust_prim[0], o.dprim, "dust ghosts")


def test_mixed_bcs_2d(hiplib):
    rng = np.random.default_rng(6)
    bc = ("periodic", "periodic", "reflecting", "outflow", "outflow", "outflow")
    from artemis_amd.pack import MeshBlockPack
    o = Oracle((16, 12, 1), (0, 0, 0), (1, 1, 1), ng=2, bc=bc, gamma=1.4)
    random_state(o, rng, shock=False)
    mb = MeshBlockPack(1, (16, 12, 1), [(0, 0, 0)], [(1, 1, 1)], ng=2, gamma=1.4)
    push([o], mb)
    o.ApplyBoundaryConditions()
    mb.ApplyBoundaryConditions([bc])
    same(mb.gas_prim[0], o.gprim, "gas ghosts")


@pytest.mark.parametrize("fluid,ns_gas,ns_dust", [(0, 1, 0), (0, 2, 1), (1, 1, 2)])
def test_estimate_timestep(hiplib, fluid, ns_gas, ns_dust):
    # gas.cpp:411-433,467 / dust.cpp:256-275: cfl * min over interior cells
    (o,), mb = make_pair((33, 7, 5), ns_gas=ns_gas, ns_dust=ns_dust, riem="hlle", seed=7, cfl=0.3)
    assert mb.EstimateTimestepMesh(fluid, cfl=0.3) == o.EstimateTimestepMesh(fluid)


def test_multiblock_pack_full_stage(hiplib):
    """Two blocks with different coordinates in one pack, gas (2 species) + dust, RK2 stage 2
    weights, outflow everywhere: the whole reference stage chain, bit for bit."""
    oracles, mb = make_pair((16, 8, 6), ng=2, ns_gas=2, ns_dust=1, recon="plm", riem="hlle",
                            seed=8, nb=2, bc=("outflow",) * 6)
    for o in oracles:
        o.DeepCopyConservedData()
    push(oracles, mb)
    run_stage_chain(oracles, mb, 0.5, 0.5, 0.5, 2.0e-4, [("outflow",) * 6] * 2)
    compare_state(oracles, mb)


@pytest.mark.parametrize("integ", ["rk2", "vl2", "rk3"])
def test_multi_step_blast_matches_oracle(hiplib, integ):
    """3-D Sedov deck at 32x24x16, 6 full steps with dt from the oracle: the GPU chain driven
    task by task must reproduce the oracle's state exactly (primitives and conserved)."""
    from artemis_amd.pack import MeshBlockPack
    nx = (32, 24, 16)
    kw = dict(ng=2, reconstruct="plm", riemann="hllc", gamma=1.4, dfloor=1e-10, siefloor=1e-10)
    o = Oracle(nx, (-1, -1, -1), (1, 1, 1), cfl=0.3, bc=("outflow",) * 6, integrator=integ, **kw)
    o.pgen_blast(radius=0.25, internal_energy=1.0, p0=1e-5, d0=1.0, samples=4)
    mb = MeshBlockPack(1, nx, [(-1, -1, -1)], [(1, 1, 1)], **kw)
    mb.gas_prim[0].copy_(torch.from_numpy(o.gprim.copy()))
    mb.PrimToCons()
    same(mb.gas_u0[0], o.gu0, "initial PrimToCons")
    coeff = {"rk2": [(0.0, 1.0, 1.0), (0.5, 0.5, 0.5)],
             "vl2": [(0.0, 1.0, 0.5), (0.0, 1.0, 1.0)],
             "rk3": [(0.0, 1.0, 1.0), (0.25, 0.75, 0.25), (2.0 / 3.0, 1.0 / 3.0, 2.0 / 3.0)]}[integ]
    for step in range(6):
        dt = o.new_dt()
        got = mb.EstimateTimestepMesh(0, cfl=0.3)
        assert got == dt, (got, dt)
        o.dt = dt
        o.step()
        mb.DeepCopyConservedData()
        for s, (g0, g1, be) in enumerate(coeff):
            pcm = (integ == "vl2" and s == 0)
            mb.CalculateFluxes(0, pcm)
            mb.ApplyUpdate(g0, g1, be * dt)
            mb.FluxSource(be * dt)
            mb.SetAuxillaryFields()
            mb.ConsToPrim()
            mb.ApplyBoundaryConditions([("outflow",) * 6])
            mb.PrimToCons()
        same(mb.gas_prim[0], o.gprim, f"prim after step {step}")
        same(mb.gas_u0[0], o.gu0, f"cons after step {step}")


def test_halo_pack_unpack_roundtrip(hiplib):
    """Periodic images via pack -> unpack on the opposite face equal the periodic BC."""
    rng = np.random.default_rng(9)
    from artemis_amd.pack import MeshBlockPack
    nx = (10, 8, 6)
    o = Oracle(nx, (0, 0, 0), (1, 1, 1), ng=2, ns_gas=1, ns_dust=1, bc=("periodic",) * 6, gamma=1.4)
    random_state(o, rng, shock=False)
    mb = MeshBlockPack(1, nx, [(0, 0, 0)], [(1, 1, 1)], ng=2, ns_gas=1, ns_dust=1, gamma=1.4)
    push([o], mb)
    o.ApplyBoundaryConditions()
    for face in range(6):
        n = mb.halo_count(face)
        buf = torch.empty(n, dtype=torch.float64, device="cuda")
        mb.halo_pack(0, face, buf)
        mb.halo_unpack(0, face ^ 1, buf)  # my ox1 slab is the ix1 ghost of my periodic image
    I = np.s_[:, o.ks:o.ke + 1, o.js:o.je + 1, :]
    got = mb.gas_prim[0].cpu().numpy()
    # face slabs span the interior extent of the other dimensions (no edges/corners)
    assert np.array_equal(got[:, o.ks:o.ke + 1, o.js:o.je + 1, :][[0, 1, 2, 3, 5]],
                          o.gprim[:, o.ks:o.ke + 1, o.js:o.je + 1, :][[0, 1, 2, 3, 5]])
    assert np.array_equal(got[:, o.ks:o.ke + 1, :, o.is_:o.ie + 1][[0, 1, 2, 3, 5]],
                          o.gprim[:, o.ks:o.ke + 1, :, o.is_:o.ie + 1][[0, 1, 2, 3, 5]])
    assert np.array_equal(got[:, :, o.js:o.je + 1, o.is_:o.ie + 1][[0, 1, 2, 3, 5]],
                          o.gprim[:, :, o.js:o.je + 1, o.is_:o.ie + 1][[0, 1, 2, 3, 5]])
    del I


def test_error_paths_on_gpu(hiplib):
    import ctypes as C
    from artemis_amd import capi
    (o,), mb = make_pair((8, 8, 8), ng=2, recon="ppm", riem="hllc", seed=10)
    # PPM with 2 ghost cells: gas.cpp:69-71 "PPM requires at least 3 ghost cells."
    rc = hiplib.artemis_hip_calculate_fluxes(C.byref(mb.pack), 0, 0, None)
    assert rc == capi.EINVAL and b"PPM requires at least 3 ghost cells" in hiplib.artemis_hip_last_error()
    mb.pack.gas.recon = 7
    rc = hiplib.artemis_hip_calculate_fluxes(C.byref(mb.pack), 0, 0, None)
    assert rc == capi.EINVAL and b"Reconstruction method not recognized" in hiplib.artemis_hip_last_error()
    mb.pack.gas.recon = 1
    mb.pack.gas.riemann = 5
    rc = hiplib.artemis_hip_calculate_fluxes(C.byref(mb.pack), 0, 0, None)
    assert rc == capi.EINVAL and b"Riemann solver not recognized" in hiplib.artemis_hip_last_error()
    rc = hiplib.artemis_hip_calculate_fluxes(C.byref(mb.pack), 3, 0, None)
    assert rc == capi.EINVAL and b"Fluid type not recognized" in hiplib.artemis_hip_last_error()


@pytest.mark.parametrize("de_switch", [0.0, 0.05])
def test_floors_and_energy_switch(hiplib, de_switch):
    """Edge cases of the conserved -> primitive chain: densities below dfloor (zero and negative),
    total energy below the kinetic energy (negative thermal energy -> the internal-energy branch of
    GetSpecificInternalEnergy, artemis_utils.hpp:57-59), internal energies below siefloor*D, with
    de_switch = 0 and > 0 (gas.cpp:177): SetAuxillaryFields, ConsToPrim and PrimToCons must apply the
    same floors as the reference, gas and dust."""
    (o,), mb = make_pair((20, 10, 6), ns_gas=2, ns_dust=2, riem="hlle", seed=61, dfloor=1e-3, siefloor=1e-2,
                         de_switch=de_switch)
    rng = np.random.default_rng(62)
    shp = o.gu0.shape[1:]
    for arr in (o.gu0, o.du0):
        ns = arr.shape[0] // (6 if arr is o.gu0 else 4)
        for n in range(ns):
            d = arr[n]
            d[rng.random(shp) < 0.15] = 0.0            # vacuum
            d[rng.random(shp) < 0.15] = -0.5           # negative density after a too-large step
            d[rng.random(shp) < 0.15] = 5e-4           # below the floor
    E, eg = o.gu0[8:10], o.gu0[10:12]
    E[rng.random(E.shape) < 0.3] *= 1e-6               # E < KE: negative thermal energy
    E[rng.random(E.shape) < 0.1] = -1.0
    eg[rng.random(eg.shape) < 0.3] = 1e-9              # below siefloor * D
    eg[rng.random(eg.shape) < 0.1] = -2.0
    push([o], mb)
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    o.SetAuxillaryFields(), mb.SetAuxillaryFields()
    same(mb.gas_u0[0][I], o.gu0[I], "SetAuxillaryFields with floors")
    o.ConsToPrim(), mb.ConsToPrim()
    same(mb.gas_prim[0][I], o.gprim[I], "ConsToPrim gas with floors")
    same(mb.dust_prim[0][I], o.dprim[I], "ConsToPrim dust with floors")
    assert (o.gprim[0][I[1:]] >= 1e-3).all() and (o.gprim[10][I[1:]] >= 1e-2).all()
    # primitives below the floors in ghost zones (e.g. from an extrapolating boundary condition)
    o.gprim[0, :, :, :2] = 1e-5
    o.gprim[10, :, :, -2:] = 1e-7
    o.dprim[1, :2] = -1.0
    push([o], mb)
    o.PrimToCons(), mb.PrimToCons()
    same(mb.gas_prim[0], o.gprim, "PrimToCons floors prim")
    same(mb.gas_u0[0], o.gu0, "PrimToCons floors cons")
    same(mb.dust_u0[0], o.du0, "PrimToCons floors dust")
