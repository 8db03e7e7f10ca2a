"""GPU parity of the cell-centred general fused stage (artemis_hip_stage_general) against the
oracle's task chain (artemis_driver.cpp:182-255): gas and dust, several species, PCM/PLM/PPM,
HLLC/HLLE/LLF, all coordinate systems, gravity + shearing box + drag.  BIT-EXACT."""
import numpy as np
import pytest
import torch

from oracle.oracle import Oracle
from test_parity_ops import push, random_state, same

pytestmark = pytest.mark.gpu


def build(nx, lo, hi, ns_gas, ns_dust, recon, riem, driem, coordinates, ng, seed):
    from artemis_amd.pack import MeshBlockPack
    kw = dict(ng=ng, ns_gas=ns_gas, ns_dust=ns_dust, reconstruct=recon, riemann=riem,
              dust_reconstruct=recon, dust_riemann=driem, gamma=1.4, dfloor=1e-10, siefloor=1e-10,
              dust_dfloor=1e-10, coordinates=coordinates)
    o = Oracle(nx, lo, hi, bc=("outflow",) * 6, **kw)
    random_state(o, np.random.default_rng(seed), shock=True)
    mb = MeshBlockPack(1, nx, [lo], [hi], with_fluxes=False, **kw)
    push([o], mb)
    return o, mb


def oracle_stage(o, g0, g1, be, dt, pcm, time, grav, rf, drag):
    for fluid in (0, 1):
        o.CalculateFluxes(fluid, pcm)
    o.ApplyUpdate(g0, g1, be * dt)
    for fluid in (0, 1):
        o.FluxSource(be * dt, fluid)
    if grav:
        o.ExternalGravity(time, be * dt)
    if rf:
        o.RotatingFrameForce(be * dt)
    if drag:
        o.DragSource(be * dt)
    o.SetAuxillaryFields()
    o.ConsToPrim()


CASES = [
    # nx, lo, hi, ns_gas, ns_dust, recon, riem, driem, coords, ng
    ((24, 12, 10), (-1, -0.5, 0.25), (1, 0.8, 0.95), 1, 0, "plm", "hllc", "hlle", "cartesian", 2),
    ((24, 12, 10), (-1, -0.5, 0.25), (1, 0.8, 0.95), 2, 3, "plm", "hlle", "hlle", "cartesian", 2),
    ((20, 8, 6), (-1, -0.5, 0.25), (1, 0.8, 0.95), 1, 2, "ppm", "llf", "llf", "cartesian", 3),
    ((20, 8, 6), (-1, -0.5, 0.25), (1, 0.8, 0.95), 1, 1, "pcm", "hllc", "hlle", "cartesian", 2),
    ((70, 9, 1), (-1, -0.5, -0.5), (1, 0.8, 0.5), 1, 1, "plm", "hllc", "hlle", "cartesian", 2),
    ((131, 1, 1), (0, -0.5, -0.5), (1, 0.5, 0.5), 0, 2, "plm", "hlle", "llf", "cartesian", 2),
    # 2-D Cartesian, one gas species (+ <= 2 dust): the row-march kernel (kernels_stage2d.hip); strips of 60 columns,
    # ragged last strip, several row chunks
    ((130, 37, 1), (-1, -0.5, -0.5), (1, 0.8, 0.5), 1, 0, "plm", "hlle", "hlle", "cartesian", 2),
    ((64, 70, 1), (-1, -0.5, -0.5), (1, 0.8, 0.5), 1, 2, "pcm", "llf", "llf", "cartesian", 2),
    ((61, 40, 1), (-1, -0.5, -0.5), (1, 0.8, 0.5), 1, 1, "plm", "hllc", "llf", "cartesian", 3),
    ((121, 33, 1), (-1, -0.5, -0.5), (1, 0.8, 0.5), 1, 2, "plm", "hllc", "hlle", "cartesian", 2),
    ((24, 10, 1), (0.4, 0.5, -0.5), (2.5, 2.6, 0.5), 1, 1, "plm", "hlle", "hlle", "spherical", 2),
    ((16, 8, 6), (0.3, 0.7, 0.0), (1.7, 2.5, 6.0), 2, 1, "plm", "hllc", "hlle", "spherical", 2),
    ((40, 1, 1), (0.0, 0.0, -0.5), (1.0, np.pi, 0.5), 1, 1, "ppm", "hlle", "hlle", "spherical", 3),
    ((16, 8, 6), (0.5, 0.0, -1.0), (2.0, 6.0, 1.0), 1, 2, "plm", "llf", "hlle", "cylindrical", 2),
    ((24, 12, 1), (0.0, -1.0, -0.5), (2.0, 1.0, 0.5), 1, 1, "plm", "hlle", "hlle", "axisymmetric", 2),
    # one gas species on a curvilinear mesh: the streaming tile kernel's curvilinear instantiation
    # (kernels_fused.hip; ragged tiles in x1 and x2, several x3 chunks, every solver, PCM / PLM, 1-D / 2-D / 3-D)
    ((40, 19, 37), (0.3, 0.7, 0.0), (1.7, 2.5, 6.0), 1, 0, "plm", "hllc", "hlle", "spherical", 2),
    ((35, 10, 18), (0.5, 0.0, -1.0), (2.0, 6.0, 1.0), 1, 0, "plm", "hlle", "hlle", "cylindrical", 2),
    ((33, 9, 5), (0.3, 0.7, 0.0), (1.7, 2.5, 6.0), 1, 0, "pcm", "llf", "hlle", "spherical", 2),
    ((70, 21, 1), (0.4, 0.5, -0.5), (2.5, 2.6, 0.5), 1, 0, "plm", "llf", "hlle", "spherical", 3),
    ((45, 12, 1), (0.2, -1.0, -0.5), (2.0, 1.0, 0.5), 1, 0, "plm", "hllc", "hlle", "axisymmetric", 2),
    ((45, 12, 1), (0.5, 0.0, -0.5), (2.0, 6.0, 0.5), 1, 0, "pcm", "hlle", "hlle", "cylindrical", 2),
    ((77, 1, 1), (0.1, 0.0, -0.5), (1.0, np.pi, 0.5), 1, 0, "plm", "hlle", "hlle", "spherical", 2),
    # one gas + ONE dust species on a curvilinear mesh: the dust species on the same march (DUST instantiations of
    # stage_curv_kernel: HLLE and LLF, PCM and PLM, 3-D with several x3 chunks and ragged tiles, 2-D, 1-D)
    ((35, 18, 20), (0.5, 0.0, -1.0), (2.0, 6.0, 1.0), 1, 1, "plm", "llf", "llf", "cylindrical", 2),
    ((34, 12, 20), (0.3, 0.7, 0.0), (1.7, 2.5, 6.0), 1, 1, "pcm", "hlle", "hlle", "spherical", 2),
    ((40, 19, 21), (0.3, 0.7, 0.0), (1.7, 2.5, 6.0), 1, 1, "plm", "hllc", "hlle", "spherical", 2),
    ((45, 12, 1), (0.5, 0.0, -0.5), (2.0, 6.0, 0.5), 1, 1, "plm", "hlle", "llf", "cylindrical", 2),
    ((77, 1, 1), (0.1, 0.0, -0.5), (1.0, np.pi, 0.5), 1, 1, "plm", "hlle", "hlle", "spherical", 2),
    # one gas species on a 3-D CARTESIAN block through artemis_hip_stage_general: the same tile march, SYS = cartesian
    # (plain PLM, every metric factor 1; round 6 -- the deck-level users are Cartesian packs with gravity / viscosity,
    # e.g. inputs/disk/disk_cart.in): 32 x 8 and 16 x 16 tiles, ragged in x1 and x2, several x3 chunks, ng 2 and 4
    ((40, 19, 37), (-1, -0.5, 0.25), (1, 0.8, 0.95), 1, 0, "plm", "hlle", "hlle", "cartesian", 2),
    ((16, 16, 8), (-1, -0.5, 0.25), (1, 0.8, 0.95), 1, 0, "plm", "hllc", "hlle", "cartesian", 4),
    ((33, 9, 5), (-1, -0.5, 0.25), (1, 0.8, 0.95), 1, 0, "pcm", "llf", "hlle", "cartesian", 2),
]
CURV_TILE_CASES = range(15, 22)


@pytest.mark.parametrize("nx,lo,hi,nsg,nsd,recon,riem,driem,coords,ng", CASES)
@pytest.mark.parametrize("stage2", [False, True])
def test_general_stage_hydro(hiplib, nx, lo, hi, nsg, nsd, recon, riem, driem, coords, ng, stage2):
    """No source packages: stage 1 weights (u1 = in) and RK2 stage 2 weights with a distinct u1."""
    o, mb = build(nx, lo, hi, nsg, nsd, recon, riem, driem, coords, ng, seed=31)
    gin, din = mb.gas_prim_table, mb.dust_prim_table
    _, gout = mb.new_prim_buffer("o") if nsg else (None, None)
    dbuf, dout = mb.new_dust_prim_buffer("o") if nsd else (None, None)
    gu1, du1 = gin, din
    o.DeepCopyConservedData()
    if stage2:  # a different start-of-step state: perturb a copy of the prims for u1
        rng = np.random.default_rng(5)
        o2, mb2 = build(nx, lo, hi, nsg, nsd, recon, riem, driem, coords, ng, seed=77)
        if nsg:
            o.gu1[:] = o2.gu0
            t, gu1 = mb.new_prim_buffer("u1")
            t.copy_(mb2.gas_prim)
        if nsd:
            o.du1[:] = o2.du0
            t, du1 = mb.new_dust_prim_buffer("u1")
            t.copy_(mb2.dust_prim)
    g0, g1, be = (0.5, 0.5, 0.5) if stage2 else (0.0, 1.0, 1.0)
    dt = 1.0e-4
    oracle_stage(o, g0, g1, be, dt, False, 0.0, False, False, False)
    mb.stage_general(g0, g1, be * dt, be * dt, gas=(gin, gu1, gout), dust=(din, du1, dout))
    if coords != "cartesian" and nsg == 1 and nsd <= 1 and recon != "ppm":
        assert mb.last_stage_variant == 3  # the curvilinear tile march (kernels_curv.hip) really ran
    if coords == "cartesian" and nx[2] > 1 and nsg == 1 and nsd == 0 and recon != "ppm":
        assert mb.last_stage_variant == 3  # ... and its Cartesian instantiation
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    if nsg:
        out = mb._extra_prim["o"][0][0][I].cpu().numpy()
        ref = o.gprim[I]
        keep = [v for v in range(6 * nsg) if not (4 * nsg <= v < 5 * nsg)]  # P is not written
        assert np.array_equal(out[keep], ref[keep]), "gas prim"
    if nsd:
        same(dbuf[0][I], o.dprim[I], "dust prim")


@pytest.mark.parametrize("case", [6, 9])
def test_row_march_kernel_with_vanishing_velocities(hiplib, case):
    """The 2-D row-march kernel shares the tuned Cartesian kernel's hand-scheduled divisions; a wave whose five-row
    window holds a gas or dust velocity below 2^-200 (or whose updated momenta come out below 2^-480) takes IEEE
    divisions for that row (kernels_stage2d.hip `fast`).  Next to velocities of 1e-150 .. 1e-320, exact zeros and
    ordinary values every gas and dust value equals the oracle's bit for bit."""
    nx, lo, hi, nsg, nsd, recon, riem, driem, coords, ng = CASES[case]
    o, mb = build(nx, lo, hi, nsg, nsd, recon, riem, driem, coords, ng, seed=33)
    rng = np.random.default_rng(9)
    choices, prob = [1.0, 0.0, 1e-300, 1e-306, 1e-250, 1e-160, 1e-150, 1e-100, 1e-40], [0.3, 0.1, 0.1, 0.1, 0.08, 0.08, 0.08, 0.08, 0.08]
    for arr, ns in ((o.gprim, nsg), (o.dprim if nsd else None, nsd)):
        if arr is None:
            continue
        scale = rng.choice(choices, size=arr[0].shape, p=prob)
        for v in range(ns, 4 * ns):
            arr[v] *= scale
    o.PrimToCons()
    push([o], mb)
    gin, din = mb.gas_prim_table, mb.dust_prim_table
    _, gout = mb.new_prim_buffer("o")
    dbuf, dout = mb.new_dust_prim_buffer("o") if nsd else (None, None)
    o.DeepCopyConservedData()
    dt = 1.0e-4
    oracle_stage(o, 0.0, 1.0, 1.0, dt, False, 0.0, False, False, False)
    mb.stage_general(0.0, 1.0, dt, dt, gas=(gin, gin, gout), dust=(din, din, dout))
    assert mb.last_stage_variant == 1  # the row-march kernel really ran
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    keep = [v for v in range(6 * nsg) if not (4 * nsg <= v < 5 * nsg)]  # P is not written
    pairs = [(mb._extra_prim["o"][0][0][I].cpu().numpy()[keep], o.gprim[I][keep], "gas")]
    if nsd:
        pairs.append((dbuf[0][I].cpu().numpy(), o.dprim[I], "dust"))
    for got, ref, what in pairs:
        bad = got != ref
        assert not bad.any(), f"{what}: {np.count_nonzero(bad)} entries differ, the largest of magnitude {np.abs(ref[bad]).max():.3e}"


@pytest.mark.parametrize("case", [1, 4, 5, 7, 8, 9])
def test_general_stage_with_sources_and_drag(hiplib, case):
    """Cartesian: point-mass gravity + shearing box + simple_dust drag in one stage, with the
    fused dt estimate of the new state."""
    from artemis_amd.pack import drag_params, gravity_point
    nx, lo, hi, nsg, nsd, recon, riem, driem, coords, ng = CASES[case]
    nsg = 1
    nsd = max(nsd, 2)
    from artemis_amd.pack import MeshBlockPack
    kw = dict(ng=ng, ns_gas=nsg, ns_dust=nsd, reconstruct=recon, riemann=riem, dust_reconstruct=recon,
              dust_riemann=driem, gamma=1.4, dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10)
    o = Oracle(nx, lo, hi, bc=("outflow",) * 6, cfl=0.3, dust_cfl=0.4, **kw)
    random_state(o, np.random.default_rng(41), shock=True)
    mb = MeshBlockPack(1, nx, [lo], [hi], with_fluxes=False, **kw)
    push([o], mb)
    o.DeepCopyConservedData()
    tau = [0.05, 2.0, 0.0][:nsd]
    o.set_gravity_point(0.7, soft=0.1, x=0.1, y=0.05, z=0.0)
    o.set_rotating_frame(1.2, 1.5)
    o.set_drag("simple_dust", "constant", tau=tau)
    grav = gravity_point(0.7, soft=0.1, pos=(0.1, 0.05, 0.0))
    drag = drag_params("simple_dust", "constant", tau=tau, mesh_min=lo, mesh_max=hi)
    dt = 2.0e-4
    oracle_stage(o, 0.0, 1.0, 1.0, dt, False, 0.25, True, True, True)
    gbuf, gout = mb.new_prim_buffer("o")
    dbuf, dout = mb.new_dust_prim_buffer("o")
    dtd = torch.full((1,), 1.7976931348623157e308, dtype=torch.float64, device="cuda")
    mb.stage_general(0.0, 1.0, dt, dt, gas=(mb.gas_prim_table, mb.gas_prim_table, gout),
                     dust=(mb.dust_prim_table, mb.dust_prim_table, dout), time=0.25, gravity=grav,
                     rotating_frame=(1.2, 1.5), drag=drag, cfl=(0.3, 0.4), dt_dev=dtd.data_ptr())
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    keep = [0, 1, 2, 3, 5]
    assert np.array_equal(gbuf[0][I].cpu().numpy()[keep], o.gprim[I][keep])
    same(dbuf[0][I], o.dprim[I], "dust prim")
    assert dtd.item() == min(o.EstimateTimestepMesh(0), o.EstimateTimestepMesh(1))


EXTRA_BLOCKS = [
    ("cartesian", (24, 12, 10), (-1.0, -0.5, 0.25), (1.0, 0.8, 0.95)),
    ("cartesian", (33, 9, 1), (0.4, -0.5, -0.5), (2.0, 0.8, 0.5)),
    ("spherical", (16, 8, 8), (0.9, 1.06, -3.1), (5.6, 2.08, 3.1)),
    ("spherical", (24, 10, 1), (0.6, 1.06, -0.5), (5.6, 2.08, 0.5)),
    ("cylindrical", (16, 8, 6), (0.8, -3.1, -1.0), (4.3, 3.1, 1.0)),
    ("axisymmetric", (24, 12, 1), (0.6, -2.0, -0.5), (4.3, 2.0, 0.5)),
]


@pytest.mark.parametrize("coordinates,nx,lo,hi", EXTRA_BLOCKS)
def test_general_stage_with_diffusion_rotating_frame_and_cooling(hiplib, coordinates, nx, lo, hi):
    """The optional tasks of the general stage: DiffusionUpdate from the stored viscous + thermal fluxes
    (alpha viscosity, bulk viscosity, conduction), the rotating frame (shearing box in Cartesian,
    RotatingFrameImpl from the cell's own mass fluxes elsewhere, plus the frame velocity in
    FluxSource), point-mass gravity and beta cooling, gas and dust, against the oracle's task chain."""
    from artemis_amd.pack import MeshBlockPack, diffusion_params, gravity_point
    cart = coordinates == "cartesian"
    kw = dict(ng=2, ns_gas=2, ns_dust=1, reconstruct="plm", riemann="hlle", dust_reconstruct="plm",
              dust_riemann="hlle", gamma=1.4, dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10,
              coordinates=coordinates)
    o = Oracle(nx, lo, hi, bc=("outflow",) * 6, cfl=0.3, dust_cfl=0.3, **kw)
    random_state(o, np.random.default_rng(91), shock=False, mach=0.5, contrast=10.0)
    om, q = 0.8, (1.5 if cart else 0.0)
    mb = MeshBlockPack(1, nx, [lo], [hi], with_diffusion=True, omega_frame=om, **kw)
    push([o], mb)
    o.DeepCopyConservedData()
    pos = (0.1, 0.05, 0.0) if coordinates in ("cartesian", "cylindrical") or nx[2] > 1 else (0.0, 0.0, 0.0)
    o.set_gravity_point(1.3, soft=0.05, x=pos[0], y=pos[1], z=pos[2])
    o.set_rotating_frame(om, q)
    o.set_viscosity("alpha", alpha=2e-2, eta_bulk=0.3, r0=0.9, Omega0=1.2)
    o.set_conductivity("conductivity", cond=0.03, averaging="harmonic")
    ckw = dict(beta0=2.0, beta_min=1e-3, exp_scale=0.3, tfloor=1e-3, tcyl=0.02, cyl_plaw=-1.0, tsph=0.01, sph_plaw=-0.5)
    o.set_cooling(**ckw)
    D = diffusion_params(1.4, viscosity=dict(type="alpha", alpha=2e-2, eta_bulk=0.3, r0=0.9, Omega0=1.2),
                         conductivity=dict(type="conductivity", cond=0.03, averaging="harmonic"))
    mb.viscosity_radial_table(D)
    cool = mb.cooling_params(1.4, 1.3, **ckw)
    grav = gravity_point(1.3, soft=0.05, pos=pos)
    dt, time = 2.0e-4, 0.25
    # oracle: the reference's task order (artemis_driver.cpp:182-255)
    for fluid in (0, 1):
        o.CalculateFluxes(fluid, False)
    o.ZeroDiffusionFlux(), o.ViscousFlux(), o.ThermalFlux()
    o.ApplyUpdate(0.0, 1.0, dt)
    for fluid in (0, 1):
        o.FluxSource(dt, fluid)
    o.DiffusionUpdate(dt)
    o.ExternalGravity(time, dt)
    o.RotatingFrameForce(dt)
    o.CoolingSource(time, dt)
    o.SetAuxillaryFields()
    o.ConsToPrim()
    # product: diffusion fluxes of the input primitives, then one kernel per fluid
    mb.ZeroDiffusionFlux(), mb.ViscousFlux(D), mb.ThermalFlux(D)
    gbuf, gout = mb.new_prim_buffer("o")
    dbuf, dout = mb.new_dust_prim_buffer("o")
    dtd = torch.full((1,), 1.7976931348623157e308, dtype=torch.float64, device="cuda")
    mb.stage_general(0.0, 1.0, dt, dt, gas=(mb.gas_prim_table, mb.gas_prim_table, gout),
                     dust=(mb.dust_prim_table, mb.dust_prim_table, dout), time=time, gravity=grav,
                     rotating_frame=(om, q), cfl=(0.3, 0.3), dt_dev=dtd.data_ptr(), diffusion=D, cooling=cool)
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    keep = [v for v in range(12) if not (8 <= v < 10)]  # P is not written
    assert np.array_equal(gbuf[0][I].cpu().numpy()[keep], o.gprim[I][keep])
    same(dbuf[0][I], o.dprim[I], "dust prim")


CURV_SRC_BLOCKS = [
    ("spherical", (40, 19, 21), (0.9, 1.06, -3.1), (5.6, 2.08, 3.1)),
    ("spherical", (70, 21, 1), (0.6, 1.06, -0.5), (5.6, 2.08, 0.5)),
    ("cylindrical", (35, 10, 18), (0.8, -3.1, -1.0), (4.3, 3.1, 1.0)),
    ("axisymmetric", (45, 12, 1), (0.6, -2.0, -0.5), (4.3, 2.0, 0.5)),
    ("spherical", (77, 1, 1), (0.6, 0.0, -0.5), (5.6, np.pi, 0.5)),
]


@pytest.mark.parametrize("coordinates,nx,lo,hi", CURV_SRC_BLOCKS)
@pytest.mark.parametrize("stage2", [False, True])
def test_curvilinear_tile_kernel_with_sources(hiplib, coordinates, nx, lo, hi, stage2, monkeypatch, option):
    """One gas species on a curvilinear block with everything the curvilinear instantiation of the streaming
    tile kernel folds in: DiffusionUpdate from stored viscous + thermal fluxes, point-mass gravity (off-centre
    where the system allows), RotatingFrameImpl from the cell's own mass fluxes, the frame velocity in
    FluxSource's coordinate sources, and the timestep of the new state -- against the oracle's task chain,
    bit for bit; the cell-centred general stage (ARTEMIS_NO_FUSED_CURV) gives the same bits."""
    from artemis_amd.pack import MeshBlockPack, diffusion_params, gravity_point
    kw = dict(ng=2, ns_gas=1, ns_dust=0, reconstruct="plm", riemann="hlle", dust_reconstruct="plm",
              dust_riemann="hlle", gamma=1.4, dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10,
              coordinates=coordinates)
    o = Oracle(nx, lo, hi, bc=("outflow",) * 6, cfl=0.3, dust_cfl=0.3, **kw)
    random_state(o, np.random.default_rng(91), shock=False, mach=0.5, contrast=10.0)
    om = 0.8
    mb = MeshBlockPack(1, nx, [lo], [hi], with_diffusion=True, omega_frame=om, **kw)
    push([o], mb)
    o.DeepCopyConservedData()
    gin = gu1 = mb.gas_prim_table
    if stage2:
        o2 = Oracle(nx, lo, hi, bc=("outflow",) * 6, cfl=0.3, dust_cfl=0.3, **kw)
        random_state(o2, np.random.default_rng(17), shock=False, mach=0.5, contrast=10.0)
        o.gu1[:] = o2.gu0
        t, gu1 = mb.new_prim_buffer("u1")
        t.copy_(torch.from_numpy(o2.gprim[None]).to(t.device))
    pos = (0.1, 0.05, 0.0) if coordinates == "cylindrical" or nx[2] > 1 else (0.0, 0.0, 0.0)
    o.set_gravity_point(1.3, soft=0.05, x=pos[0], y=pos[1], z=pos[2])
    o.set_rotating_frame(om, 0.0)
    o.set_viscosity("alpha", alpha=2e-2, eta_bulk=0.3, r0=0.9, Omega0=1.2)
    o.set_conductivity("conductivity", cond=0.03, averaging="harmonic")
    D = diffusion_params(1.4, viscosity=dict(type="alpha", alpha=2e-2, eta_bulk=0.3, r0=0.9, Omega0=1.2),
                         conductivity=dict(type="conductivity", cond=0.03, averaging="harmonic"))
    mb.viscosity_radial_table(D)
    grav = gravity_point(1.3, soft=0.05, pos=pos)
    dt, time = 2.0e-4, 0.25
    g0, g1, be = (0.5, 0.5, 0.5) if stage2 else (0.0, 1.0, 1.0)
    o.CalculateFluxes(0, False)
    o.ZeroDiffusionFlux(), o.ViscousFlux(), o.ThermalFlux()
    o.ApplyUpdate(g0, g1, be * dt)
    o.FluxSource(be * dt, 0)
    o.DiffusionUpdate(be * dt)
    o.ExternalGravity(time, be * dt)
    o.RotatingFrameForce(be * dt)
    o.SetAuxillaryFields()
    o.ConsToPrim()
    mb.ZeroDiffusionFlux(), mb.ViscousFlux(D), mb.ThermalFlux(D)
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    keep = [0, 1, 2, 3, 5]
    dts = []
    for nofuse in (False, True):
        if nofuse:
            option("no_fused_curv", 1)
        gbuf, gout = mb.new_prim_buffer("o%d" % nofuse)
        dtd = torch.full((1,), 1.7976931348623157e308, dtype=torch.float64, device="cuda")
        mb.stage_general(g0, g1, be * dt, be * dt, gas=(gin, gu1, gout), time=time, gravity=grav,
                         rotating_frame=(om, 0.0), cfl=(0.3, 0.3), dt_dev=dtd.data_ptr(), diffusion=D)
        assert mb.last_stage_variant == (0 if nofuse else 2)
        assert np.array_equal(gbuf[0][I].cpu().numpy()[keep], o.gprim[I][keep]), nofuse
        dts.append(dtd.item())
    # hydro limit of the new state: the tile kernel reduces it itself, the cell-centred path runs
    # estimate_dt_kernel (checked against the oracle in test_parity_ops); the oracle's number also folds in
    # the diffusive limits (gas.cpp:435-467), which the driver adds with artemis_hip_diffusion_dt
    assert dts[0] == dts[1] and dts[0] >= o.EstimateTimestepMesh(0)


@pytest.mark.parametrize("coordinates,nx,lo,hi", EXTRA_BLOCKS[:2] + EXTRA_BLOCKS[4:5])
def test_general_stage_with_diffusion_and_drag(hiplib, coordinates, nx, lo, hi):
    """DiffusionUpdate inside the stage kernel followed by the drag / SetAuxillaryFields / ConsToPrim pass
    (the stage leaves the post-source conserved state in cons0 when drag couples the fluids)."""
    from artemis_amd.pack import MeshBlockPack, diffusion_params, drag_params
    kw = dict(ng=2, ns_gas=1, ns_dust=2, reconstruct="plm", riemann="hllc", dust_reconstruct="plm",
              dust_riemann="hlle", gamma=1.4, dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10,
              coordinates=coordinates)
    o = Oracle(nx, lo, hi, bc=("outflow",) * 6, cfl=0.3, dust_cfl=0.3, **kw)
    random_state(o, np.random.default_rng(93), shock=False, mach=0.5, contrast=10.0)
    mb = MeshBlockPack(1, nx, [lo], [hi], with_diffusion=True, **kw)
    push([o], mb)
    o.DeepCopyConservedData()
    o.set_viscosity("constant", nu=0.04, eta_bulk=0.5)
    o.set_conductivity("diffusivity", kappa=0.02)
    o.set_drag("simple_dust", "constant", tau=[0.05, 2.0])
    D = diffusion_params(1.4, viscosity=dict(type="constant", nu=0.04, eta_bulk=0.5),
                         conductivity=dict(type="diffusivity", kappa=0.02))
    drag = drag_params("simple_dust", "constant", tau=[0.05, 2.0], mesh_min=lo, mesh_max=hi)
    dt = 2.0e-4
    for fluid in (0, 1):
        o.CalculateFluxes(fluid, False)
    o.ZeroDiffusionFlux(), o.ViscousFlux(), o.ThermalFlux()
    o.ApplyUpdate(0.0, 1.0, dt)
    for fluid in (0, 1):
        o.FluxSource(dt, fluid)
    o.DiffusionUpdate(dt)
    o.DragSource(dt)
    o.SetAuxillaryFields()
    o.ConsToPrim()
    mb.ZeroDiffusionFlux(), mb.ViscousFlux(D), mb.ThermalFlux(D)
    gbuf, gout = mb.new_prim_buffer("o")
    dbuf, dout = mb.new_dust_prim_buffer("o")
    mb.stage_general(0.0, 1.0, dt, dt, gas=(mb.gas_prim_table, mb.gas_prim_table, gout),
                     dust=(mb.dust_prim_table, mb.dust_prim_table, dout), drag=drag, diffusion=D)
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    keep = [0, 1, 2, 3, 5]
    assert np.array_equal(gbuf[0][I].cpu().numpy()[keep], o.gprim[I][keep])
    same(dbuf[0][I], o.dprim[I], "dust prim")


@pytest.mark.parametrize("coordinates,nx,lo,hi", EXTRA_BLOCKS)
@pytest.mark.parametrize("stage2", [False, True])
def test_stage_epilogue_over_stored_fluxes(hiplib, coordinates, nx, lo, hi, stage2):
    """artemis_hip_stage_epilogue: after CalculateFluxes and the diffusion-flux tasks, ApplyUpdate (with a
    distinct cons1 for RK2's second stage), FluxSource, DiffusionUpdate, gravity, rotating frame,
    cooling, SetAuxillaryFields and ConsToPrim in one pass, primitives written in place."""
    from artemis_amd.pack import MeshBlockPack, diffusion_params, gravity_point
    cart = coordinates == "cartesian"
    kw = dict(ng=2, ns_gas=2, ns_dust=1, reconstruct="plm", riemann="hlle", dust_reconstruct="plm",
              dust_riemann="hlle", gamma=1.4, dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10,
              coordinates=coordinates)
    o = Oracle(nx, lo, hi, bc=("outflow",) * 6, cfl=0.3, dust_cfl=0.3, **kw)
    random_state(o, np.random.default_rng(95), shock=False, mach=0.5, contrast=10.0)
    om, q = 0.8, (1.5 if cart else 0.0)
    mb = MeshBlockPack(1, nx, [lo], [hi], with_diffusion=True, omega_frame=om, **kw)
    o.DeepCopyConservedData()
    if stage2:
        o2 = Oracle(nx, lo, hi, bc=("outflow",) * 6, **kw)
        random_state(o2, np.random.default_rng(96), shock=False, mach=0.5, contrast=10.0)
        o.gu1[:] = o2.gu0
        o.du1[:] = o2.du0
    push([o], mb)
    pos = (0.1, 0.05, 0.0) if coordinates in ("cartesian", "cylindrical") or nx[2] > 1 else (0.0, 0.0, 0.0)
    o.set_gravity_point(1.3, soft=0.05, x=pos[0], y=pos[1], z=pos[2])
    o.set_rotating_frame(om, q)
    o.set_viscosity("constant", nu=0.03, eta_bulk=0.3)
    o.set_conductivity("conductivity", cond=0.03)
    ckw = dict(beta0=2.0, beta_min=1e-3, tcyl=0.02, cyl_plaw=-1.0)
    o.set_cooling(**ckw)
    D = diffusion_params(1.4, viscosity=dict(type="constant", nu=0.03, eta_bulk=0.3),
                         conductivity=dict(type="conductivity", cond=0.03))
    cool = mb.cooling_params(1.4, 1.3, **ckw)
    grav = gravity_point(1.3, soft=0.05, pos=pos)
    g0, g1, be = (0.5, 0.5, 0.5) if stage2 else (0.0, 1.0, 1.0)
    dt, time = 2.0e-4, 0.25
    for fluid in (0, 1):
        o.CalculateFluxes(fluid, False)
        mb.CalculateFluxes(fluid, False)
    o.ZeroDiffusionFlux(), o.ViscousFlux(), o.ThermalFlux()
    mb.ZeroDiffusionFlux(), mb.ViscousFlux(D), mb.ThermalFlux(D)
    o.ApplyUpdate(g0, g1, be * dt)
    for fluid in (0, 1):
        o.FluxSource(be * dt, fluid)
    o.DiffusionUpdate(be * dt)
    o.ExternalGravity(time, be * dt)
    o.RotatingFrameForce(be * dt)
    o.CoolingSource(time, be * dt)
    o.SetAuxillaryFields()
    o.ConsToPrim()
    mb.stage_epilogue(g0, g1, be * dt, be * dt, time=time, gravity=grav, rotating_frame=(om, q), diffusion=D, cooling=cool)
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    keep = [v for v in range(12) if not (8 <= v < 10)]  # P is ConsToPrim's business only after PrimToCons
    assert np.array_equal(mb.gas_prim[0][I].cpu().numpy()[keep], o.gprim[I][keep])
    same(mb.dust_prim[0][I], o.dprim[I], "dust prim")


@pytest.mark.parametrize("coordinates,nx,lo,hi", EXTRA_BLOCKS)
@pytest.mark.parametrize("with_drag", [True, False])
def test_stage_epilogue_cons_then_finish(hiplib, coordinates, nx, lo, hi, with_drag):
    """The per-task chain of a deck with drag (or N-body gravity) in two passes: artemis_hip_stage_epilogue_cons leaves
    ApplyUpdate + FluxSource + DiffusionUpdate + ExternalGravity + RotatingFrameForce in cons0 (checked on its own: that
    is where NBodyGravity acts), artemis_hip_stage_finish does DragSource + SetAuxillaryFields + ConsToPrim -- every bit
    of the eight reference tasks run one by one."""
    from artemis_amd.pack import MeshBlockPack, diffusion_params, gravity_point, drag_params
    cart = coordinates == "cartesian"
    kw = dict(ng=2, ns_gas=1, ns_dust=2, reconstruct="plm", riemann="hllc", dust_reconstruct="plm",
              dust_riemann="hlle", gamma=1.4, dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10,
              coordinates=coordinates)
    o = Oracle(nx, lo, hi, bc=("outflow",) * 6, cfl=0.3, dust_cfl=0.3, **kw)
    random_state(o, np.random.default_rng(195), shock=False, mach=0.5, contrast=10.0)
    om, q = 0.8, (1.5 if cart else 0.0)
    mb = MeshBlockPack(1, nx, [lo], [hi], with_diffusion=True, omega_frame=om, **kw)
    o.DeepCopyConservedData()
    o2 = Oracle(nx, lo, hi, bc=("outflow",) * 6, **kw)
    random_state(o2, np.random.default_rng(196), shock=False, mach=0.5, contrast=10.0)
    o.gu1[:] = o2.gu0
    o.du1[:] = o2.du0
    push([o], mb)
    pos = (0.1, 0.05, 0.0) if coordinates in ("cartesian", "cylindrical") or nx[2] > 1 else (0.0, 0.0, 0.0)
    o.set_gravity_point(1.3, soft=0.05, x=pos[0], y=pos[1], z=pos[2])
    o.set_rotating_frame(om, q)
    o.set_viscosity("constant", nu=0.03, eta_bulk=0.3)
    tau = [0.05, 2.0]
    o.set_drag("simple_dust", "constant", tau=tau)
    D = diffusion_params(1.4, viscosity=dict(type="constant", nu=0.03, eta_bulk=0.3))
    drag = drag_params("simple_dust", "constant", tau=tau, mesh_min=lo, mesh_max=hi)
    grav = gravity_point(1.3, soft=0.05, pos=pos)
    g0, g1, be = 0.5, 0.5, 0.5
    dt, time = 2.0e-4, 0.25
    for fluid in (0, 1):
        o.CalculateFluxes(fluid, False)
        mb.CalculateFluxes(fluid, False)
    o.ZeroDiffusionFlux(), o.ViscousFlux()
    mb.ZeroDiffusionFlux(), mb.ViscousFlux(D)
    o.ApplyUpdate(g0, g1, be * dt)
    for fluid in (0, 1):
        o.FluxSource(be * dt, fluid)
    o.DiffusionUpdate(be * dt)
    o.ExternalGravity(time, be * dt)
    o.RotatingFrameForce(be * dt)
    mb.stage_epilogue_cons(g0, g1, be * dt, be * dt, time=time, gravity=grav, rotating_frame=(om, q), diffusion=D)
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    assert np.array_equal(mb.gas_u0[0][I].cpu().numpy(), o.gu0[I])
    assert np.array_equal(mb.dust_u0[0][I].cpu().numpy(), o.du0[I])
    if with_drag:
        o.DragSource(be * dt)
    o.SetAuxillaryFields()
    o.ConsToPrim()
    mb.stage_finish(time, be * dt, drag if with_drag else None)
    keep = [0, 1, 2, 3, 5]  # P is ConsToPrim's business only after PrimToCons
    assert np.array_equal(mb.gas_prim[0][I].cpu().numpy()[keep], o.gprim[I][keep])
    same(mb.dust_prim[0][I], o.dprim[I], "dust prim")


@pytest.mark.parametrize("coordinates,nx,lo,hi", EXTRA_BLOCKS[:1] + EXTRA_BLOCKS[3:5])
def test_refined_mesh_fixup_contract(hiplib, coordinates, nx, lo, hi):
    """The two entry points behind the one-kernel stages on refined meshes, on one block (the multi-block runs are in
    test_multilevel.py):
      * artemis_hip_ml_face_fluxes over a slab of faces stores exactly what CalculateFluxes stores there;
      * artemis_hip_ml_stage_fixup redoes the listed zones with the flagged faces' mass / momentum / energy / pressure
        fluxes taken from the flux arrays -- here altered by hand on the upper x1 boundary face and the lower x2 one, the
        way SetFluxCorrections alters them -- and equals the reference chain (ApplyUpdate ... ConsToPrim) run on the altered
        arrays bit for bit, face velocity untouched; zones not listed keep the stage kernel's result."""
    from artemis_amd.pack import MeshBlockPack, diffusion_params, gravity_point
    kw = dict(ng=2, ns_gas=1, ns_dust=1, reconstruct="plm", riemann="hllc", dust_reconstruct="plm",
              dust_riemann="hlle", gamma=1.4, dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10,
              coordinates=coordinates)
    o = Oracle(nx, lo, hi, bc=("outflow",) * 6, cfl=0.3, dust_cfl=0.3, **kw)
    random_state(o, np.random.default_rng(295), shock=False, mach=0.5, contrast=10.0)
    mb = MeshBlockPack(1, nx, [lo], [hi], with_diffusion=True, **kw)
    o.DeepCopyConservedData()
    push([o], mb)
    ndim = 3 if nx[2] > 1 else (2 if nx[1] > 1 else 1)
    o.set_viscosity("constant", nu=0.03, eta_bulk=0.3)
    D = diffusion_params(1.4, viscosity=dict(type="constant", nu=0.03, eta_bulk=0.3))
    dt = 2.0e-4
    # per-task fluxes on both sides (the arrays the fix-up reads its flagged faces from)
    for fluid in (0, 1):
        o.CalculateFluxes(fluid, False)
        mb.CalculateFluxes(fluid, False)
    want = [mb.gas_flux[d].clone() for d in range(ndim)], [mb.gas_pflux[d].clone() for d in range(ndim)], \
           [mb.gas_vface[d].clone() for d in range(ndim)], [mb.dust_flux[d].clone() for d in range(ndim)]
    for d in range(ndim):  # wipe, then let ml_face_fluxes restore two slabs
        mb.gas_flux[d].fill_(7.0), mb.gas_pflux[d].fill_(7.0), mb.gas_vface[d].fill_(7.0), mb.dust_flux[d].fill_(7.0)
    o.ZeroDiffusionFlux(), o.ViscousFlux()
    mb.ZeroDiffusionFlux(), mb.ViscousFlux(D)
    gbuf, gout = mb.new_prim_buffer("o")
    dbuf, dout = mb.new_dust_prim_buffer("o")
    mb.stage_general(0.0, 1.0, dt, dt, gas=(mb.gas_prim_table, mb.gas_prim_table, gout),
                     dust=(mb.dust_prim_table, mb.dust_prim_table, dout), diffusion=D)
    stage_g, stage_d = gbuf.clone(), dbuf.clone()
    n = [o.ie - o.is_ + 1, o.je - o.js + 1, o.ke - o.ks + 1]
    boxes = [(0, 0, (o.ie + 1, o.js, o.ks), (1, n[1], n[2]))]
    if ndim > 1:
        boxes.append((0, 1, (o.is_, o.js, o.ks), (n[0], 1, n[2])))
    mb.ml_face_fluxes(boxes)
    sl = [(slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.ie + 1, o.ie + 2)),
          (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.js + 1), slice(o.is_, o.ie + 1))]
    for (b, d, _, _), I in zip(boxes, sl):
        for got, ref in zip((mb.gas_flux[d], mb.gas_pflux[d], mb.gas_vface[d], mb.dust_flux[d]),
                            (want[0][d], want[1][d], want[2][d], want[3][d])):
            assert torch.equal(got[0][I], ref[0][I]), ("face fluxes", d)
    # SetFluxCorrections by hand: other numbers in the flux fields of those two faces, on both sides
    rng = np.random.default_rng(7)
    for (b, d, _, _), I in zip(boxes, sl):
        for arr_o, arr_m in ((o.gflux(d), mb.gas_flux[d]), (o.gpflux(d), mb.gas_pflux[d]), (o.dflux(d), mb.dust_flux[d])):
            f = 1.0 + 0.05 * rng.standard_normal(arr_o[I].shape)
            arr_o[I] = arr_o[I] * f
            arr_m[0][I] = torch.from_numpy(np.ascontiguousarray(arr_o[I])).to(arr_m.device)
    o.ApplyUpdate(0.0, 1.0, dt)
    for fluid in (0, 1):
        o.FluxSource(dt, fluid)
    o.DiffusionUpdate(dt)
    o.SetAuxillaryFields()
    o.ConsToPrim()
    cells = {}
    for k in range(o.ks, o.ke + 1):
        for j in range(o.js, o.je + 1):
            cells[(k, j, o.ie)] = cells.get((k, j, o.ie), 0) | 2      # upper x1 face
        if ndim > 1:
            for i in range(o.is_, o.ie + 1):
                cells[(k, o.js, i)] = cells.get((k, o.js, i), 0) | 4  # lower x2 face
    mb.ml_stage_fixup([(0, k, j, i, f) for (k, j, i), f in cells.items()])
    keep = [0, 1, 2, 3, 5]
    g, dd = gbuf[0].cpu().numpy(), dbuf[0].cpu().numpy()
    listed = np.zeros(g.shape[1:], dtype=bool)
    for (k, j, i) in cells:
        listed[k, j, i] = True
    inter = np.zeros_like(listed)
    inter[o.ks:o.ke + 1, o.js:o.je + 1, o.is_:o.ie + 1] = True
    assert listed.sum() > 0 and np.array_equal(g[keep][:, listed], o.gprim[keep][:, listed])
    assert np.array_equal(dd[:, listed], o.dprim[:, listed])
    rest = inter & ~listed
    assert np.array_equal(g[keep][:, rest], stage_g[0].cpu().numpy()[keep][:, rest])
    assert np.array_equal(dd[:, rest], stage_d[0].cpu().numpy()[:, rest])
    # and the altered fluxes did matter: the stage kernel's own result differs at the listed zones
    assert not np.array_equal(stage_g[0].cpu().numpy()[0][listed], g[0][listed])


NBODY_BLOCKS = [
    ("cartesian", (24, 12, 10), (-1.0, -0.5, -0.35), (1.0, 0.8, 0.35), 0),
    ("cylindrical", (35, 18, 20), (0.5, -3.1, -0.6), (2.3, 3.1, 0.6), 3),   # the tile march + dust cell kernel + drag finish
    ("cylindrical", (16, 16, 1), (0.5, -3.1, -0.5), (2.3, 3.1, 0.5), 3),    # ... 2-D
    ("spherical", (34, 12, 20), (0.5, 1.1, -3.1), (2.3, 2.04, 3.1), 3),
]


@pytest.mark.parametrize("coordinates,nx,lo,hi,variant", NBODY_BLOCKS)
@pytest.mark.parametrize("mode", ["drag", "drag_launch", "defer", "finish2", "defer_cells", "gas_only"])
def test_general_stage_with_nbody_gravity_and_drag(hiplib, coordinates, nx, lo, hi, variant, mode, option):
    """Gravity::NBodyGravity inside the one-kernel stage (artemis_stage_general_args_t.nbody_dev): a spline-softened
    accreting sink, a Plummer particle with a momentum sink and an uncoupled one, with the frame correction of a
    rotating frame, followed by the rotating-frame task and simple_dust drag -- against the oracle's task chain
    (artemis_driver.cpp:182-255), bit for bit; the seven sums per particle (artemis_hip_nbody_force_sums) to 1e-12.
    mode = drag: DragSource + SetAuxillaryFields + ConsToPrim inside the call; defer: the call stops at the conserved
    state (defer_finish, what a refined mesh does around its fix-up) and artemis_hip_stage_finish completes it;
    gas_only: no dust, no drag (the march stores primitives itself).  Round 6: `drag` on the curvilinear marches couples
    the fluids INSIDE the dust march (simple_drag1_finish on its registers) -- drag_launch is the same call with the finish
    as its own launch (NO_DRAG_IN_MARCH); finish2 = defer_finish 2 (the stage finishes every zone itself, what a refined
    mesh asks for since round 6); defer_cells = the stage stops at the conserved state and artemis_hip_stage_finish_cells
    finishes a LISTED third of the zones into NaN-filled primitive tables: listed zones equal the oracle, no other zone
    is touched."""
    from artemis_amd.pack import MeshBlockPack, drag_params
    if mode == "drag_launch":
        option("no_drag_in_march")
    cart = coordinates == "cartesian"
    nsd = 0 if mode == "gas_only" else 1
    kw = dict(ng=2, ns_gas=1, ns_dust=nsd, reconstruct="plm", riemann="hllc", dust_reconstruct="plm", dust_riemann="hlle",
              gamma=1.4, dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10, coordinates=coordinates)
    o = Oracle(nx, lo, hi, bc=("outflow",) * 6, cfl=0.3, dust_cfl=0.3, **kw)
    random_state(o, np.random.default_rng(17), shock=False, mach=0.5, contrast=10.0)
    om = 0.8
    mb = MeshBlockPack(1, nx, [lo], [hi], with_fluxes=False, omega_frame=om, **kw)
    push([o], mb)
    o.DeepCopyConservedData()
    parts = [dict(GM=1.3, pos=(1.1, -0.1, 0.05), vel=(0.1, 0.4, -0.2), rs=0.3, racc=0.9, gamma=4.0, beta=0.0, spline=1),
             dict(GM=0.4, pos=(-0.9, 0.75, 0.1), vel=(0.0, -0.3, 0.1), xf=(0.01, 0.02, 0.0), vf=(0.0, 0.1, 0.0), rs=0.1,
                  racc=0.8, gamma=2.0, beta=6.0, spline=0),
             dict(GM=5.0, pos=(0.0, 0.0, 0.0), couple=0)]
    o.set_rotating_frame(om, 1.5 if cart else 0.0)
    o.set_gravity_nbody(parts, frame_correction=True)
    tau = [0.05]
    if nsd:
        o.set_drag("simple_dust", "constant", tau=tau)
    dt, time = 2.0e-3, 0.1
    for fluid in (0, 1):
        o.CalculateFluxes(fluid, False)
    o.ApplyUpdate(0.0, 1.0, dt)
    for fluid in (0, 1):
        o.FluxSource(dt, fluid)
    o.ExternalGravity(time, dt)
    want_force = o.nbody_force(reset=True)
    o.RotatingFrameForce(dt)
    if nsd:
        o.DragSource(dt)
    o.SetAuxillaryFields()
    o.ConsToPrim()
    nb = mb.nbody_device(parts)
    got_force = mb.nbody_force_sums(nb, om, dt)
    scale = np.abs(want_force).max(axis=1, keepdims=True) + 1e-300
    assert np.all(got_force[2] == 0.0) and np.max(np.abs(got_force - want_force) / scale) < 1e-12, (got_force, want_force)
    drag = drag_params("simple_dust", "constant", tau=tau, mesh_min=lo, mesh_max=hi) if nsd else None
    gbuf, gout = mb.new_prim_buffer("o")
    dbuf, dout = mb.new_dust_prim_buffer("o") if nsd else (None, None)
    mb.stage_general(0.0, 1.0, dt, dt, gas=(mb.gas_prim_table, mb.gas_prim_table, gout),
                     dust=(mb.dust_prim_table, mb.dust_prim_table, dout) if nsd else (None, None, None), time=time,
                     rotating_frame=(om, 1.5 if cart else 0.0), drag=drag, nbody=nb, nbody_omf=om,
                     defer_finish={"defer": 1, "defer_cells": 1, "finish2": 2}.get(mode, 0))
    assert mb.last_stage_variant == variant
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    keep = [0, 1, 2, 3, 5]
    if mode == "defer_cells":  # the listed zones only, into poisoned tables
        mb.gas_prim.fill_(float("nan")), mb.dust_prim.fill_(float("nan"))
        zones = [(0, k, j, i, 0) for k in range(o.ks, o.ke + 1) for j in range(o.js, o.je + 1) for i in range(o.is_, o.ie + 1)][::3]
        mb.stage_finish_cells(time, dt, drag, zones)
        got_g, got_d = mb.gas_prim[0].cpu().numpy(), mb.dust_prim[0].cpu().numpy()
        listed = np.zeros(got_g.shape[1:], dtype=bool)
        for _, k, j, i, _f in zones:
            listed[k, j, i] = True
        assert np.array_equal(got_g[keep][:, listed], o.gprim[keep][:, listed])
        assert np.array_equal(got_d[:, listed], o.dprim[:, listed])
        assert np.isnan(got_g[keep][:, ~listed]).all() and np.isnan(got_d[:, ~listed]).all()
        return
    if mode == "defer":  # the finish works on the pack's own primitive tables
        mb.stage_finish(time, dt, drag)
        got_g, got_d = mb.gas_prim[0][I].cpu().numpy(), mb.dust_prim[0][I]
    else:
        got_g, got_d = gbuf[0][I].cpu().numpy(), (dbuf[0][I] if nsd else None)
    bad = got_g[keep] != o.gprim[I][keep]
    assert not bad.any(), (np.count_nonzero(bad), np.argwhere(bad)[:5])
    if nsd:
        same(got_d, o.dprim[I], "dust prim")
