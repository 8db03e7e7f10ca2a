"""GPU parity of the gas diffusion tasks (ZeroDiffusionFlux, ViscousFlux, ThermalFlux,
DiffusionUpdate, diffusive timestep; reference gas.cpp:522-641, utils/diffusion/*.hpp) against
the CPU oracle, BIT-EXACT, Cartesian 1-D / 2-D / 3-D and every curvilinear system (scale
factors, connection coefficients, Coords::Distance through ConvertToCart), several species, both
face averagings."""
import numpy as np
import pytest
import torch

from oracle.oracle import Oracle
from test_parity_ops import face_slices, push, random_state, same

pytestmark = pytest.mark.gpu


def pair(nx, ns_gas=1, seed=0, ng=2, coordinates="cartesian", lo=(-1.0, -0.5, 0.25), hi=(1.0, 0.8, 0.95)):
    from artemis_amd.pack import MeshBlockPack
    kw = dict(ng=ng, ns_gas=ns_gas, ns_dust=0, reconstruct="plm", riemann="hlle", gamma=1.4,
              dfloor=1e-10, siefloor=1e-10, coordinates=coordinates)
    o = Oracle(nx, lo, hi, bc=("outflow",) * 6, cfl=0.3, **kw)
    random_state(o, np.random.default_rng(seed), shock=False, mach=0.5, contrast=10.0)
    mb = MeshBlockPack(1, nx, [lo], [hi], with_diffusion=True, **kw)
    push([o], mb)
    return o, mb


# curvilinear blocks: name, nx, lower and upper corner (spherical x2 ghosts stay inside (0, pi),
# see test_parity_geometry.py)
CURVI = [
    ("spherical", (40, 1, 1), (0.0, 0.0, -0.5), (1.0, np.pi, 0.5)),
    ("spherical", (24, 12, 1), (0.4, 0.5, -0.5), (2.5, 2.6, 0.5)),
    ("spherical", (14, 10, 8), (0.3, 0.7, 0.0), (1.7, 2.5, 2 * np.pi)),
    ("cylindrical", (16, 12, 6), (0.5, 0.0, -1.0), (2.0, 2 * np.pi, 1.0)),
    ("cylindrical", (33, 1, 1), (0.0, -0.5, -0.5), (1.0, 0.5, 0.5)),
    ("axisymmetric", (24, 12, 1), (0.0, -1.0, -0.5), (2.0, 1.0, 0.5)),
    ("axisymmetric", (12, 8, 6), (0.7, -1.0, 0.0), (2.0, 1.0, 1.0)),
]
CART = [("cartesian", nx, (-1.0, -0.5, 0.25), (1.0, 0.8, 0.95))
        for nx in [(24, 12, 10), (33, 9, 1), (70, 1, 1), (5, 4, 3)]]


@pytest.mark.parametrize("coordinates,nx,lo,hi", CART + CURVI)
@pytest.mark.parametrize("avg", ["arithmetic", "harmonic"])
@pytest.mark.parametrize("ctype,table", [("conductivity", False), ("diffusivity", False), ("conductivity", True)])
def test_diffusion_tasks(hiplib, coordinates, nx, lo, hi, avg, ctype, table):
    from artemis_amd.pack import diffusion_params
    o, mb = pair(nx, ns_gas=2, seed=51, coordinates=coordinates, lo=lo, hi=hi)
    o.set_viscosity("constant", nu=0.03, eta_bulk=0.4, averaging=avg)
    ck = dict(cond=0.07) if ctype == "conductivity" else dict(kappa=0.07)
    o.set_conductivity(ctype, averaging=avg, **ck)
    D = diffusion_params(1.4, viscosity=dict(type="constant", nu=0.03, eta_bulk=0.4, averaging=avg),
                         conductivity=dict(type=ctype, averaging=avg, **ck))
    if table:  # the static Coords::Distance table (artemis_hip_viscous_distance_fill): same bits as on the fly
        mb.distance_table(D)
    o.ZeroDiffusionFlux(), mb.ZeroDiffusionFlux()
    o.ViscousFlux(), mb.ViscousFlux(D)
    for d in range(o.ndim):
        same(mb.gas_diff_flux[d][0][face_slices(o, d)], o.qflux(d)[face_slices(o, d)], f"viscous flux x{d+1}")
    # the fused pair (ZeroDiffusionFlux + ViscousFlux in one pass) overwrites whatever the arrays held
    for d in range(o.ndim):
        mb.gas_diff_flux[d].fill_(7.25)
    mb.ZeroViscousFlux(D)
    for d in range(o.ndim):
        same(mb.gas_diff_flux[d][0][face_slices(o, d)], o.qflux(d)[face_slices(o, d)], f"zero + viscous flux x{d+1}")
    o.ThermalFlux(), mb.ThermalFlux(D)
    for d in range(o.ndim):
        same(mb.gas_diff_flux[d][0][face_slices(o, d)], o.qflux(d)[face_slices(o, d)], f"visc+thermal flux x{d+1}")
    o.DiffusionUpdate(2.0e-4), mb.DiffusionUpdate(D, 2.0e-4)
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    same(mb.gas_u0[0][I], o.gu0[I], "DiffusionUpdate")
    # Gas::EstimateTimestepMesh = cfl * min(hydro, viscous, conductive) (gas.cpp:435-467)
    hyd = mb.EstimateTimestepMesh(0, cfl=0.3)
    assert min(hyd, mb.DiffusionTimestep(D, 0.3)) == o.EstimateTimestepMesh(0)
    # ... and every limit of both fluids in one pass (artemis_hip_timestep_all)
    want = o.EstimateTimestepMesh(0)
    if o.cfg.ns_dust:
        want = min(want, o.EstimateTimestepMesh(1))
    assert mb.TimestepAll(0.3, o.cfg.cfl_dust, D) == want
    assert mb.TimestepAll(0.3, o.cfg.cfl_dust, None) == min(
        hyd, o.EstimateTimestepMesh(1) if o.cfg.ns_dust else np.inf)


@pytest.mark.parametrize("coordinates,nx,lo,hi", [CART[0], CURVI[2], CURVI[3]])
@pytest.mark.parametrize("table", [False, True])
def test_viscous_flux_with_vanishing_velocities(hiplib, coordinates, nx, lo, hi, table):
    """The face kernels divide through shared reciprocals, which is the bits of an IEEE division only while the
    numerators are zero or not tiny; a wave that sees a tiny one takes the plain divisions.  Velocities of
    1e-300 (ahead of a shock they decay like that), exact zeros and ordinary values side by side."""
    from artemis_amd.pack import diffusion_params
    o, mb = pair(nx, ns_gas=1, seed=77, coordinates=coordinates, lo=lo, hi=hi)
    rng = np.random.default_rng(5)
    w = o.gprim  # [nvar, nk, nj, ni]: density, three velocities, ...
    scale = rng.choice([1.0, 0.0, 1e-300, 1e-306, 1e-250, 1e-160, 1e-40], size=w[1].shape,
                       p=[0.35, 0.15, 0.1, 0.1, 0.1, 0.1, 0.1])
    for v in (1, 2, 3):
        w[v] *= scale
    o.PrimToCons()
    push([o], mb)
    o.set_viscosity("constant", nu=0.03, eta_bulk=0.4, averaging="harmonic")
    D = diffusion_params(1.4, viscosity=dict(type="constant", nu=0.03, eta_bulk=0.4, averaging="harmonic"))
    if table:
        mb.distance_table(D)
    o.ZeroDiffusionFlux(), o.ViscousFlux()
    mb.ZeroViscousFlux(D)
    for d in range(o.ndim):
        same(mb.gas_diff_flux[d][0][face_slices(o, d)], o.qflux(d)[face_slices(o, d)], f"viscous flux x{d+1}")


def test_conduction_only_update_and_contract(hiplib):
    from artemis_amd import capi
    from artemis_amd.pack import MeshBlockPack, diffusion_params
    o, mb = pair((20, 10, 6), seed=52)
    o.set_conductivity("conductivity", cond=0.2)
    D = diffusion_params(1.4, conductivity=dict(type="conductivity", cond=0.2))
    o.ZeroDiffusionFlux(), mb.ZeroDiffusionFlux()
    o.ViscousFlux(), mb.ViscousFlux(D)  # no-ops: viscosity is off
    o.ThermalFlux(), mb.ThermalFlux(D)
    o.DiffusionUpdate(1.0e-3), mb.DiffusionUpdate(D, 1.0e-3)
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    same(mb.gas_u0[0][I], o.gu0[I], "DiffusionUpdate, conduction only")
    with pytest.raises(capi.ArtemisHipError) as e:  # radial laws need their host-filled table
        mb.ViscousFlux(diffusion_params(1.4, viscosity=dict(type="powerlaw", nu=0.1, r_exp=-0.5)))
    assert e.value.code == capi.EINVAL and "radial" in str(e.value)
    with pytest.raises(capi.ArtemisHipError) as e:
        mb.ViscousFlux(diffusion_params(1.4, viscosity=dict(type="alpha", alpha=0.01)))
    assert e.value.code == capi.EINVAL
    with pytest.raises(capi.ArtemisHipError) as e:  # state power laws need positive reference values
        mb.ThermalFlux(diffusion_params(1.4, conductivity=dict(type="conductivity", cond=0.1, rho_exp=1.0, rho_ref=0.0)))
    assert e.value.code == capi.EINVAL
    cyl = MeshBlockPack(1, (8, 4, 1), [(0.5, 0.0, -0.5)], [(1.0, 3.0, 0.5)], coordinates="cylindrical",
                        with_diffusion=True)
    cyl.pack.metric = None  # Coords::Distance needs the azimuth's cos / sin
    with pytest.raises(capi.ArtemisHipError) as e:
        cyl.ThermalFlux(D)
    assert e.value.code == capi.EINVAL
    nofl = MeshBlockPack(1, (8, 8, 1), [(0, 0, 0)], [(1, 1, 1)])
    with pytest.raises(capi.ArtemisHipError) as e:
        nofl.ThermalFlux(D)
    assert e.value.code == capi.EINVAL
    torch.cuda.synchronize()


@pytest.mark.parametrize("coordinates,nx,lo,hi", CART[:2] + CURVI[2:4])
@pytest.mark.parametrize("ctype", ["conductivity", "diffusivity"])
def test_state_power_law_conductivity(hiplib, coordinates, nx, lo, hi, ctype):
    """K = K0 (T/T_ref)^a (rho/rho_ref)^b (diffusion_coeff.hpp:312-316, :353-359) with a = 2.5 (Spitzer),
    b = -0.5: std::pow of the state per cell in the reference, pow() on the device here -- fluxes,
    update and timestep agree to 1e-13 of their maxima instead of bitwise."""
    from artemis_amd.pack import diffusion_params
    o, mb = pair(nx, ns_gas=2, seed=57, coordinates=coordinates, lo=lo, hi=hi)
    ck = dict(cond=0.07) if ctype == "conductivity" else dict(kappa=0.07)
    law = dict(temp_exp=2.5, rho_exp=-0.5, rho_ref=0.7, T_ref=1.3)
    o.set_conductivity(ctype, averaging="harmonic", **ck, **law)
    D = diffusion_params(1.4, conductivity=dict(type=ctype, averaging="harmonic", **ck, **law))
    o.ZeroDiffusionFlux(), mb.ZeroDiffusionFlux()
    o.ThermalFlux(), mb.ThermalFlux(D)
    for d in range(o.ndim):
        a, b = mb.gas_diff_flux[d][0][face_slices(o, d)].cpu().numpy(), o.qflux(d)[face_slices(o, d)]
        assert np.max(np.abs(a - b)) < 1e-13 * np.abs(b).max(), d
    before = o.gu0.copy()
    o.DiffusionUpdate(2.0e-4), mb.DiffusionUpdate(D, 2.0e-4)
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    a, b = mb.gas_u0[0][I].cpu().numpy(), o.gu0[I]
    assert np.max(np.abs(a - b)) < 1e-13 * np.abs(b - before[I]).max() + 1e-15 * np.abs(b).max()
    t = mb.DiffusionTimestep(D, 0.3)
    ref = o.EstimateTimestepMesh(0)
    hyd = mb.EstimateTimestepMesh(0, cfl=0.3)
    assert abs(min(hyd, t) - ref) < 1e-13 * ref


# ---- the viscous source march (artemis_hip_viscous_source): ZeroDiffusionFlux + ViscousFlux + DiffusionUpdate's sums ----
VSRC = [
    ("cartesian", (40, 19, 21), (-1.0, -0.5, 0.25), (1.0, 0.8, 0.95)),       # ragged 32 x 8 tiles, ragged chunk
    ("cartesian", (16, 16, 8), (-1.0, -0.5, 0.25), (1.0, 0.8, 0.95)),        # the 16 x 16 tile (refined-mesh blocks)
    ("spherical", (40, 19, 21), (0.9, 1.06, -3.1), (5.6, 2.08, 3.1)),
    ("spherical", (64, 16, 12), (0.3, 0.7, 0.0), (1.7, 2.5, 2 * np.pi)),
    ("spherical", (16, 16, 16), (0.4, 0.9, 0.0), (0.9, 2.2, 1.5)),
    ("cylindrical", (35, 10, 18), (0.8, -3.1, -1.0), (4.3, 3.1, 1.0)),
    ("cylindrical", (16, 32, 9), (0.5, 0.0, -1.0), (2.0, 2 * np.pi, 1.0)),
    ("axisymmetric", (24, 12, 10), (0.7, -1.0, 0.0), (2.0, 1.0, 1.0)),
]


@pytest.mark.parametrize("coordinates,nx,lo,hi", VSRC)
@pytest.mark.parametrize("law,tiny", [("alpha", False), ("constant", False), ("constant", True), ("powerlaw", False)])
def test_viscous_source_march(hiplib, coordinates, nx, lo, hi, law, tiny, monkeypatch, option):
    """artemis_hip_viscous_source == ZeroDiffusionFlux -> ViscousFlux -> DiffusionUpdate of the oracle: conserved state
    minus the five sums equals the oracle's updated state bit for bit on every active zone -- three viscosity laws, both
    face averages, velocities of 1e-300 next to zeros and ordinary values
    (`tiny`: the waves that see one take the plain divisions), short chunks (three priming planes per chunk)."""
    from artemis_amd.pack import diffusion_params
    o, mb = pair(nx, ns_gas=1, seed=63, coordinates=coordinates, lo=lo, hi=hi)
    if tiny:
        rng = np.random.default_rng(5)
        scale = rng.choice([1.0, 0.0, 1e-300, 1e-306, 1e-250, 1e-160, 1e-40], size=o.gprim[1].shape,
                           p=[0.35, 0.15, 0.1, 0.1, 0.1, 0.1, 0.1])
        for v in (1, 2, 3):
            o.gprim[v] *= scale
        o.PrimToCons()
        push([o], mb)
    avg = "harmonic" if law == "constant" else "arithmetic"
    if law == "alpha":
        visc = dict(type="alpha", alpha=2e-2, eta_bulk=0.3, r0=0.9, Omega0=1.2, averaging=avg)
        o.set_viscosity("alpha", alpha=2e-2, eta_bulk=0.3, r0=0.9, Omega0=1.2, averaging=avg)
    elif law == "powerlaw":
        visc = dict(type="powerlaw", nu=0.05, r_exp=-0.5, r0=0.8, eta_bulk=1.5, averaging=avg)
        o.set_viscosity("powerlaw", nu=0.05, r_exp=-0.5, r0=0.8, eta_bulk=1.5, averaging=avg)
    else:
        visc = dict(type="constant", nu=0.03, eta_bulk=0.4, averaging=avg)
        o.set_viscosity("constant", nu=0.03, eta_bulk=0.4, averaging=avg)
    D = diffusion_params(1.4, viscosity=visc)
    if law != "constant":
        mb.viscosity_radial_table(D)
    assert mb.viscous_source_covers()
    dt = 2.0e-4
    o.ZeroDiffusionFlux(), o.ViscousFlux()
    before = o.gu0.copy()
    o.DiffusionUpdate(dt)
    I = (slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    mb.distance_table(D)  # (required by the march: it reads the table a plane ahead)
    for kchunk in (None, "5"):
        if kchunk:
            option("visc_kchunk", int(kchunk))
        sums, _ = mb.viscous_source(D, dt)
        got = sums[0].cpu().numpy()
        for q in range(5):
            assert np.array_equal(before[1 + q][I] - got[q][I], o.gu0[1 + q][I]), (kchunk, q)


@pytest.mark.parametrize("coordinates,nx,lo,hi", [VSRC[2], VSRC[5], VSRC[0]])
@pytest.mark.parametrize("stage2", [False, True])
def test_stage_general_with_viscous_sums(hiplib, coordinates, nx, lo, hi, stage2, monkeypatch, option):
    """The stage kernels fed with the sums instead of the twelve flux arrays (artemis_stage_general_args_t.diffusion_sums):
    the streaming tile kernel (curvilinear blocks, and Cartesian ones through its SYS = cartesian instantiation) and the
    cell-centred kernel, against the oracle's task chain."""
    from artemis_amd.pack import MeshBlockPack, diffusion_params, gravity_point
    kw = dict(ng=2, ns_gas=1, ns_dust=0, reconstruct="plm", riemann="hlle", dust_reconstruct="plm", dust_riemann="hlle",
              gamma=1.4, dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10, coordinates=coordinates)
    o = Oracle(nx, lo, hi, bc=("outflow",) * 6, cfl=0.3, dust_cfl=0.3, **kw)
    random_state(o, np.random.default_rng(91), shock=False, mach=0.5, contrast=10.0)
    om = 0.8 if coordinates != "cartesian" else 0.0
    mb = MeshBlockPack(1, nx, [lo], [hi], with_diffusion=False, omega_frame=om, **kw)
    push([o], mb)
    o.DeepCopyConservedData()
    gin = gu1 = mb.gas_prim_table
    if stage2:
        o2 = Oracle(nx, lo, hi, bc=("outflow",) * 6, cfl=0.3, dust_cfl=0.3, **kw)
        random_state(o2, np.random.default_rng(17), shock=False, mach=0.5, contrast=10.0)
        o.gu1[:] = o2.gu0
        t, gu1 = mb.new_prim_buffer("u1")
        t.copy_(torch.from_numpy(o2.gprim[None]).to(t.device))
    o.set_gravity_point(1.3, soft=0.05, x=0.1, y=0.05, z=0.0)
    grav = gravity_point(1.3, soft=0.05, pos=(0.1, 0.05, 0.0))
    if om:
        o.set_rotating_frame(om, 0.0)
    o.set_viscosity("alpha", alpha=2e-2, eta_bulk=0.3, r0=0.9, Omega0=1.2)
    D = diffusion_params(1.4, viscosity=dict(type="alpha", alpha=2e-2, eta_bulk=0.3, r0=0.9, Omega0=1.2))
    mb.viscosity_radial_table(D)
    mb.distance_table(D)
    dt, time = 2.0e-4, 0.25
    g0, g1, be = (0.5, 0.5, 0.5) if stage2 else (0.0, 1.0, 1.0)
    o.CalculateFluxes(0, False)
    o.ZeroDiffusionFlux(), o.ViscousFlux()
    o.ApplyUpdate(g0, g1, be * dt)
    o.FluxSource(be * dt, 0)
    o.DiffusionUpdate(be * dt)
    o.ExternalGravity(time, be * dt)
    if om:
        o.RotatingFrameForce(be * dt)
    o.SetAuxillaryFields()
    o.ConsToPrim()
    _, sums = mb.viscous_source(D, be * dt)
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    keep = [0, 1, 2, 3, 5]
    for nofuse in (False, True):
        if nofuse:
            option("no_fused_curv", 1)
        gbuf, gout = mb.new_prim_buffer("o%d" % nofuse)
        mb.stage_general(g0, g1, be * dt, be * dt, gas=(gin, gu1, gout), time=time, gravity=grav,
                         rotating_frame=(om, 0.0), cfl=(0.3, 0.3), diffusion=D, diffusion_sums=sums)
        assert mb.last_stage_variant == (0 if nofuse else 3)  # (Cartesian packs with sources run the same tile march since round 6)
        assert np.array_equal(gbuf[0][I].cpu().numpy()[keep], o.gprim[I][keep]), nofuse
