"""GPU parity of what the reference's `disk` problem adds on top of the hydro path (SURVEY section 8(f)
rank 2): alpha / radial power-law viscosity (diffusion_coeff.hpp:190-268), the `ic` and `extrap`
user conditions of pgen/disk.hpp, against the CPU oracle.  Everything that only takes
transcendentals of the cell POSITION is bit-exact (host-filled tables); the `extrap` condition
takes log / exp of the STATE on the device and is compared to rounding (tolerance in the test).
Point-mass gravity and the curvilinear rotating frame of the disk decks are covered in
test_parity_sources.py."""
import numpy as np
import pytest
import torch

from oracle.oracle import Oracle
from test_parity_ops import face_slices, push, random_state, same

pytestmark = pytest.mark.gpu

PI = 3.141592653589793
# the layouts of inputs/disk/disk_{sph,cyl,axi}.in at reduced resolution + a Cartesian block; the
# inner radius is moved out so that the two ghost layers of these coarse meshes stay at r > 0 (the
# decks' 128 radial zones keep them positive with x1min = 0.2 / 0.3)
BLOCKS = [
    ("spherical", (16, 8, 8), (0.9, 1.059856161608513, -PI), (5.6, 2.081736491981280, PI)),
    ("spherical", (24, 10, 1), (0.6, 1.06, -0.5), (5.6, 2.08, 0.5)),
    ("spherical", (32, 1, 1), (0.5, 0.0, -0.5), (5.6, PI, 0.5)),
    ("cylindrical", (16, 8, 6), (0.8, -PI, -1.0), (4.3, PI, 1.0)),
    ("axisymmetric", (24, 12, 1), (0.6, -2.0, -0.5), (4.3, 2.0, 0.5)),
    ("cartesian", (12, 10, 8), (0.4, -1.0, -0.6), (2.4, 1.3, 0.7)),
]


def pair(coordinates, nx, lo, hi, ns_gas=1, ns_dust=0, seed=0, bc=("outflow",) * 6, **kw):
    from artemis_amd.pack import MeshBlockPack
    base = dict(ng=2, ns_gas=ns_gas, ns_dust=ns_dust, reconstruct="plm", riemann="hlle", dust_reconstruct="plm",
                dust_riemann="hlle", gamma=1.4, dfloor=1e-10, siefloor=1e-30, dust_dfloor=1e-30,
                coordinates=coordinates)  # floors below the disk profile: PrimToCons leaves the IC as generated
    o = Oracle(nx, lo, hi, bc=bc, cfl=0.3, **base)
    mb = MeshBlockPack(1, nx, [lo], [hi], with_diffusion=True, **base, **kw)
    return o, mb


@pytest.mark.parametrize("coordinates,nx,lo,hi", BLOCKS)
@pytest.mark.parametrize("law", ["alpha", "powerlaw"])
def test_radial_viscosity_laws(hiplib, coordinates, nx, lo, hi, law):
    """nu = alpha c_s^2 / Omega_K(r) and nu = nu0 (R/r0)^r_exp: the pow() of the cell position comes
    from artemis_hip_diffusion_radial_fill (host libm), fluxes / update / dt are bit-exact."""
    from artemis_amd.pack import diffusion_params
    o, mb = pair(coordinates, nx, lo, hi, ns_gas=2)
    random_state(o, np.random.default_rng(61), shock=False, mach=0.5, contrast=10.0)
    push([o], mb)
    if law == "alpha":
        om0 = float(np.sqrt(1.3 / 0.9 ** 3))
        o.set_viscosity("alpha", alpha=2e-2, eta_bulk=0.2, r0=0.9, Omega0=om0)
        D = diffusion_params(1.4, viscosity=dict(type="alpha", alpha=2e-2, eta_bulk=0.2, r0=0.9, Omega0=om0))
    else:
        o.set_viscosity("powerlaw", nu=0.03, eta_bulk=1.7, r_exp=-0.75, r0=0.9)
        D = diffusion_params(1.4, viscosity=dict(type="powerlaw", nu=0.03, eta_bulk=1.7, r_exp=-0.75, r0=0.9))
    tab = mb.viscosity_radial_table(D)
    # (zones at negative cylindrical radius behind an inner ghost layer hold pow(negative) = NaN, in the
    # reference too; no task reads a coefficient more than one zone outside the active region)
    assert tab.shape[1:] == mb.gas_prim.shape[2:]
    assert bool(torch.isfinite(tab[0][o.ks:o.ke + 1, o.js:o.je + 1, o.is_:o.ie + 1]).all())
    o.ZeroDiffusionFlux(), mb.ZeroDiffusionFlux()
    o.ViscousFlux(), mb.ViscousFlux(D)
    for d in range(o.ndim):
        same(mb.gas_diff_flux[d][0][face_slices(o, d)], o.qflux(d)[face_slices(o, d)], f"viscous flux x{d+1}")
    o.DiffusionUpdate(2.0e-4), mb.DiffusionUpdate(D, 2.0e-4)
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    same(mb.gas_u0[0][I], o.gu0[I], "DiffusionUpdate")
    hyd = mb.EstimateTimestepMesh(0, cfl=0.3)
    assert min(hyd, mb.DiffusionTimestep(D, 0.3)) == o.EstimateTimestepMesh(0)


def test_radial_viscosity_contract(hiplib):
    from artemis_amd import capi
    from artemis_amd.pack import diffusion_params
    o, mb = pair("cylindrical", (8, 8, 4), (0.3, -PI, -1.0), (4.3, PI, 1.0))
    D = diffusion_params(1.4, viscosity=dict(type="alpha", alpha=1e-3, r0=1.0, Omega0=1.0))
    with pytest.raises(capi.ArtemisHipError) as e:  # no table yet
        mb.ViscousFlux(D)
    assert e.value.code == capi.EINVAL and "radial" in str(e.value)
    D2 = diffusion_params(1.4, viscosity=dict(type="alpha", alpha=1e-3, r0=1.0))  # Omega0 = 0
    mb.viscosity_radial_table(D2)
    with pytest.raises(capi.ArtemisHipError) as e:
        mb.ViscousFlux(D2)
    assert e.value.code == capi.EINVAL and "omega0" in str(e.value)
    torch.cuda.synchronize()


def disk_pair(coordinates, nx, lo, hi, bcname, ns_dust, gam=1.4):
    from artemis_amd.pack import _ptr_table
    faces = [bcname if d < (3 if nx[2] > 1 else (2 if nx[1] > 1 else 1)) else "periodic" for d in range(3)]
    bc = tuple(f for f in faces for _ in range(2))
    o, mb = pair(coordinates, nx, lo, hi, ns_dust=ns_dust, bc=bc)
    o.set_gravity_point(mass=1.0)
    o.set_rotating_frame(0.7, 0.0)
    o.set_viscosity("alpha", alpha=1e-3, r0=1.0, Omega0=1.0)
    o.pgen_disk(r0=1.0, rho0=1.0, dslope=-2.25, flare=0.25, h0=0.05, dens_min=1e-10, pres_min=1e-15,
                polytropic_index=gam, dust_to_gas=0.02, post_init=False)
    push([o], mb)
    ic_g, ic_d = mb.gas_prim.clone(), mb.dust_prim.clone()
    keep = (ic_g, ic_d, _ptr_table(ic_g), _ptr_table(ic_d) if ns_dust else None)
    disk = dict(ic_gas=keep[2].data_ptr(), ic_dust=keep[3].data_ptr() if ns_dust else None, omf=0.7)
    return o, mb, bc, disk, keep


def perturb(o, rng):
    """the disk state with 10% noise: densities, sie and the inertial azimuthal velocity stay positive"""
    shp = (o.nk, o.nj, o.ni)
    ns, nd = o.cfg.ns_gas, o.cfg.ns_dust
    p = o.gprim
    p[0] *= 1.0 + 0.1 * rng.uniform(-1, 1, shp)
    p[5 * ns] *= 1.0 + 0.1 * rng.uniform(-1, 1, shp)
    for d in range(3):
        p[ns + d] += 0.02 * rng.normal(0.0, 1.0, shp)
    q = o.dprim if nd else None
    for n in range(nd):
        q[n] *= 1.0 + 0.1 * rng.uniform(-1, 1, shp)
        for d in range(3):
            q[nd + 3 * n + d] += 0.02 * rng.normal(0.0, 1.0, shp)
    o.PrimToCons()


@pytest.mark.parametrize("coordinates,nx,lo,hi", BLOCKS)
@pytest.mark.parametrize("ns_dust", [0, 2])
def test_disk_ic_condition(hiplib, coordinates, nx, lo, hi, ns_dust):
    """`ic` on every active face: the oracle re-evaluates the disk profile in each ghost zone (libm
    pow / exp per zone), the product copies from the stored initial primitives: bit-identical."""
    o, mb, bc, disk, keep = disk_pair(coordinates, nx, lo, hi, "ic", ns_dust)
    perturb(o, np.random.default_rng(71))
    push([o], mb)
    o.ApplyBoundaryConditions()
    mb.ApplyBoundaryConditions([bc], disk=disk)
    same(mb.gas_prim[0], o.gprim, "gas ghosts")
    if ns_dust:
        same(mb.dust_prim[0], o.dprim, "dust ghosts")


@pytest.mark.parametrize("coordinates,nx,lo,hi", BLOCKS)
@pytest.mark.parametrize("ns_dust", [0, 2])
def test_disk_extrap_condition(hiplib, coordinates, nx, lo, hi, ns_dust):
    """`extrap` on every active face.  log / exp of state ratios run on the device (ocml, <= 1 ulp
    each) against glibc in the oracle: relative agreement 1e-13 (the exponent dg * xma/dx is O(1),
    so a 1-ulp difference in a log moves the result by a few ulp)."""
    o, mb, bc, disk, keep = disk_pair(coordinates, nx, lo, hi, "disk_extrap", ns_dust)
    perturb(o, np.random.default_rng(72))
    push([o], mb)
    o.ApplyBoundaryConditions()
    mb.ApplyBoundaryConditions([bc], disk=disk)
    a, b = mb.gas_prim[0].cpu().numpy(), o.gprim
    assert np.isfinite(b).all()
    vscale = np.abs(b[1:4]).max()
    assert np.max(np.abs(a[[0, 5]] - b[[0, 5]]) / np.abs(b[[0, 5]])) < 1e-13
    assert np.max(np.abs(a[1:4] - b[1:4])) < 1e-13 * vscale
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    assert np.array_equal(a[I], b[I])  # active zones untouched
    if ns_dust:
        a, b = mb.dust_prim[0].cpu().numpy(), o.dprim
        assert np.max(np.abs(a[:2] - b[:2]) / np.abs(b[:2])) < 1e-13
        assert np.max(np.abs(a[2:] - b[2:])) < 1e-13 * np.abs(b[2:]).max()


@pytest.mark.parametrize("coordinates,nx,lo,hi", BLOCKS[:5])
@pytest.mark.parametrize("ns_dust", [0, 1])
def test_disk_viscous_condition(hiplib, coordinates, nx, lo, hi, ns_dust):
    """`viscous` on the radial faces (DiskBoundaryVisc, disk.hpp:415-595): extrapolated sie and
    azimuthal velocity, density and radial velocity from the steady viscous solution with
    nu(R) = nu0 (R/r0)^nu_indx -- log / exp / pow on the device, 1e-13 against glibc."""
    from artemis_amd.pack import MeshBlockPack
    bc = ("viscous", "viscous") + ("periodic",) * 4
    o, mb = pair(coordinates, nx, lo, hi, ns_dust=ns_dust, bc=bc)
    o.set_gravity_point(mass=1.0)
    o.set_rotating_frame(0.7, 0.0)
    o.set_viscosity("alpha", alpha=2e-2, r0=1.0, Omega0=1.0)
    o.pgen_disk(r0=1.0, rho0=1.0, dslope=-0.5, flare=0.25, h0=0.05, dens_min=1e-10, pres_min=1e-15,
                polytropic_index=1.0, dust_to_gas=0.02, mdot=3e-4, post_init=False)
    perturb(o, np.random.default_rng(73))
    push([o], mb)
    o.ApplyBoundaryConditions()
    nu0 = 2e-2 * 1.4 * (0.05 * 1.0 * 1.0) ** 2  # alpha gamma (h0 r0 Omega0)^2, disk.hpp:300-303
    mb.ApplyBoundaryConditions([bc], disk=dict(omf=0.7, nu0=nu0, nu_indx=1.5 + (2 * 0.25 - 1.0), r0=1.0, mdot=3e-4))
    a, b = mb.gas_prim[0].cpu().numpy(), o.gprim
    assert np.isfinite(b).all()
    assert np.max(np.abs(a[[0, 5]] - b[[0, 5]]) / np.abs(b[[0, 5]])) < 1e-13
    assert np.max(np.abs(a[1:4] - b[1:4])) < 1e-13 * np.abs(b[1:4]).max()
    assert not np.array_equal(a[0, :, :, :2], a[0, :, :, 2:4])  # the inner ghost zones were written
    if ns_dust:
        a, b = mb.dust_prim[0].cpu().numpy(), o.dprim
        assert np.max(np.abs(a[:1] - b[:1]) / np.abs(b[:1])) < 1e-13
        assert np.max(np.abs(a[1:] - b[1:])) < 1e-13 * np.abs(b[1:]).max()


@pytest.mark.parametrize("coordinates,nx,lo,hi", BLOCKS)
def test_beta_cooling(hiplib, coordinates, nx, lo, hi):
    """Gas::Cooling::BetaCooling<GEOM, powerlaw> (beta_cooling.cpp:40-126): Tref = tfloor + tcyl R^a +
    tsph r^b and beta = beta_min + beta0 exp(-s z^2/Tref) from the host-filled tables, bit-exact."""
    o, mb = pair(coordinates, nx, lo, hi, ns_gas=2)
    random_state(o, np.random.default_rng(81), shock=False, mach=0.5, contrast=10.0)
    push([o], mb)
    o.set_gravity_point(mass=1.3)
    kw = dict(beta0=2.0, beta_min=1e-3, exp_scale=0.3, tfloor=1e-3, tcyl=0.02, cyl_plaw=-1.0, tsph=0.01, sph_plaw=-0.5)
    o.set_cooling(**kw)
    c = mb.cooling_params(1.4, 1.3, **kw)
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    before = o.gu0[I].copy()
    o.CoolingSource(0.0, 3e-3), mb.CoolingSource(0.0, 3e-3, c)
    same(mb.gas_u0[0][I], o.gu0[I], "BetaCooling")
    assert np.array_equal(o.gu0[I][:8], before[:8]) and not np.array_equal(o.gu0[I][8:], before[8:])


def test_disk_condition_contract(hiplib):
    from artemis_amd import capi
    o, mb, bc, disk, keep = disk_pair("axisymmetric", (8, 8, 1), (0.3, -2.0, -0.5), (4.3, 2.0, 0.5), "ic", 0)
    with pytest.raises(capi.ArtemisHipError) as e:
        mb.ApplyBoundaryConditions([bc])
    assert e.value.code == capi.EINVAL
    with pytest.raises(capi.ArtemisHipError) as e:
        mb.ApplyBoundaryConditions([bc], disk=dict(omf=0.7))
    assert e.value.code == capi.EINVAL and "ic_gas" in str(e.value)
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [
    ("ic", "ic", "none", "none", "ic", "ic"),
    ("ic", "none", "none", "none", "ic", "none"),
    ("ic", "ic", "periodic", "periodic", "outflow", "ic"),
    ("outflow", "ic", "ic", "reflecting", "periodic", "periodic"),
    ("reflecting", "reflecting", "ic", "ic", "ic", "outflow"),
    ("none", "ic", "outflow", "none", "none", "ic"),
])
@pytest.mark.parametrize("nx", [(16, 8, 8), (12, 6, 1), (4, 4, 4), (8, 8, 8)])
def test_ic_faces_inside_the_one_launch_fill(hiplib, option, flags, nx):
    """`ic` faces ride the one-launch boundary fill of the copy-type conditions since round 6 (bc_shell_kernel: a zone whose
    last covering pass is an `ic` pass reads the initial-state tables).  Against the sequential per-face passes
    (NO_IC_IN_SHELL: parthenon's order literally -- periodic images, then x1, x2, x3 over the entire extent of the other
    directions) on random states, random initial-state tables, several blocks, gas + dust, every mixture of `ic` with
    neighbour faces (none), periodic, outflow and reflecting ones, 3-D and 2-D: every zone of every array, bit for bit."""
    import torch
    from artemis_amd.pack import MeshBlockPack, _ptr_table
    ndim = 3 if nx[2] > 1 else 2
    if ndim == 2:
        flags = flags[:4] + ("none", "none")
    kw = dict(ng=2, ns_gas=1, ns_dust=2, reconstruct="plm", riemann="hlle", dust_reconstruct="plm", dust_riemann="hlle",
              gamma=1.4, dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10, coordinates="cylindrical")
    los = [(0.5, 0.0, -1.0), (0.5, 1.0, -1.0), (0.5, 2.0, -1.0)]
    his = [(2.0, 1.0, 1.0), (2.0, 2.0, 1.0), (2.0, 3.0, 1.0)]
    mb = MeshBlockPack(3, nx, los, his, with_fluxes=False, **kw)
    g = torch.Generator(device="cpu").manual_seed(7)
    state_g = torch.rand(mb.gas_prim.shape, generator=g, dtype=torch.float64) + 0.1
    state_d = torch.rand(mb.dust_prim.shape, generator=g, dtype=torch.float64) + 0.1
    ic_g = (torch.rand(mb.gas_prim.shape, generator=g, dtype=torch.float64) + 2.0).cuda()
    ic_d = (torch.rand(mb.dust_prim.shape, generator=g, dtype=torch.float64) + 2.0).cuda()
    ic_g[:, 0, 0, 1, 1] = 1e-12  # (a value below the floors: no floor pass runs on this per-task call, it must survive)
    tg, td = _ptr_table(ic_g), _ptr_table(ic_d)
    disk = dict(ic_gas=tg.data_ptr(), ic_dust=td.data_ptr())
    out = []
    for old in (False, True):
        if old:
            option("no_ic_in_shell")
        mb.gas_prim.copy_(state_g), mb.dust_prim.copy_(state_d)
        mb.ApplyBoundaryConditions([flags] * 3, disk=disk)
        torch.cuda.synchronize()
        out.append((mb.gas_prim.cpu().numpy().copy(), mb.dust_prim.cpu().numpy().copy()))
    keep = [0, 1, 2, 3, 5]
    assert np.array_equal(out[0][0][:, keep], out[1][0][:, keep])
    assert np.array_equal(out[0][1], out[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("nx", [(4, 4, 4), (8, 8, 8)])
def test_ic_faces_of_many_blocks_with_their_own_flags(hiplib, option, nx):
    """The coarse buffers of a refined disk: hundreds of small blocks (more than one batch of the one-launch fill), each
    with its own mixture of `ic` and neighbour faces, gas + one dust species.  One-launch fill == sequential passes."""
    import torch
    from artemis_amd.pack import MeshBlockPack, _ptr_table
    nb = 600
    kw = dict(ng=2, ns_gas=1, ns_dust=1, reconstruct="plm", riemann="hlle", dust_reconstruct="plm", dust_riemann="hlle",
              gamma=1.4, dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10, coordinates="cylindrical")
    los = [(0.5, 0.01 * b, -1.0) for b in range(nb)]
    his = [(2.0, 0.01 * (b + 1), 1.0) for b in range(nb)]
    mb = MeshBlockPack(nb, nx, los, his, with_fluxes=False, **kw)
    rng = np.random.default_rng(11)
    names = ["none", "ic", "ic", "none", "outflow", "reflecting"]
    flags = [tuple(names[q] for q in rng.integers(0, 4 if b % 3 else 6, size=6)) for b in range(nb)]
    g = torch.Generator(device="cpu").manual_seed(7)
    state_g = torch.rand(mb.gas_prim.shape, generator=g, dtype=torch.float64) + 0.1
    state_d = torch.rand(mb.dust_prim.shape, generator=g, dtype=torch.float64) + 0.1
    ic_g = (torch.rand(mb.gas_prim.shape, generator=g, dtype=torch.float64) + 2.0).cuda()
    ic_d = (torch.rand(mb.dust_prim.shape, generator=g, dtype=torch.float64) + 2.0).cuda()
    tg, td = _ptr_table(ic_g), _ptr_table(ic_d)
    disk = dict(ic_gas=tg.data_ptr(), ic_dust=td.data_ptr())
    out = []
    for old in (False, True):
        if old:
            option("no_ic_in_shell")
        mb.gas_prim.copy_(state_g), mb.dust_prim.copy_(state_d)
        mb.ApplyBoundaryConditions(flags, disk=disk)
        torch.cuda.synchronize()
        out.append((mb.gas_prim.cpu().numpy().copy(), mb.dust_prim.cpu().numpy().copy()))
    keep = [0, 1, 2, 3, 5]
    assert np.array_equal(out[0][0][:, keep], out[1][0][:, keep])
    assert np.array_equal(out[0][1], out[1][1])
