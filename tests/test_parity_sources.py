"""GPU parity of the source-term tasks and the strat user boundary conditions against the CPU
oracle, BIT-EXACT: ExternalGravity (uniform, point mass), RotatingFrameForce (shearing box),
DragSource (simple_dust constant / stokes, self damping) -- reference artemis_driver.cpp:222-241,
gravity/*.cpp, rotating_frame_impl.hpp:28-93, drag.hpp:171-482, pgen/strat.hpp:158-466."""
import numpy as np
import pytest
import torch

from oracle.oracle import Oracle
from test_parity_ops import push, random_state, same

pytestmark = pytest.mark.gpu
BIG = 1.7976931348623157e308


def pair(nx, lo, hi, ns_gas=1, ns_dust=2, coordinates="cartesian", seed=0, bc=("outflow",) * 6, ng=2):
    from artemis_amd.pack import MeshBlockPack
    kw = dict(ng=ng, ns_gas=ns_gas, ns_dust=ns_dust, reconstruct="plm", riemann="hlle",
              dust_reconstruct="plm", dust_riemann="hlle", gamma=1.4, dfloor=1e-10, siefloor=1e-10,
              dust_dfloor=1e-10, coordinates=coordinates)
    o = Oracle(nx, lo, hi, bc=bc, **kw)
    random_state(o, np.random.default_rng(seed), shock=False)
    mb = MeshBlockPack(1, nx, [lo], [hi], **kw)
    push([o], mb)
    return o, mb


def interior(o):
    return (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))


def check_cons(o, mb, what):
    I = interior(o)
    if o.cfg.ns_gas:
        same(mb.gas_u0[0][I], o.gu0[I], what + " gas cons")
    if o.cfg.ns_dust:
        same(mb.dust_u0[0][I], o.du0[I], what + " dust cons")


GRAV_GEOMS = [("cartesian", (20, 12, 8), (-1.0, -0.5, -0.25), (1.0, 0.5, 0.75)),
              ("cartesian", (33, 9, 1), (-1.0, -1.0, -0.2), (1.0, 1.0, 0.2)),
              ("cartesian", (40, 1, 1), (0.5, -0.5, -0.5), (2.0, 0.5, 0.5)),
              ("spherical", (40, 1, 1), (0.2, 0.0, -0.5), (2.0, np.pi, 0.5)),
              ("spherical", (24, 10, 1), (0.3, 0.6, -0.5), (2.0, 2.5, 0.5)),
              ("axisymmetric", (24, 12, 1), (0.1, -1.0, -0.5), (2.0, 1.0, 0.5))]
# systems that go through ConvertToCartWithVec (point_mass.cpp:91-112): an offset mass is allowed
GRAV_FRAME = [("cylindrical", (16, 12, 6), (0.5, 0.0, -1.0), (2.0, 2 * np.pi, 1.0)),   # disk_cyl.in layout
              ("cylindrical", (16, 8, 1), (0.5, 0.3, -0.5), (2.0, 2.0, 0.5)),
              ("spherical", (16, 8, 6), (0.3, 0.6, 0.0), (1.5, 2.5, 6.0)),              # disk_sph.in layout
              ("spherical", (12, 10, 8), (0.2, 1.06, -np.pi), (5.6, 2.08, np.pi))]


@pytest.mark.parametrize("coordinates,nx,lo,hi", GRAV_GEOMS + GRAV_FRAME)
def test_point_mass_gravity(hiplib, coordinates, nx, lo, hi):
    from artemis_amd.pack import gravity_point
    o, mb = pair(nx, lo, hi, ns_gas=2, ns_dust=2, coordinates=coordinates, seed=21)
    cart = coordinates == "cartesian" or (coordinates, nx, lo, hi) in GRAV_FRAME
    pos = (0.13, -0.07, 0.2) if cart else (0.0, 0.0, 0.0)
    # softened, with an active sink region so density and energy are drained too
    o.set_gravity_point(2.5, soft=0.05, sink=0.6, sink_rate=3.0, x=pos[0], y=pos[1], z=pos[2])
    g = gravity_point(2.5, soft=0.05, sink=0.6, sink_rate=3.0, pos=pos)
    o.ExternalGravity(0.3, 1.0e-3)
    mb.ExternalGravity(0.3, 1.0e-3, g)
    check_cons(o, mb, "PointMassGravity")
    # sink radius 0: (dr - 0)/0 is inf or nan in the reference arithmetic, the drain stays off
    o.set_gravity_point(2.5, soft=0.0, sink=0.0, sink_rate=0.0, x=pos[0], y=pos[1], z=pos[2])
    g = gravity_point(2.5, pos=pos)
    o.ExternalGravity(0.3, 1.0e-3)
    mb.ExternalGravity(0.3, 1.0e-3, g)
    check_cons(o, mb, "PointMassGravity, no sink")


@pytest.mark.parametrize("coordinates,nx,lo,hi", GRAV_GEOMS[:3] + GRAV_FRAME)
def test_binary_gravity(hiplib, coordinates, nx, lo, hi):
    """BinaryMassGravity (binary_mass.cpp:27-203): two softened point masses with sinks at the
    positions Orbit::solve (gravity.hpp:66-94) gives for the time of the step, in a frame rotating
    with omf; Cartesian, cylindrical, spherical3D."""
    from artemis_amd.pack import binary_orbit, gravity_binary
    o, mb = pair(nx, lo, hi, ns_gas=2, ns_dust=2, coordinates=coordinates, seed=31)
    kw = dict(soft1=0.02, soft2=0.03, sink1=0.5, sink2=0.4, sink_rate1=2.0, sink_rate2=5.0)
    o.set_rotating_frame(0.8, 0.0)
    o.set_gravity_binary(1.7, 0.3, a=0.9, e=0.2, i=20.0, omega=35.0, Omega=50.0, f=110.0, x=0.1, y=-0.05, z=0.02, **kw)
    t = 0.37
    rb = binary_orbit(1.7, 0.9, e=0.2, i=20.0, omega=35.0, Omega=50.0, f=110.0)(t, 0.8)
    g = gravity_binary(1.7, 0.3, rb, com=(0.1, -0.05, 0.02), **kw)
    o.ExternalGravity(t, 1.0e-3)
    mb.ExternalGravity(t, 1.0e-3, g)
    check_cons(o, mb, "BinaryMassGravity")


@pytest.mark.parametrize("coordinates,nx,lo,hi", [GRAV_GEOMS[0], GRAV_GEOMS[1]] + GRAV_FRAME)
@pytest.mark.parametrize("omf", [0.0, 0.7])
@pytest.mark.parametrize("species", [(2, 2), (1, 1), (1, 0)])  # (1, 1) / (1, 0): the one-species kernel (state in registers)
def test_nbody_gravity(hiplib, coordinates, nx, lo, hi, omf, species):
    """Gravity::NBodyGravity<GEOM> (gravity/nbody_gravity.hpp:28-221, nbody/particle_base.hpp:96-258): three particles
    -- spline softening with an accreting sink, Plummer softening with a sink that also removes tangential momentum,
    an uncoupled one -- in a (optionally rotating) frame; Cartesian, cylindrical, spherical3D, gas + dust.  The fluid
    update is bitwise; the seven back-reaction sums per particle (what NBody::Advance hands to REBOUND,
    nbody_advance.cpp:123-131) are reduced in a different order than the oracle's serial loop: 1e-12 relative."""
    o, mb = pair(nx, lo, hi, ns_gas=species[0], ns_dust=species[1], coordinates=coordinates, seed=77)
    parts = [dict(GM=1.3, pos=(0.21, -0.1, 0.05), vel=(0.1, 0.4, -0.2), rs=0.3, racc=0.9, gamma=4.0, beta=0.0, spline=1),
             dict(GM=0.4, pos=(-0.3, 0.25, 0.1), vel=(0.0, -0.3, 0.1), xf=(0.01, 0.02, 0.0), vf=(0.0, 0.1, 0.0), rs=0.1,
                  racc=0.8, gamma=2.0, beta=6.0, spline=0),
             dict(GM=5.0, pos=(0.0, 0.0, 0.0), couple=0)]
    if omf:
        o.set_rotating_frame(omf, 0.0)
    o.set_gravity_nbody(parts, frame_correction=True)
    dt = 2.0e-3
    o.ExternalGravity(0.1, dt)
    want = o.nbody_force(reset=True)
    got = mb.NBodyGravity(0.1, dt, parts, omf=omf)
    check_cons(o, mb, "NBodyGravity")
    assert np.all(got[2] == 0.0) and np.all(want[2] == 0.0)  # couple = 0
    scale = np.abs(want).max(axis=1, keepdims=True) + 1e-300
    assert np.max(np.abs(got - want) / scale) < 1e-12, (got, want)
    if species == (2, 2):
        assert np.abs(want[0, 0]) > 0 and np.abs(want[1, 4:]).max() > 0  # the sinks did accrete
    # a second call accumulates on the caller's side exactly like the reference's particle_force rows
    o.ExternalGravity(0.1, dt)
    got2 = mb.NBodyGravity(0.1, dt, parts, omf=omf)
    check_cons(o, mb, "NBodyGravity, second call")
    assert np.max(np.abs(got2 - o.nbody_force()) / scale) < 1e-12


@pytest.mark.parametrize("coordinates,nx,lo,hi", GRAV_GEOMS + [
    ("cylindrical", (16, 8, 6), (0.5, 0.0, -1.0), (2.0, 6.0, 1.0)),
    ("spherical", (16, 8, 6), (0.3, 0.6, 0.0), (1.5, 2.5, 6.0))])
def test_uniform_gravity_and_time_window(hiplib, coordinates, nx, lo, hi):
    from artemis_amd.pack import gravity_uniform
    o, mb = pair(nx, lo, hi, coordinates=coordinates, seed=22)
    o.set_gravity_uniform(0.3, -1.1, 0.7)
    g = gravity_uniform(0.3, -1.1, 0.7)
    o.ExternalGravity(0.0, 2.0e-3)
    mb.ExternalGravity(0.0, 2.0e-3, g)
    check_cons(o, mb, "UniformGravity")
    before = mb.gas_u0.clone()
    g.tstart, g.tstop = 1.0, 2.0  # gravity.cpp:134: active for tstart <= time < tstop only
    mb.ExternalGravity(2.0, 2.0e-3, g)
    assert torch.equal(mb.gas_u0, before)
    mb.ExternalGravity(1.0, 2.0e-3, g)
    assert not torch.equal(mb.gas_u0, before)


@pytest.mark.parametrize("nx", [(24, 12, 8), (33, 9, 1), (40, 1, 1)])
def test_shearing_box(hiplib, nx):
    o, mb = pair(nx, (-1.0, -0.5, -0.3), (1.0, 0.5, 0.3), ns_gas=2, ns_dust=3, seed=23)
    o.set_rotating_frame(0.9, 1.5)
    o.RotatingFrameForce(1.5e-3)
    mb.RotatingFrameForce(0.9, 1.5, 0.0, 1.5e-3)
    check_cons(o, mb, "ShearingBox")


RF_GEOMS = [("spherical", (40, 1, 1), (0.2, 0.0, -0.5), (2.0, np.pi, 0.5)),
            ("spherical", (24, 10, 1), (0.3, 0.6, -0.5), (2.0, 2.5, 0.5)),
            ("spherical", (16, 8, 6), (0.3, 0.6, 0.0), (1.5, 2.5, 6.0)),
            ("cylindrical", (16, 12, 6), (0.5, 0.0, -1.0), (2.0, 2 * np.pi, 1.0)),
            ("cylindrical", (33, 1, 1), (0.3, -0.5, -0.5), (1.0, 0.5, 0.5)),
            ("axisymmetric", (24, 12, 1), (0.1, -1.0, -0.5), (2.0, 1.0, 0.5)),
            ("axisymmetric", (12, 8, 6), (0.7, -1.0, 0.0), (2.0, 1.0, 1.0))]


@pytest.mark.parametrize("coordinates,nx,lo,hi", RF_GEOMS)
def test_rotating_frame_curvilinear(hiplib, coordinates, nx, lo, hi):
    """RotatingFrameImpl<GEOM> (rotating_frame_impl.hpp:95-199): angular-momentum-conserving source
    from the stage's mass fluxes with the RFWeights of each system, gas and dust."""
    o, mb = pair(nx, lo, hi, ns_gas=2, ns_dust=2, coordinates=coordinates, seed=29)
    for fluid in (0, 1):
        o.CalculateFluxes(fluid, False)
        mb.CalculateFluxes(fluid, False)
    o.set_rotating_frame(0.9, 0.0)
    mb.pack.omega_frame = 0.9
    # the centrifugal / Coriolis part lives in FluxSource's coordinate sources: (v + vf)^2 dh/dx with
    # vf = RotationVelocity(xv, omf) (fluid_fluxes.hpp:345, :395-415)
    before = mb.gas_u0.clone()
    for fluid in (0, 1):
        o.FluxSource(1.5e-3, fluid)
        mb.FluxSource(1.5e-3, fluid)
    check_cons(o, mb, "FluxSource in the rotating frame")
    mb2_u0 = mb.gas_u0.clone()
    mb.gas_u0.copy_(before)
    mb.pack.omega_frame = 0.0
    mb.FluxSource(1.5e-3, 0)
    assert not torch.equal(mb.gas_u0, mb2_u0)  # the frame velocity matters
    mb.gas_u0.copy_(mb2_u0)
    mb.pack.omega_frame = 0.9
    o.RotatingFrameForce(1.5e-3)
    mb.RotatingFrameForce(0.9, 0.0, 0.0, 1.5e-3)
    check_cons(o, mb, "RotatingFrame")


DRAG_GEOMS = [("cartesian", (20, 12, 8), (-1.0, -0.5, -0.25), (1.0, 0.5, 0.75)),
              ("cartesian", (40, 1, 1), (0.0, -0.5, -0.5), (1.0, 0.5, 0.5)),
              ("cylindrical", (16, 8, 6), (0.5, 0.0, -1.0), (2.0, 6.0, 1.0)),
              ("spherical", (24, 10, 1), (0.3, 0.6, -0.5), (2.0, 2.5, 0.5)),
              ("spherical", (16, 8, 6), (0.3, 0.6, 0.0), (1.5, 2.5, 6.0)),
              ("spherical", (40, 1, 1), (0.2, 0.0, -0.5), (2.0, np.pi, 0.5)),
              ("axisymmetric", (24, 12, 1), (0.1, -1.0, -0.5), (2.0, 1.0, 0.5))]


@pytest.mark.parametrize("coordinates,nx,lo,hi", DRAG_GEOMS)
@pytest.mark.parametrize("model", ["constant", "stokes"])
def test_simple_dust_drag(hiplib, coordinates, nx, lo, hi, model):
    """Implicit gas-dust coupling with 4 species spanning stiff to loose stopping times (one is
    tau <= 0 -> infinitely stiff, drag.hpp:410), with damping ramps active on both fluids."""
    from artemis_amd.pack import drag_params
    o, mb = pair(nx, lo, hi, ns_gas=1, ns_dust=4, coordinates=coordinates, seed=24)
    tau = [1e-3, 0.1, 0.0, 10.0]
    sizes = [1e-4, 1e-3, 1e-2, 0.1]
    span = [hi[d] - lo[d] for d in range(3)]
    inner = tuple(lo[d] + 0.3 * span[d] for d in range(3))
    outer = tuple(hi[d] - 0.2 * span[d] for d in range(3))
    damp = dict(inner=inner, inner_rate=(2.0, 0.5, 1.0), outer=outer, outer_rate=(1.0, 3.0, 0.25))
    o.set_drag("simple_dust", model, tau=tau, scale=1.5, grain_density=2.0, sizes=sizes)
    o.set_damping(0, **damp)
    o.set_damping(1, **damp)
    d = drag_params("simple_dust", model, tau=tau, scale=1.5, grain_density=2.0, sizes=sizes,
                    mesh_min=lo, mesh_max=hi, gas_damping=damp, dust_damping=damp)
    o.DragSource(0.02)
    mb.DragSource(0.0, 0.02, d)
    check_cons(o, mb, "SimpleDragSource " + model)
    # total momentum of gas + dust is conserved by the exchange when nothing is damped
    o.set_damping(0), o.set_damping(1)
    d = drag_params("simple_dust", model, tau=tau, scale=1.5, grain_density=2.0, sizes=sizes,
                    mesh_min=lo, mesh_max=hi)
    m0 = (o.gu0[1:4].sum(axis=0) * 0 + o.gu0[1] + o.du0[4] + o.du0[7] + o.du0[10] + o.du0[13])[interior(o)[1:]]
    o.DragSource(0.02)
    mb.DragSource(0.0, 0.02, d)
    check_cons(o, mb, "SimpleDragSource undamped " + model)
    m1 = (o.gu0[1] + o.du0[4] + o.du0[7] + o.du0[10] + o.du0[13])[interior(o)[1:]]
    assert np.max(np.abs(m1 - m0) / (np.abs(m0) + 1.0)) < 1e-12


@pytest.mark.parametrize("coordinates,nx,lo,hi", DRAG_GEOMS[:4])
def test_self_drag(hiplib, coordinates, nx, lo, hi):
    from artemis_amd.pack import drag_params
    o, mb = pair(nx, lo, hi, ns_gas=2, ns_dust=2, coordinates=coordinates, seed=25)
    span = [hi[d] - lo[d] for d in range(3)]
    damp = dict(inner=tuple(lo[d] + 0.25 * span[d] for d in range(3)), inner_rate=(5.0, 1.0, 2.0),
                outer=tuple(hi[d] - 0.25 * span[d] for d in range(3)), outer_rate=(0.5, 4.0, 1.0))
    o.set_drag("self", "constant", tau=[1.0, 1.0])
    o.set_damping(0, **damp)
    o.set_damping(1, **damp)
    d = drag_params("self", mesh_min=lo, mesh_max=hi, gas_damping=damp, dust_damping=damp)
    o.DragSource(0.05)
    mb.DragSource(0.0, 0.05, d)
    check_cons(o, mb, "SelfDragSource")


DAMP_VISC = [dict(type="constant", nu=0.02), dict(type="powerlaw", nu=0.03, r_exp=0.5, r0=1.2),
             dict(type="alpha", alpha=0.05, r0=1.1, Omega0=0.8)]


@pytest.mark.parametrize("coordinates,nx,lo,hi", DRAG_GEOMS[2:])
@pytest.mark.parametrize("visc", DAMP_VISC, ids=lambda v: v["type"])
@pytest.mark.parametrize("kind", ["self", "simple_dust"])
def test_drag_damp_to_visc(hiplib, coordinates, nx, lo, hi, visc, kind):
    """<gas/damping> damp_to_visc (drag.cpp:109-121,135-157): the gas is damped towards the viscous inflow
    velocity v_R = -1.5 mu / (R rho) along the cylindrical radius, mu from the gas viscosity evaluated with the
    conserved density and GetSpecificInternalEnergy; the radial factor of mu is the host-filled table, so
    the task stays bit-exact.  Both couplings, both stopping-time models' shared path, every curvilinear system."""
    from artemis_amd.pack import diffusion_params, drag_params
    ns_gas = 2 if kind == "self" else 1
    o, mb = pair(nx, lo, hi, ns_gas=ns_gas, ns_dust=2, coordinates=coordinates, seed=31)
    span = [hi[d] - lo[d] for d in range(3)]
    damp = dict(inner=tuple(lo[d] + 0.3 * span[d] for d in range(3)), inner_rate=(4.0, 1.0, 2.0),
                outer=tuple(hi[d] - 0.3 * span[d] for d in range(3)), outer_rate=(0.5, 3.0, 1.0))
    o.set_viscosity(**visc)
    o.set_damp_to_visc(True)
    o.set_drag(kind, "constant", tau=[0.05, 2.0])
    o.set_damping(0, **damp)
    o.set_damping(1, **damp)
    D = diffusion_params(1.4, viscosity=visc)
    if visc["type"] != "constant":
        mb.viscosity_radial_table(D)
    d = drag_params(kind, "constant", tau=[0.05, 2.0], mesh_min=lo, mesh_max=hi, gas_damping=damp, dust_damping=damp,
                    damp_visc=D.visc)
    before = o.gu0.copy()
    o.DragSource(0.04)
    mb.DragSource(0.0, 0.04, d)
    check_cons(o, mb, "DragSource damp_to_visc " + kind)
    # and it is not the plain damping: the target velocity changed the answer
    o2, _ = pair(nx, lo, hi, ns_gas=ns_gas, ns_dust=2, coordinates=coordinates, seed=31)
    o2.set_drag(kind, "constant", tau=[0.05, 2.0])
    o2.set_damping(0, **damp), o2.set_damping(1, **damp)
    o2.DragSource(0.04)
    assert np.max(np.abs(o2.gu0 - o.gu0)) > 1e-6 * np.max(np.abs(before))


def test_damp_to_visc_contract(hiplib):
    from artemis_amd import capi
    from artemis_amd.pack import diffusion_params, drag_params
    o, mb = pair((16, 8, 6), (0.5, 0.0, -1.0), (2.0, 6.0, 1.0), ns_gas=1, ns_dust=1, coordinates="cylindrical")
    D = diffusion_params(1.4, conductivity=dict(type="conductivity", cond=0.1))
    with pytest.raises(capi.ArtemisHipError) as e:  # drag.cpp:120
        mb.DragSource(0.0, 1e-3, drag_params("self", damp_visc=D.cond))
    assert "does not work with damping" in str(e.value)
    D = diffusion_params(1.4, viscosity=dict(type="alpha", alpha=0.05, Omega0=1.0))
    with pytest.raises(capi.ArtemisHipError) as e:
        mb.DragSource(0.0, 1e-3, drag_params("self", damp_visc=D.visc))
    assert "radial table" in str(e.value)
    with pytest.raises(ValueError):
        o.set_damp_to_visc(True)  # no viscosity set


@pytest.mark.parametrize("nx,ns_dust", [((24, 16, 1), 0), ((24, 16, 1), 2), ((16, 12, 6), 1), ((30, 1, 1), 1)])
def test_strat_boundary_conditions(hiplib, nx, ns_dust):
    """extrap on x1, inflow on x2 (strat.hpp:158-466) incl. the corner zones, where the x2 pass
    reads what the x1 pass wrote; outflow on x3 so the three passes mix user and built-in fills."""
    bc = ("extrap", "extrap", "inflow", "inflow", "outflow", "outflow")
    o, mb = pair(nx, (-1.0, -1.0, -0.3), (1.0, 1.0, 0.3), ns_gas=1, ns_dust=ns_dust, seed=26, bc=bc)
    o.set_rotating_frame(1.1, 1.5)
    o.ApplyBoundaryConditions()
    mb.ApplyBoundaryConditions([bc], strat=(1.5, 1.1))
    same(mb.gas_prim[0], o.gprim, "gas ghosts")
    if ns_dust:
        same(mb.dust_prim[0], o.dprim, "dust ghosts")


@pytest.mark.parametrize("ns_dust", [0, 2])
def test_strat_vertical_extrap_condition(hiplib, ns_dust):
    """`extrap` on the x3 faces of the 3-D stratified box (strat.hpp:476-640): copy with no inflow in v3 and
    the density continued as rho_a (rho_b / rho_a)^((z - z_a)/dz) -- std::pow of a state ratio, evaluated
    with the device's pow(): 1e-13 relative on the densities, everything else bit for bit."""
    bc = ("extrap", "extrap", "inflow", "inflow", "extrap", "extrap")
    o, mb = pair((20, 12, 10), (-1.0, -1.0, -0.6), (1.0, 1.0, 0.6), ns_gas=1, ns_dust=ns_dust, seed=33, bc=bc)
    o.set_rotating_frame(1.1, 1.5)
    o.ApplyBoundaryConditions()
    mb.ApplyBoundaryConditions([bc], strat=(1.5, 1.1))
    a, b = mb.gas_prim[0].cpu().numpy(), o.gprim
    assert np.array_equal(a[1:], b[1:]) and np.max(np.abs(a[0] - b[0]) / b[0]) < 1e-13
    assert not np.array_equal(a[0, :2], a[0, 2:4])  # the lower x3 ghost layers were written
    if ns_dust:
        a, b = mb.dust_prim[0].cpu().numpy(), o.dprim
        assert np.array_equal(a[ns_dust:], b[ns_dust:]) and np.max(np.abs(a[:ns_dust] - b[:ns_dust]) / b[:ns_dust]) < 1e-13


def test_source_abi_contract(hiplib):
    import ctypes as C
    from artemis_amd import capi
    from artemis_amd.pack import MeshBlockPack, drag_params, gravity_point
    mb = MeshBlockPack(1, (8, 8, 4), [(0.5, 0.6, 0.0)], [(1.0, 2.5, 1.0)], ns_dust=1, coordinates="spherical")
    cyl = MeshBlockPack(1, (8, 8, 4), [(0.5, 0.0, 0.0)], [(1.0, 2.5, 1.0)], ns_dust=1, coordinates="cylindrical")
    cyl.pack.metric = None
    with pytest.raises(capi.ArtemisHipError) as e:  # ConvertToCartWithVec needs cos / sin of the azimuth
        cyl.ExternalGravity(0.0, 1e-3, gravity_point(1.0))
    assert e.value.code == capi.EINVAL and "metric" in str(e.value)
    from artemis_amd.pack import gravity_binary
    axi = MeshBlockPack(1, (8, 8, 1), [(0.5, -1.0, 0.0)], [(1.0, 1.0, 1.0)], ns_dust=1, coordinates="axisymmetric")
    with pytest.raises(capi.ArtemisHipError) as e:  # gravity.cpp:82-83
        axi.ExternalGravity(0.0, 1e-3, gravity_binary(1.0, 0.1, (1.0, 0.0, 0.0)))
    assert e.value.code == capi.EINVAL and "Binary gravity" in str(e.value)
    with pytest.raises(capi.ArtemisHipError) as e:  # rotating_frame.cpp:34-38
        mb.RotatingFrameForce(1.0, 1.5, 0.0, 1e-3)
    assert e.value.code == capi.EINVAL and "qshear" in str(e.value)
    with pytest.raises(capi.ArtemisHipError) as e:  # rotating_frame.cpp:31-32
        mb.RotatingFrameForce(0.0, 0.0, 0.0, 1e-3)
    assert "omega cannot be zero" in str(e.value)
    cart = MeshBlockPack(1, (8, 8, 1), [(0, 0, 0)], [(1, 1, 1)], ns_dust=0)
    with pytest.raises(capi.ArtemisHipError) as e:  # drag.cpp:71-72
        cart.DragSource(0.0, 1e-3, drag_params("simple_dust", tau=[1.0]))
    assert "do_gas = do_dust = true" in str(e.value)
    with pytest.raises(capi.ArtemisHipError) as e:  # strat flags without their parameters
        cart.ApplyBoundaryConditions([("extrap", "extrap", "inflow", "inflow", "outflow", "outflow")])
    assert e.value.code == capi.EINVAL
    with pytest.raises(capi.ArtemisHipError) as e:  # inflow is an x2 condition
        cart.ApplyBoundaryConditions([("inflow",) * 6], strat=(1.5, 1.0))
    assert e.value.code == capi.EINVAL
    torch.cuda.synchronize()


COND_BLOCKS = [
    ("cartesian", (24, 10, 6), (0.2, -0.5, -0.3), (1.2, 0.5, 0.3)),
    ("cartesian", (33, 9, 1), (0.2, -0.5, -0.3), (1.2, 0.5, 0.3)),
    ("cartesian", (40, 1, 1), (0.2, -0.5, -0.3), (1.2, 0.5, 0.3)),
    ("spherical", (32, 1, 1), (0.2, 0.0, -0.5), (1.2, np.pi, 0.5)),       # thermal_diffusion.py "sph"
    ("spherical", (16, 10, 6), (0.3, 0.7, 0.0), (1.7, 2.5, 2 * np.pi)),
    ("axisymmetric", (32, 1, 1), (0.2, -0.5, -0.5), (1.2, 0.5, 0.5)),     # thermal_diffusion.py "axi"
    ("axisymmetric", (12, 8, 6), (0.7, -1.0, 0.0), (2.0, 1.0, 1.0)),
    ("cylindrical", (16, 12, 6), (0.5, 0.0, -1.0), (2.0, 2 * np.pi, 1.0)),
]


@pytest.mark.parametrize("coordinates,nx,lo,hi", COND_BLOCKS[:1] + COND_BLOCKS[4:5])
def test_conductive_boundary_conditions_power_law(hiplib, coordinates, nx, lo, hi):
    """the same condition with K = K0 (T/T_ref)^2.5 (rho/rho_ref)^-0.5 at the active zone: pow() of the
    state on the device, 1e-13 on the ghost density and sie."""
    from artemis_amd import capi
    bc = ("conductive",) * 6
    o, mb = pair(nx, lo, hi, ns_gas=1, ns_dust=0, seed=27, bc=bc, coordinates=coordinates)
    o.set_gravity_uniform(-0.3, 0.2, 0.1)
    law = dict(temp_exp=2.5, rho_exp=-0.5, rho_ref=0.7, T_ref=1.3)
    o.set_conductivity("conductivity", cond=0.1, **law)
    o.pgen_conduction(gas_rho=1.0, gas_temp=0.05, flux=0.01, post_init=False)
    random_state(o, np.random.default_rng(28), shock=False, contrast=10.0)
    push([o], mb)
    o.ApplyBoundaryConditions()
    mb.ApplyBoundaryConditions([bc], conductive=dict(
        temp=0.05, flux=0.01, g=(-0.3, 0.2, 0.1), coeff=0.1, cv=1.0 / ((1.4 - 1.0) * 1.0 * 1.0),
        type=capi.CONDUCTIVITY_PLAW, **law))
    a, b = mb.gas_prim[0].cpu().numpy(), o.gprim
    assert np.array_equal(a[1:4], b[1:4])
    for v in (0, 5):  # (sie is clamped to zero where the extrapolated temperature is negative)
        assert np.array_equal(np.isfinite(a[v]), np.isfinite(b[v]))
        m = np.isfinite(b[v])
        assert np.max(np.abs(a[v][m] - b[v][m])) < 1e-12 * np.abs(b[v][m]).max()


@pytest.mark.parametrize("coordinates,nx,lo,hi", COND_BLOCKS)
@pytest.mark.parametrize("ctype", ["conductivity", "diffusivity"])
def test_conductive_boundary_conditions(hiplib, coordinates, nx, lo, hi, ctype):
    """`conductive` on every active face (pgen/conduction.hpp:105-232) with uniform gravity: fixed
    flux at inner faces, fixed temperature at outer ones, hydrostatic density; the ghost-to-active
    distance is Coords::Distance of each system."""
    from artemis_amd import capi
    bc = ("conductive",) * 6
    o, mb = pair(nx, lo, hi, ns_gas=1, ns_dust=0, seed=27, bc=bc, coordinates=coordinates)
    o.set_gravity_uniform(-0.3, 0.2, 0.1)
    ck = dict(cond=0.1) if ctype == "conductivity" else dict(kappa=0.1)
    o.set_conductivity(ctype, **ck)
    o.pgen_conduction(gas_rho=1.0, gas_temp=0.05, flux=0.01, post_init=False)  # sets the BC parameters
    random_state(o, np.random.default_rng(28), shock=False, contrast=10.0)
    push([o], mb)
    o.ApplyBoundaryConditions()
    mb.ApplyBoundaryConditions([bc], conductive=dict(
        temp=0.05, flux=0.01, g=(-0.3, 0.2, 0.1), coeff=0.1, cv=1.0 / ((1.4 - 1.0) * 1.0 * 1.0),
        type=capi.CONDUCTIVITY_PLAW if ctype == "conductivity" else capi.THERMALDIFF_PLAW))
    same(mb.gas_prim[0], o.gprim, "gas ghosts")
