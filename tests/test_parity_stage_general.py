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
    ((24, 10, 1), (0.4, 0.5, -0.5), (2.5, 2.6, 0.5), 1, 1, "plm", "hlle", "hlle", "spherical", 2),
    ((16, 8, 6), (0.3, 0.7, 0.0), (1.7, 2.5, 6.0), 2, 1, "plm", "hllc", "hlle", "spherical", 2),
    ((40, 1, 1), (0.0, 0.0, -0.5), (1.0, np.pi, 0.5), 1, 1, "ppm", "hlle", "hlle", "spherical", 3),
    ((16, 8, 6), (0.5, 0.0, -1.0), (2.0, 6.0, 1.0), 1, 2, "plm", "llf", "hlle", "cylindrical", 2),
    ((24, 12, 1), (0.0, -1.0, -0.5), (2.0, 1.0, 0.5), 1, 1, "plm", "hlle", "hlle", "axisymmetric", 2),
]


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
    I = (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    if nsg:
        out = mb._extra_prim["o"][0][0][I].cpu().numpy()
        ref = o.gprim[I]
        keep = [v for v in range(6 * nsg) if not (4 * nsg <= v < 5 * nsg)]  # P is not written
        assert np.array_equal(out[keep], ref[keep]), "gas prim"
    if nsd:
        same(dbuf[0][I], o.dprim[I], "dust prim")


@pytest.mark.parametrize("case", [1, 4, 5])
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
