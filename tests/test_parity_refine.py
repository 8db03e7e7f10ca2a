"""GPU parity of the mesh-refinement data-path operators (the operators only):
ArtemisUtils::RestrictAverage<GEOM> (utils/refinement/restriction.hpp:42-114) and
ProlongateSharedMinMod<GEOM> (utils/refinement/prolongation.hpp:83-184) against the CPU oracle, BIT-EXACT,
in every coordinate system, plus the properties the operators must have (the reference holds no test
that isolates them: parity is pinned on these properties only)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle.oracle import Oracle
from test_parity_ops import push, random_state, same

BLOCKS = [
    ("cartesian", (12, 8, 6), (-1.0, -0.5, 0.25), (1.0, 0.8, 0.95)),
    ("cartesian", (16, 10, 1), (-1.0, -0.5, -0.5), (1.0, 0.8, 0.5)),
    ("cartesian", (24, 1, 1), (0.0, -0.5, -0.5), (1.0, 0.5, 0.5)),
    ("spherical", (12, 8, 6), (0.4, 0.7, 0.0), (1.7, 2.5, 6.0)),
    ("spherical", (12, 8, 1), (0.4, 0.7, -0.5), (2.5, 2.4, 0.5)),
    ("spherical", (20, 1, 1), (0.3, 0.0, -0.5), (1.0, np.pi, 0.5)),
    ("cylindrical", (12, 8, 6), (0.5, 0.0, -1.0), (2.0, 6.0, 1.0)),
    ("axisymmetric", (12, 8, 1), (0.3, -1.0, -0.5), (2.0, 1.0, 0.5)),
]


def meshes(coordinates, nxc, lo, hi, gpu=True):
    """a coarse mesh and the fine mesh of twice the resolution over the same domain"""
    nxf = tuple(2 * n if n > 1 else 1 for n in nxc)
    kw = dict(ng=2, ns_gas=1, ns_dust=0, gamma=1.4, coordinates=coordinates)
    oc, of = Oracle(nxc, lo, hi, **kw), Oracle(nxf, lo, hi, **kw)
    rng = np.random.default_rng(17)
    random_state(oc, rng, shock=False), random_state(of, rng, shock=False)
    if not gpu:
        return oc, of, None, None
    from artemis_amd.pack import MeshBlockPack
    mc, mf = MeshBlockPack(1, nxc, [lo], [hi], **kw), MeshBlockPack(1, nxf, [lo], [hi], **kw)
    push([oc], mc), push([of], mf)
    return oc, of, mc, mf


def descriptor(oc, of, mc, mf, grow=0):
    from artemis_amd import capi
    r = capi.Refine()
    r.coords, r.ndim, r.nvar = mc.pack.coords, oc.ndim, 6
    r.fni, r.fnj, r.fnk, r.cni, r.cnj, r.cnk = of.ni, of.nj, of.nk, oc.ni, oc.nj, oc.nk
    r.fgeom, r.cgeom = mf.geom.data_ptr(), mc.geom.data_ptr()
    r.fmetric = mf.pack.metric
    r.cmetric = mc.pack.metric
    r.fine, r.coarse = mf.gas_prim_table, mc.gas_prim_table
    r.cis, r.cie, r.cjs, r.cje, r.cks, r.cke = oc.is_, oc.ie, oc.js, oc.je, oc.ks, oc.ke
    r.cib, r.cjb, r.ckb, r.fib, r.fjb, r.fkb = oc.is_, oc.js, oc.ks, of.is_, of.js, of.ks
    return r


def ranges(oc, of):
    return (oc.is_, oc.ie, oc.js, oc.je, oc.ks, oc.ke), (oc.is_, oc.js, oc.ks), (of.is_, of.js, of.ks)


@pytest.mark.gpu
@pytest.mark.parametrize("coordinates,nxc,lo,hi", BLOCKS)
def test_restrict_average(hiplib, coordinates, nxc, lo, hi):
    from artemis_amd import capi
    oc, of, mc, mf = meshes(coordinates, nxc, lo, hi)
    of.RestrictAverage(oc, *ranges(oc, of))
    capi.check(mf.L.artemis_hip_restrict_average(C.byref(descriptor(oc, of, mc, mf)), None))
    torch.cuda.synchronize()
    same(mc.gas_prim[0], oc.gprim, "restricted field (entire coarse array: ghosts untouched)")


@pytest.mark.gpu
@pytest.mark.parametrize("coordinates,nxc,lo,hi", BLOCKS)
def test_prolongate_shared_minmod(hiplib, coordinates, nxc, lo, hi):
    from artemis_amd import capi
    oc, of, mc, mf = meshes(coordinates, nxc, lo, hi)
    of.ProlongateSharedMinMod(oc, *ranges(oc, of))
    capi.check(mf.L.artemis_hip_prolongate_minmod(C.byref(descriptor(oc, of, mc, mf)), None))
    torch.cuda.synchronize()
    same(mf.gas_prim[0], of.gprim, "prolongated field (entire fine array: ghosts untouched)")


@pytest.mark.gpu
def test_refine_abi_contract(hiplib):
    from artemis_amd import capi
    oc, of, mc, mf = meshes("spherical", (8, 6, 4), (0.4, 0.7, 0.0), (1.7, 2.5, 6.0))
    r = descriptor(oc, of, mc, mf)
    r.cie = oc.ni - 1  # prolongation would read beyond the coarse array
    assert mf.L.artemis_hip_prolongate_minmod(C.byref(r), None) == capi.EINVAL
    r = descriptor(oc, of, mc, mf)
    r.cmetric = None
    assert mf.L.artemis_hip_restrict_average(C.byref(r), None) == capi.EINVAL
    assert b"metric" in mf.L.artemis_hip_last_error()


@pytest.mark.parametrize("coordinates,nxc,lo,hi", BLOCKS)
def test_operator_properties_on_the_oracle(coordinates, nxc, lo, hi):
    """What the reference's operators guarantee by construction: restriction preserves constants and the
    volume integral; prolongation preserves constants, is exact for linear data in Cartesian
    coordinates, never creates new extrema, and restriction undoes it on a Cartesian mesh."""
    oc, of, _, _ = meshes(coordinates, nxc, lo, hi, gpu=False)
    I = lambda o: (slice(None), slice(o.ks, o.ke + 1), slice(o.js, o.je + 1), slice(o.is_, o.ie + 1))
    rr = ranges(oc, of)
    of.gprim[:] = 3.25
    of.RestrictAverage(oc, *rr)
    assert np.max(np.abs(oc.gprim[I(oc)] - 3.25)) < 4e-15  # sum(V q) / sum(V): exact to an ulp or two
    oc.gprim[:] = -1.5
    of.ProlongateSharedMinMod(oc, *rr)
    assert np.array_equal(of.gprim[I(of)], np.full_like(of.gprim[I(of)], -1.5))
    rng = np.random.default_rng(5)
    oc.gprim[:] = rng.uniform(0.5, 2.0, oc.gprim.shape)
    coarse = oc.gprim.copy()
    of.ProlongateSharedMinMod(oc, *rr)
    fine = of.gprim[I(of)]
    # each fine value lies between the extrema of its parent's 3^ndim neighbourhood (minmod: monotone)
    assert fine.min() >= coarse.min() - 1e-14 and fine.max() <= coarse.max() + 1e-14
    if coordinates == "cartesian":
        of.RestrictAverage(oc, *rr)
        assert np.max(np.abs(oc.gprim[I(oc)] - coarse[I(oc)])) < 1e-14  # symmetric offsets cancel
        # linear data are reproduced exactly (to round-off)
        x = [(lo[d] + (np.arange(n + (4 if n > 1 else 0)) - (2 if n > 1 else 0) + 0.5) * (hi[d] - lo[d]) / n) for d, n in enumerate(nxc)]
        lin = 0.3 + 1.7 * x[0][None, None, :] - 0.6 * x[1][None, :, None] * (oc.ndim > 1) + 0.9 * x[2][:, None, None] * (oc.ndim > 2)
        oc.gprim[:] = lin[None]
        of.ProlongateSharedMinMod(oc, *rr)
        nf = tuple(2 * n if n > 1 else 1 for n in nxc)
        xf = [(lo[d] + (np.arange(n) + 0.5) * (hi[d] - lo[d]) / n) for d, n in enumerate(nf)]
        linf = 0.3 + 1.7 * xf[0][None, None, :] - 0.6 * xf[1][None, :, None] * (oc.ndim > 1) + 0.9 * xf[2][:, None, None] * (oc.ndim > 2)
        assert np.max(np.abs(of.gprim[I(of)][0] - linf)) < 1e-13
    # volume integral: sum(V_f q_f) over the children == V_c q_c (checked through the mass history integral)
    of.gprim[:] = rng.uniform(0.5, 2.0, of.gprim.shape)
    of.PrimToCons()
    mass_fine = of.history()[0]
    of.RestrictAverage(oc, *rr)
    oc.PrimToCons()
    assert abs(oc.history()[0] - mass_fine) < 1e-12 * mass_fine


# ---- refinement criteria (utils/refinement/amr_criteria.hpp) ------------------------------------------
def criterion(o, m, field, thr, deref=0.0):
    from artemis_amd import capi
    a = capi.AmrCriterion()
    a.coords, a.ndim = m.pack.coords, o.ndim
    a.ni, a.nj, a.nk = o.ni, o.nj, o.nk
    a.geom, a.metric, a.field = m.geom.data_ptr(), m.pack.metric, field.data_ptr()
    a.is_, a.ie, a.js, a.je, a.ks, a.ke = o.is_, o.ie, o.js, o.je, o.ks, o.ke
    a.refine_thr, a.deref_thr = thr, deref
    scratch = torch.full((1,), -7.0, dtype=torch.float64, device="cuda")
    a.scratch = scratch.data_ptr()
    return a, scratch


@pytest.mark.gpu
@pytest.mark.parametrize("coordinates,nxc,lo,hi", BLOCKS)
@pytest.mark.parametrize("var", [0, -1])
def test_amr_criteria(hiplib, coordinates, nxc, lo, hi, var):
    """ScalarFirstDerivative<FIELD, GEOM> and ScalarMagnitude<FIELD> on the gas density / pressure: the
    block maximum is BIT-EXACT against the oracle (a maximum has no summation order), and the AmrTag
    follows it through all three outcomes."""
    from artemis_amd import capi
    oc, _, mc, _ = meshes(coordinates, nxc, lo, hi)
    prim = mc.gas_prim[0]
    field = prim[0].contiguous() if var == 0 else prim[4].contiguous()  # FIELD = gas.prim.pressure: the stored array
    tag, m = C.c_int(9), C.c_double(-1.0)
    ref_tag, ref_max = oc.ScalarFirstDerivative(var, 1.0)
    for thr in ([1.0] if oc.ndim == 1 else [ref_max * 0.5, ref_max * 2.0, ref_max * 8.0]):
        a, keep = criterion(oc, mc, field, thr)
        capi.check(mc.L.artemis_hip_amr_first_derivative(C.byref(a), C.byref(tag), C.byref(m), None))
        want_tag, want = oc.ScalarFirstDerivative(var, thr)
        assert m.value == want and tag.value == want_tag, (thr, m.value, want, tag.value, want_tag)
    if oc.ndim > 1:
        assert ref_max > 0.0 and [oc.ScalarFirstDerivative(var, ref_max * f)[0] for f in (0.5, 2.0, 8.0)] == [1, 0, -1]
    else:
        assert (tag.value, m.value) == (0, 0.0)  # 1-D: AmrTag::same without computing (amr_criteria.hpp:122)
    _, qmax = oc.ScalarMagnitude(var, 1.0, 0.0)
    for above, below in [(qmax * 0.5, 0.0), (qmax * 2.0, qmax * 0.5), (qmax * 4.0, qmax * 2.0)]:
        a, keep = criterion(oc, mc, field, above, below)
        capi.check(mc.L.artemis_hip_amr_magnitude(C.byref(a), C.byref(tag), C.byref(m), None))
        want_tag, want = oc.ScalarMagnitude(var, above, below)
        assert m.value == want and tag.value == want_tag, (above, below, m.value, want)
    assert [oc.ScalarMagnitude(var, qmax * f, qmax * g)[0] for f, g in ((0.5, 0.0), (2.0, 0.5), (4.0, 2.0))] == [1, 0, -1]


@pytest.mark.gpu
def test_amr_criteria_abi_contract(hiplib):
    from artemis_amd import capi
    oc, _, mc, _ = meshes("spherical", (8, 6, 4), (0.4, 0.7, 0.0), (1.7, 2.5, 6.0))
    field = mc.gas_prim[0][0].contiguous()
    tag, m = C.c_int(0), C.c_double(0.0)
    a, keep = criterion(oc, mc, field, 1.0)
    a.ie = oc.ni - 2  # the grown range would read beyond the array
    assert mc.L.artemis_hip_amr_first_derivative(C.byref(a), C.byref(tag), C.byref(m), None) == capi.EINVAL
    a, keep = criterion(oc, mc, field, 1.0)
    a.metric = None
    assert mc.L.artemis_hip_amr_first_derivative(C.byref(a), C.byref(tag), C.byref(m), None) == capi.EINVAL
    assert mc.L.artemis_hip_amr_first_derivative(C.byref(a), None, C.byref(m), None) == capi.EINVAL


def test_amr_criteria_properties_on_the_oracle():
    """A uniform field never asks for refinement by gradient; a field with a jump does, and the criterion is
    invariant under a rescaling of the field (it is normalised by the local value)."""
    oc, _, _, _ = meshes("cartesian", (16, 10, 1), (-1.0, -0.5, -0.5), (1.0, 0.8, 0.5), gpu=False)
    oc.gprim[0] = 2.5
    assert oc.ScalarFirstDerivative(0, 0.1) == (-1, 0.0)
    oc.gprim[0][:, :, oc.ni // 2:] = 5.0
    tag, eps = oc.ScalarFirstDerivative(0, 0.1)
    assert tag == 1 and eps > 0.1
    oc.gprim[0] *= 8.0
    assert oc.ScalarFirstDerivative(0, 0.1) == (tag, eps)  # a power of two: exact
    assert oc.ScalarMagnitude(0, 30.0, 10.0) == (1, 40.0) and oc.ScalarMagnitude(0, 50.0, 45.0) == (-1, 40.0)
