"""Pin the CPU oracle against every known answer the reference's own regression tests hold
for the hot path.  The reference executable cannot be built here
(Parthenon/Kokkos/singularity-eos submodules are empty), so these numbers -- copied from the
reference's test scripts, cited per test -- are what anchors parity."""
import os

import numpy as np
import pytest

from oracle.oracle import Oracle
from pins import ADVECTION, DISK, DRAG, LANDINGS, LINWAVE, PINS, SSHEET, advection_history, linwave_waves

# every reference-held number below comes from tests/golden/reference_pins.json (cited there)
LW = dict(ng=LINWAVE["mesh"]["nghost"], gamma=LINWAVE["gamma"], cfl=LINWAVE["cfl"], bc=("periodic",) * 6,
          integrator=LINWAVE["integrator"])


def _linwave(N, recon, riem, wave, vflow):
    # tst/scripts/hydro/linwave.py:41-63 overrides on inputs/linwave/linear_wave.in
    o = Oracle((N, N // 2, N // 2), (0, 0, 0), (3.0, 1.5, 1.5), reconstruct=recon, riemann=riem, **LW)
    tlim = o.pgen_linear_wave(wave, LINWAVE["amp"], vflow)
    o.evolve(tlim, LINWAVE["nlim"])
    return o.linear_wave_errors()[0]


@pytest.mark.parametrize("recon", ["plm", "ppm"])
@pytest.mark.parametrize("riem", ["hllc", "hlle", "llf"])
def test_linwave_thresholds(recon, riem):
    # thresholds: tst/scripts/hydro/linwave.py:95-107
    err_thr, conv_thr = LINWAVE[recon]["rms_err_n32_max"], LINWAVE[recon]["n32_over_n16_max"]
    waves = linwave_waves()  # L-sound, R-sound, entropy (linwave.py:64-72)
    e32 = []
    for wi, (wave, vflow) in enumerate(waves):
        e16 = _linwave(16, recon, riem, wave, vflow)
        e = _linwave(32, recon, riem, wave, vflow)
        e32.append(e)
        assert e <= err_thr[wi], (recon, riem, wave, e)
        assert e / e16 <= conv_thr[wi], (recon, riem, wave, e / e16)
    # linwave.py:135-143 compares the L and R errors as parsed from the "%e" column of
    # -errs.dat (linear_wave.hpp:368): identical to the printed precision.
    assert "%e" % e32[0] == "%e" % e32[1]


def test_linwave_threshold_is_tight():
    """The reference's thresholds sit <1% above its own results (plm+hlle sound 2.23e-7,
    plm+llf entropy 2.21e-7): the oracle must land within that margin, not just below."""
    L = LANDINGS["linwave_n32"]
    assert L["plm_hlle_sound"][0] < _linwave(32, "plm", "hlle", 0, 0.0) <= L["plm_hlle_sound"][1] == LINWAVE["plm"]["rms_err_n32_max"][0]
    assert L["plm_llf_entropy"][0] < _linwave(32, "plm", "llf", 3, 1.0) <= L["plm_llf_entropy"][1] == LINWAVE["plm"]["rms_err_n32_max"][2]
    assert L["ppm_llf_sound"][0] < _linwave(32, "ppm", "llf", 0, 0.0) <= L["ppm_llf_sound"][1] == LINWAVE["ppm"]["rms_err_n32_max"][0]
    assert L["ppm_llf_entropy"][0] < _linwave(32, "ppm", "llf", 3, 1.0) <= L["ppm_llf_entropy"][1] == LINWAVE["ppm"]["rms_err_n32_max"][2]


def _advection(N, riem):
    # tst/scripts/advection/advection.py:44-68 on inputs/advection/advection.in
    o = Oracle((N, N // 2, N // 2), (0, 0, 0), (3.0, 1.5, 1.5), ns_gas=1, ns_dust=2,
               reconstruct="plm", riemann=riem, dust_reconstruct="plm", dust_riemann=riem,
               dust_cfl=0.9, **LW)
    tlim = o.pgen_advection(1.0e-6, 1.0)
    n = o.evolve(tlim, 1000)
    return o, n


@pytest.mark.parametrize("riem", ["hlle", "llf"])
def test_advection_history_and_errors(riem):
    def equiv(a, b, tol=ADVECTION["equiv_rel_tol"]):  # advection.py:95-99
        return 2.0 * abs(a - b) / (abs(a) + abs(b)) <= tol
    o16, _ = _advection(16, riem)
    o, n = _advection(32, riem)
    # advection.py:100-118 (history at t = 1 of the last run: llf, N = 32; hlle agrees to 1e-4)
    H = ADVECTION["history_n32"]
    assert n == H["cycle"]
    assert equiv(o.time, H["time"])
    assert equiv(o.dt, H["dt"])
    h = o.history()
    expected = advection_history()
    for got, exp in zip(h, expected):
        assert equiv(got, exp), (got, exp)
    # advection.py:137-178: plm thresholds for gas, dust1, dust2
    e16, e32 = o16.advection_errors(), o.advection_errors()
    for s in range(3):
        assert e32[s] <= ADVECTION["plm"]["rms_err_n32_max"]
        assert e32[s] / e16[s] <= ADVECTION["plm"]["n32_over_n16_max"]
    # advection.py:179-187: L- and R-going dust errors identical as printed
    assert "%e" % e32[1] == "%e" % e32[2]


def test_sedov_shock_radius_2d():
    """inputs/blast/blast.in as shipped (2-D Cartesian 256^2, hlle+plm, cylindrical blast)
    at reduced resolution: the pressure jump must sit at the Sedov radius
    r_s = xi0 (E t^2 / rho0)^(1/4), xi0 ~= 1.0 for gamma = 1.4 in 2-D planar symmetry
    (tst/scripts/coords/blast.py:118-183 checks the pressure profile against ExactPack with a
    loose L2 < 1 bound; here the front position is the sharper, table-free check)."""
    N = 128
    o = Oracle((N, N, 1), (-1, -1, -0.5), (1, 1, 0.5), ng=2, reconstruct="plm", riemann="hlle",
               gamma=1.4, dfloor=1e-10, siefloor=1e-10, cfl=0.3, bc=("outflow",) * 6,
               integrator="rk2")
    o.pgen_blast(radius=0.02, internal_energy=1.0, p0=1e-5, d0=1.0, samples=10,
                 symmetry="cylindrical")
    e0 = o.history()[4]
    o.evolve(0.1, -1)
    P = o.interior(o.gprim)[4, 0]
    x = -1 + (np.arange(N) + 0.5) * 2.0 / N
    X, Y = np.meshgrid(x, x)
    r = np.hypot(X, Y)
    r_peak = r.ravel()[np.argmax(P)]
    r_s = 1.0 * (1.0 * 0.1 ** 2 / 1.0) ** 0.25  # ~0.316
    assert abs(r_peak - r_s) < 0.03, (r_peak, r_s)
    # total energy is conserved to round-off while the shock is inside the box
    assert abs(e0 - 1.0) < 0.05  # sub-sampled deposit ~ internal_energy = 1
    assert abs(o.history()[4] - e0) < 1e-12 * e0


def test_leaf_edge_cases():
    from oracle import oracle as orc
    # plm.hpp:41: dq2 <= 0 -> zero slope; harmonic mean otherwise
    assert orc.plm(1.0, 2.0, 1.0) == (2.0, 2.0)
    assert orc.plm(1.0, 1.0, 1.0) == (1.0, 1.0)
    ql, qr = orc.plm(0.0, 1.0, 3.0)
    assert (ql, qr) == (1.0 + 2.0 / 3.0, 1.0 - 2.0 / 3.0)
    # ppm.hpp:49-52: local extremum -> flat
    assert orc.ppm4(0.0, 1.0, 2.0, 1.0, 0.0) == (2.0, 2.0)
    # hllc.hpp:112-113: identical states at rest -> zero mass flux, interface pressure = P
    w = [1.0, 0.0, 0.0, 0.0, 1.0, 1.5]
    out = orc.riemann(0, "hllc", 2.0 / 3.0, w, w)
    assert out[0] == 0.0 and out[6] == 1.0 and out[7] == 0.0
    # supersonic to the right: upwind flux = left flux; energy flux includes P*v
    wl = [1.0, 10.0, 0.0, 0.0, 1.0, 1.5]
    out = orc.riemann(0, "hllc", 2.0 / 3.0, wl, wl)
    assert out[0] == 10.0 and abs(out[4] - (1.5 + 50.0 + 1.0) * 10.0) < 1e-12


# ---- curvilinear geometry ------------------------------------
@pytest.mark.parametrize("g", ["sph", "cyl"])
def test_sedov_curvilinear_1d(g):
    """tst/scripts/coords/blast.py:36-80 `sph` (spherical1D, reflecting centre) and `cyl`
    (axisymmetric, nx2 = 1) at 256 instead of 1024 cells: the pressure peak sits at the Sedov
    radius of the deposited energy (3-D: xi0 = 1.033, 2-D: xi0 = 1.004 for gamma = 1.4) and the
    volume-integrated total energy (history.hpp:29-62 with the curvilinear cell volume) is
    conserved to round-off -- both fail for a wrong face area, volume or momentum scale factor.
    The reference bounds the same runs by a pressure L2 error < 1 against ExactPack tables."""
    N = 256
    sph = g == "sph"
    o = Oracle((N, 1, 1), (0.0, 0.0 if sph else -0.5, -0.5), (1.0, np.pi if sph else 0.5, 0.5), ng=2,
               reconstruct="plm", riemann="hlle", gamma=1.4, dfloor=1e-10, siefloor=1e-10, cfl=0.3,
               bc=(("reflecting" if sph else "outflow"),) + ("outflow",) * 5, integrator="rk2",
               coordinates="spherical" if sph else "axisymmetric")
    o.pgen_blast(radius=0.04, internal_energy=1.0, p0=1e-5, d0=1.0, samples=0,
                 symmetry="spherical" if sph else "cylindrical")
    h0 = o.history()
    o.evolve(0.1, -1)
    h1 = o.history()
    assert abs(h1[4] - h0[4]) < 1e-12 * h0[4] and abs(h1[0] - h0[0]) < 1e-12 * h0[0]
    assert abs(h0[0] - (1.0 / 3.0 if sph else 0.5)) < 1e-12  # int r^2 dr / int r dr with d0 = 1
    P = o.interior(o.gprim)[4, 0, 0]
    r = (np.arange(N) + 0.5) / N
    E = (4 * np.pi if sph else 2 * np.pi) * h0[4]
    r_s = 1.033 * (E * 0.01) ** 0.2 if sph else 1.004 * (E * 0.01) ** 0.25
    assert abs(r[np.argmax(P)] - r_s) < 0.015, (r[np.argmax(P)], r_s)
    # Rankine-Hugoniot post-shock pressure 2/(gamma+1) rho0 U^2, U = (2/5 | 1/2) r_s / t
    U = (0.4 if sph else 0.5) * r_s / 0.1
    assert 0.6 < P.max() / (2.0 / 2.4 * U * U) < 1.1


def test_spherical_2d_3d_reduce_to_1d():
    """A centred blast on spherical2D / spherical3D grids must reproduce the spherical1D radial
    profile: every theta / phi dependence of the face areas, volumes, scale factors
    (spherical.hpp:36-146, :240-345) and of ScaleMomentumFlux cancels for a radial flow."""
    kw = dict(ng=2, reconstruct="plm", riemann="hlle", gamma=1.4, dfloor=1e-10, siefloor=1e-10, cfl=0.3,
              integrator="rk2", coordinates="spherical")
    bc = ("reflecting", "outflow", "reflecting", "reflecting", "periodic", "periodic")
    runs = []
    for nx, hi in (((64, 1, 1), (1.0, np.pi, 0.5)), ((64, 6, 1), (1.0, 2.4, 0.5)), ((64, 6, 4), (1.0, 2.4, 2 * np.pi))):
        lo = (0.0, 0.0 if nx[1] == 1 else 0.7, -0.5 if nx[2] == 1 else 0.0)
        o = Oracle(nx, lo, hi, bc=bc, **kw)
        o.pgen_blast(radius=0.2, internal_energy=1.0, p0=1e-3, d0=1.0, samples=0)
        o.evolve(0.05, -1)
        runs.append(o.interior(o.gprim).copy())
    one = runs[0][:, 0, 0]
    for r in runs[1:]:
        # the multi-D runs take ~4x smaller steps (r*dtheta near the centre limits dt), so the
        # profiles agree to truncation error, not round-off; a wrong metric factor is an O(1) error
        for v in (0, 1, 4, 5):  # rho, v_r, P, sie
            assert np.max(np.abs(r[v] - one[v][None, None, :])) < 0.03 * np.max(np.abs(one[v])), v
        # and they stay exactly spherically symmetric: no v_theta / v_phi beyond round-off,
        # every ray identical to round-off
        assert np.max(np.abs(r[2])) < 1e-13 and np.max(np.abs(r[3])) < 1e-13
        assert np.max(np.abs(r - r[:, :1, :1, :])) < 1e-12
    assert np.max(np.abs(runs[2][:, 0] - runs[1][:, 0])) < 2e-3  # 3-D vs 2-D: same dt limit up to dphi


def test_curvilinear_static_equilibrium_and_geometry_identities():
    """(i) A uniform gas at rest stays bit-for-bit static in every coordinate system (the
    pressure-free momentum flux + dt/dx pressure difference + coordinate source must cancel
    exactly).  (ii) Summed cell volumes equal the closed-form volume of the domain."""
    cases = [("cylindrical", (12, 8, 4), (0.5, 0.0, -1.0), (2.0, 2 * np.pi, 1.0),
              lambda lo, hi: 0.5 * (hi[0] ** 2 - lo[0] ** 2) * (hi[1] - lo[1]) * (hi[2] - lo[2])),
             ("axisymmetric", (12, 8, 1), (0.0, -1.0, -0.5), (2.0, 1.0, 0.5),
              lambda lo, hi: 0.5 * (hi[0] ** 2 - lo[0] ** 2) * (hi[1] - lo[1]) * (hi[2] - lo[2])),
             ("spherical", (12, 8, 4), (0.3, 0.6, 0.0), (1.5, 2.5, 2.0),
              lambda lo, hi: (hi[0] ** 3 - lo[0] ** 3) / 3 * (np.cos(lo[1]) - np.cos(hi[1])) * (hi[2] - lo[2])),
             ("spherical", (12, 8, 1), (0.3, 0.6, -0.5), (1.5, 2.5, 0.5),
              lambda lo, hi: (hi[0] ** 3 - lo[0] ** 3) / 3 * (np.cos(lo[1]) - np.cos(hi[1]))),
             ("spherical", (12, 1, 1), (0.3, 0.0, -0.5), (1.5, np.pi, 0.5),
              lambda lo, hi: (hi[0] ** 3 - lo[0] ** 3) / 3)]
    for sys_, nx, lo, hi, vol in cases:
        o = Oracle(nx, lo, hi, ng=2, reconstruct="plm", riemann="hllc", gamma=1.4, cfl=0.3,
                   bc=("outflow",) * 6, coordinates=sys_)
        o.pgen_blast(radius=1e-9, internal_energy=1.0, p0=0.7, d0=1.3, samples=0)
        before = o.gprim.copy()
        assert abs(o.history()[0] - 1.3 * vol(lo, hi)) < 1e-12 * vol(lo, hi), sys_
        o.evolve(-1.0, 5)
        assert np.array_equal(o.gprim, before), sys_


# ---- source packages ---------------------------------------------------------
def test_drag_reference_test_pins():
    """tst/scripts/drag/drag.py:36-37,57-59,127-129 on inputs/drag/simple_drag.in: for every
    output time up to t = 10 and each of the four stopping times, |<v_dust - v_gas> - ans| <= 3e-3
    with ans = -exp(-(1 + 0.01/10) t / tau), and total momentum conserved to 1e-13."""
    tau = DRAG["tau"]
    o = Oracle((128, 1, 1), (0.0, -0.5, -0.5), (1.0, 0.5, 0.5), ng=2, ns_gas=1, ns_dust=4,
               reconstruct="plm", riemann="hlle", dust_reconstruct="plm", dust_riemann="hlle", gamma=1.4,
               dfloor=1e-10, siefloor=1e-10, dust_dfloor=1e-10, cfl=0.3, dust_cfl=0.3,
               bc=("periodic",) * 6, integrator="rk2")
    o.set_drag("simple_dust", "constant", tau=tau)
    o.pgen_constant(gas_rho=10.0, gas_v=(1.0, 0, 0), gas_temp=1.0, dust_rho=0.01, dust_v=(0, 0, 0))
    mom = lambda h: h[1] + sum(h[7 + 4 * n] for n in range(4))
    m0 = mom(o.history())
    c = 0.01 / 10.0
    worst, worst_mom = 0.0, 0.0
    for tout in np.arange(0.05, 10.0 - 1e-9, 0.05):  # drag.py:61: outputs 1 .. tlim/0.05 - 1
        o.evolve(tout, -1)
        vg = o.interior(o.gprim)[1, 0, 0]
        dp = o.interior(o.dprim)
        for d in range(4):
            ans = -np.exp(-(1.0 + c) * o.time / tau[d])
            worst = max(worst, abs((dp[4 + 3 * d, 0, 0] - vg).mean() - ans))
        worst_mom = max(worst_mom, abs(mom(o.history()) / m0 - 1))
    assert worst <= DRAG["velocity_difference_tol"], worst  # the oracle gives 2.43e-3: the threshold is a 25 % margin
    assert abs(worst - LANDINGS["drag_worst"]) < 2e-5  # ... and not vacuous
    assert worst_mom <= DRAG["momentum_tol"], worst_mom


def test_shearing_sheet_reference_test_pins():
    """tst/scripts/ssheet/ssheet.py:40-128 on inputs/ssheet/ssheet.in to t = 2 pi: the density
    wake of the embedded point mass, located as the maximum of Sigma - <Sigma>_y on the rings
    x = -0.1 (last face <= -0.1) and x = +0.1 (first centre >= 0.1), sits within 0.03 of the
    Ogilvie & Lubow position y = -+ 3/4 x^2 / h.  Exercises the strat pgen, point-mass gravity,
    the shearing-box source and the extrap / inflow user boundary conditions together."""
    N = 128
    o = Oracle((N, N, 1), (-1.0, -1.0, -0.2), (1.0, 1.0, 0.2), ng=2, reconstruct="plm", riemann="hllc",
               gamma=1.000001, dfloor=1e-10, siefloor=1e-10, cfl=0.3,
               bc=("extrap", "extrap", "inflow", "inflow", "extrap", "extrap"), integrator="rk2")
    o.set_rotating_frame(1.0, 1.5)
    o.set_gravity_point(1e-5, soft=0.03)
    o.pgen_strat(rho0=1.0, dens_min=1e-10, h=0.05)
    o.evolve(2.0 * np.pi, -1)
    d = o.interior(o.gprim)[0, 0]  # [j, i]
    x = np.linspace(-1, 1, N + 1)
    xc = 0.5 * (x[1:] + x[:-1])
    sig = d - d.mean(axis=0)[None, :]
    ii = np.argwhere(x <= -0.1)[-1][0]
    io = np.argwhere(xc >= 0.1)[0][0]
    pi_ = xc[np.argmax(sig[:, ii])]
    po = xc[np.argmax(sig[:, io])]
    h, tol = SSHEET["h"], SSHEET["wake_position_tol"]
    assert abs(pi_ - 0.75 * SSHEET["ring_x"] ** 2 / h) < tol, pi_
    assert abs(po + 0.75 * SSHEET["ring_x"] ** 2 / h) < tol, po
    assert 0.01 < sig.max() < 1.0  # a wake exists and the sheet has not blown up


@pytest.mark.parametrize("d", [1, 2])
def test_viscous_diffusion_reference_test_pins(d):
    """tst/scripts/diffusion/viscous_diffusion.py:36-46,95-150 on inputs/diffusion/gaussian_bump.in:
    a v3 Gaussian of width sqrt(2 nu t0) diffuses under nu = 0.25 for t = 2; the mean absolute
    deviation from eps (2 pi s2)^(-d/2) exp(-r^2 / 2 s2), s2 = 2 nu (t + t0), must stay below 1e-8
    (eps = 1e-6) in 1-D and 2-D."""
    nu, t0, eps, tlim, nx = 0.25, 0.5, 1e-6, 2.0, 64
    sig2 = 2.0 * nu * t0
    n = (nx, 1, 1) if d == 1 else (nx, nx, 1)
    o = Oracle(n, (-6.0, -6.0, -0.5), (6.0, 6.0, 0.5), ng=2, reconstruct="plm", riemann="hllc",
               gamma=1.000001, dfloor=1e-10, siefloor=1e-10, cfl=0.3,
               bc=("outflow",) * 4 + ("periodic",) * 2, integrator="rk2")
    o.set_viscosity("constant", nu=nu)
    o.pgen_gaussian_bump(sigma=np.sqrt(sig2), v_bump=(0.0, 0.0, eps * (2.0 * np.pi * sig2) ** (-0.5 * d)))
    o.evolve(tlim, -1)
    w = o.interior(o.gprim)[3, 0]
    xc = -6.0 + (np.arange(nx) + 0.5) * 12.0 / nx
    s2 = 2.0 * nu * (o.time + t0)
    if d == 1:
        ans = eps * (2.0 * np.pi * s2) ** (-0.5 * d) * np.exp(-xc ** 2 / (2.0 * s2))
        err = np.abs(ans - w[0]).mean()
    else:
        yy, xx = np.meshgrid(xc, xc)
        ans = eps * (2.0 * np.pi * s2) ** (-0.5 * d) * np.exp(-(xx ** 2 + yy ** 2) / (2.0 * s2))
        err = np.abs(ans.ravel() - w.T.ravel()).mean()
    assert err <= PINS["viscous_diffusion"]["mean_abs_err_max"], err   # the oracle gives 2.2e-10 (1-D) and 2.6e-11 (2-D)
    assert err > 1e-13        # a second-order scheme on 64 zones is not exact either


def _conduction_oracle(g, nx):
    x2 = (np.pi / 2 - 0.5, np.pi / 2 + 0.5) if g == "spherical" else (-0.5, 0.5)
    o = Oracle((nx, 1, 1), (0.2, x2[0], -0.5), (1.2, x2[1], 0.5), ng=2, reconstruct="plm", riemann="hllc",
               gamma=1.66667, dfloor=1e-10, siefloor=1e-15, cfl=0.3,
               bc=("conductive", "conductive") + ("periodic",) * 4, integrator="rk2", coordinates=g)
    o.set_gravity_uniform(0.0, 0.0, 0.0)
    o.set_conductivity("conductivity", cond=0.1)
    o.set_drag("self", "constant")
    o.set_damping(0, inner=(4.0, -1.7976931348623157e308, -1.7976931348623157e308), inner_rate=(1.0e4, 0.0, 0.0))
    o.pgen_conduction(gas_rho=1.0, gas_temp=0.05, flux=0.01)
    return o


def _conduction_answer(x, d, f=0.01, T0=0.05, x0=1.2, xi=0.2, k=0.1):
    # steady state of div(K grad T) = 0 with flux f through x = xi and T(x0) = T0 in slab (d = 0),
    # cylindrical (1) and spherical (2) geometry -- thermal_diffusion.py:72-79
    f = f * xi ** d
    return (T0 + (x - x0) * -f / k, T0 + np.log(x / x0) * -f / k, T0 + (1.0 / x - 1.0 / x0) * f / k)[d]


@pytest.mark.parametrize("g,d,e64,e128", [(g, d, LANDINGS["thermal_diffusion_n64"][g], LANDINGS["thermal_diffusion_n128"][g])
                                          for d, g in enumerate(("cartesian", "axisymmetric", "spherical"))])
def test_thermal_diffusion_reference_test_pins(g, d, e64, e128):
    """tst/scripts/diffusion/thermal_diffusion.py:36-70,99-125 on inputs/diffusion/conduction.in in
    Cartesian, axisymmetric and spherical coordinates (conduction pgen, `conductive` boundary
    conditions, heat conduction with the curvilinear face areas / volumes / distances): mean
    |T/T_analytic - 1| <= 5e-3 at t = 50.  The reference runs 128 zones (372 529 cycles, about a
    minute of oracle time per geometry); the oracle gives 4.11e-3 / 1.04e-3 / 1.91e-4 there (checked
    once, numbers below) and exactly twice that at 64 zones -- the error is the first-order
    boundary closure -- which is what this test runs."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    nx = 64
    o = _conduction_oracle(g, nx)
    o.evolve(50.0, -1)
    T = o.interior(o.gprim)[5, 0, 0] * (1.66667 - 1.0)  # T = sie / cv, cv = 1/(gamma - 1)
    xc = 0.2 + (np.arange(nx) + 0.5) / nx
    err = np.abs(T / _conduction_answer(xc, d) - 1.0).mean()
    assert abs(err - e64) < 0.01 * e64, err
    assert abs(err / 2.0 - e128) < 0.05 * e128 and e128 <= PINS["thermal_diffusion"]["mean_rel_err_max"]  # the 128-zone value and the reference bound


_PI = 3.141592653589793
_DISK_DECKS = {  # inputs/disk/disk_{sph,cyl,axi}.in: mesh, boundary faces carrying the user condition, solver
    "sph": dict(nx=(128, 64, 64), lo=(0.2, 1.059856161608513, -_PI), hi=(5.6, 2.081736491981280, _PI), riemann="hlle",
                bc=lambda b: (b, b, b, b, "periodic", "periodic"), coordinates="spherical", siefloor=1e-20),
    "cyl": dict(nx=(128, 64, 32), lo=(0.3, -_PI, -1.0), hi=(4.3, _PI, 1.0), riemann="hllc",
                bc=lambda b: (b, b, "periodic", "periodic", b, b), coordinates="cylindrical", siefloor=1e-10),
    "axi": dict(nx=(128, 64, 1), lo=(0.3, -2.0, -0.5), hi=(4.3, 2.0, 0.5), riemann="hlle",
                bc=lambda b: (b, b, b, b, "periodic", "periodic"), coordinates="axisymmetric", siefloor=1e-20),
}


def disk_oracle(g, gam, b, nx=None):
    D = _DISK_DECKS[g]
    o = Oracle(nx or D["nx"], D["lo"], D["hi"], ng=2, reconstruct="plm", riemann=D["riemann"], gamma=1.4,
               dfloor=1e-10, siefloor=D["siefloor"], cfl=0.3, integrator="rk2", coordinates=D["coordinates"],
               bc=D["bc"]("ic" if b == "ic" else "disk_extrap"), de_switch=1e-2 if g == "sph" else 0.0)
    o.set_gravity_point(mass=1.0)
    o.set_rotating_frame(1.0, 0.0)
    o.set_viscosity("alpha", alpha=1e-3, r0=1.0, Omega0=1.0)
    o.pgen_disk(r0=1.0, rho0=1.0, dslope=-2.25, flare=0.25, h0=0.05, dens_min=1e-10, pres_min=1e-15,
                polytropic_index=gam)
    return o


@pytest.mark.parametrize("g,gam,b,err_ref,dt_ref", [tuple(c) for c in LANDINGS["disk"]])
def test_disk_reference_test_pins(g, gam, b, err_ref, dt_ref):
    """tst/scripts/disk/disk.py:36-45,58-96,118-187 on the shipped uniform-mesh decks
    inputs/disk/disk_{axi,cyl,sph}.in (disk pgen, `ic` / `extrap` user conditions, point-mass
    gravity, alpha viscosity, rotating frame in its angular-momentum-conserving flux form): after
    the test's 10 cycles (5 + 5 across its restart) no NaN, positive density and temperature,
    1e-4 < dt < 3e-2 and density error sqrt(sum d0 (d-d0)^2)/sum d0 <= 6e-3.  The oracle's values
    (all twelve geometry x polytropic index x condition cases were run once: axi 5.4e-3 / 4.3e-3,
    cyl 1.3e-4 .. 1.7e-4, sph 4.6e-4 .. 5.8e-4) sit under the reference's tolerance, the axisymmetric
    ones (whose mesh resolves the vertical structure over +-40 scale heights) within 10 % of it.  The Cartesian deck is
    statically refined (SMR) and out of scope."""
    o = disk_oracle(g, gam, b)
    d0 = o.interior(o.gprim)[0].copy()
    o.evolve(62.8, DISK["cycles"])
    P = o.interior(o.gprim)
    d, T = P[0], P[5] * 0.4
    assert o.ncycle == DISK["cycles"] and not np.isnan(P).any() and d.min() > 0.0 and T.min() > 0.0
    assert DISK["dt_low"] < o.dt < DISK["dt_high"] and abs(o.dt - dt_ref) < 2e-4 * dt_ref
    err = np.sqrt((d0 * (d - d0) ** 2).sum()) / d0.sum()
    assert err <= DISK["density_err_max"] and abs(err - err_ref) < 2e-3 * err_ref, err


def alpha_disk_oracle(nx=64, alpha=0.1, h=0.1):
    """inputs/diffusion/alpha_disk.in with the overrides of tst/scripts/diffusion/alpha_disk.py:44-75"""
    o = Oracle((nx, 1, 1), (0.3, -0.5, -0.5), (2.0, 0.5, 0.5), ng=2, reconstruct="plm", riemann="hllc",
               gamma=1.0001, dfloor=1e-10, siefloor=1e-15, cfl=0.3, integrator="rk2",
               coordinates="axisymmetric", bc=("viscous", "viscous") + ("periodic",) * 4)
    o.set_gravity_point(mass=1.0)
    o.set_viscosity("alpha", alpha=alpha, r0=1.0, Omega0=1.0)
    # the script passes its numbers as "{:.8e}" command-line overrides (alpha_disk.py:52-61)
    o.set_cooling(beta0=0.0, tcyl=float(f"{h ** 2:.8e}"), cyl_plaw=-1.0)
    o.pgen_disk(r0=1.0, dslope=0.0, flare=0.0, h0=h, dens_min=1e-10, pres_min=1e-15, polytropic_index=1.0,
                mdot=float(f"{alpha * h ** 2 * 3 * np.pi:.8e}"), quiet_start=True)
    return o


@pytest.mark.usefixtures("one_openmp_thread")
def test_alpha_disk_reference_test_pin():
    """tst/scripts/diffusion/alpha_disk.py:36-75,82-139: a 1-D axisymmetric alpha disk (alpha = 0.1,
    h = 0.1, locally isothermal through beta cooling with beta0 = 0, `viscous` inflow / outflow
    conditions with the steady accretion rate mdot = 3 pi alpha h^2) relaxes by t = 8000 to
    Sigma = R^-1/2 and a radially constant accretion rate: mean relative errors of both <= 2e-3.
    The oracle gives 8.8e-4 and 1.81e-3 (181 938 cycles, half a minute) -- again just inside a
    tolerance evidently set from the reference's own output.  Exercises the disk pgen, alpha viscosity,
    point-mass gravity, BetaCooling and DiskBoundaryVisc together."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    o = alpha_disk_oracle()
    o.evolve(8e3, -1)
    P = o.interior(o.gprim)
    r = 0.3 + (np.arange(64) + 0.5) * 1.7 / 64
    dens, u = P[0, 0, 0], P[1, 0, 0]
    mdot = -2 * np.pi * r * dens * u
    e_d = np.abs((1.0 / np.sqrt(r) - dens) * np.sqrt(r)).mean()
    e_m = np.abs((3 * np.pi * 0.1 * 0.1 ** 2 - mdot) / (3 * np.pi * 0.1 * 0.1 ** 2)).mean()
    assert abs(o.time - 8e3) < 1e-9 and o.ncycle > 150000
    tol = PINS["alpha_disk"]["mean_rel_err_max"]
    assert e_d <= tol and e_m <= tol, (e_d, e_m)
    assert abs(e_d - LANDINGS["alpha_disk"]["density"]) < 2e-5 and abs(e_m - LANDINGS["alpha_disk"]["mdot"]) < 2e-5, (e_d, e_m)
