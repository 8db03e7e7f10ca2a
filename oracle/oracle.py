"""ctypes binding for the CPU oracle (oracle/artemis_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under artemis_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")

RS = {"hllc": 0, "hlle": 1, "llf": 2}
RC = {"pcm": 0, "plm": 1, "ppm": 2}
BC = {"periodic": 0, "outflow": 1, "reflecting": 2, "reflect": 2, "none": 3,
      "extrap": 4, "inflow": 5, "conductive": 6, "ic": 7, "disk_extrap": 8, "viscous": 9}  # 4/5: the strat pgen's user conditions (problem_modifier.hpp:114-128)
INTEG = {"rk1": 0, "rk2": 1, "vl2": 2, "rk3": 3}
GAS, DUST = 0, 1

# field ids of oracle_field()
F_GPRIM, F_GU0, F_GU1 = 0, 1, 2
F_GFLUX, F_GPFLUX, F_GVFACE = 3, 6, 9
F_DPRIM, F_DU0, F_DU1, F_DFLUX = 12, 13, 14, 15


class Cfg(C.Structure):
    _fields_ = [
        ("nx1", C.c_int), ("nx2", C.c_int), ("nx3", C.c_int), ("ng", C.c_int),
        ("x1min", C.c_double), ("x1max", C.c_double), ("x2min", C.c_double),
        ("x2max", C.c_double), ("x3min", C.c_double), ("x3max", C.c_double),
        ("ns_gas", C.c_int), ("ns_dust", C.c_int),
        ("recon_gas", C.c_int), ("riemann_gas", C.c_int),
        ("recon_dust", C.c_int), ("riemann_dust", C.c_int),
        ("gamma", C.c_double), ("dfloor_gas", C.c_double), ("siefloor_gas", C.c_double),
        ("de_switch", C.c_double), ("dfloor_dust", C.c_double),
        ("cfl_gas", C.c_double), ("cfl_dust", C.c_double),
        ("bc", C.c_int * 6), ("integrator", C.c_int), ("nthreads", C.c_int),
        ("coords", C.c_int),
    ]


COORDS = {"cartesian": 0, "cylindrical": 1, "spherical1D": 2, "spherical2D": 3,
          "spherical3D": 4, "axisymmetric": 5}


def coord_select(sys, ndim):
    """geometry.hpp:38-56 CoordSelect: `spherical` picks its variant from the problem dimension."""
    if sys == "spherical":
        return (2, 2, 3, 4)[ndim]
    return COORDS[sys]


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(
            os.path.join(_HERE, "artemis_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(Cfg)]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_field.restype = C.POINTER(C.c_double)
        L.oracle_field.argtypes = [C.c_void_p, C.c_int]
        L.oracle_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.oracle_set_mesh_bounds.argtypes = [C.c_void_p] + [C.c_double] * 6
        L.oracle_calculate_fluxes.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_apply_update.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.oracle_flux_source.argtypes = [C.c_void_p, C.c_int, C.c_double]
        for f in ("set_aux", "cons_to_prim", "prim_to_cons", "deep_copy", "apply_bcs"):
            getattr(L, "oracle_" + f).argtypes = [C.c_void_p]
        L.oracle_estimate_dt.restype = C.c_double
        L.oracle_estimate_dt.argtypes = [C.c_void_p, C.c_int]
        L.oracle_new_dt.restype = C.c_double
        L.oracle_new_dt.argtypes = [C.c_void_p]
        L.oracle_time.restype = C.c_double
        L.oracle_time.argtypes = [C.c_void_p]
        L.oracle_dt.restype = C.c_double
        L.oracle_dt.argtypes = [C.c_void_p]
        L.oracle_ncycle.restype = C.c_long
        L.oracle_ncycle.argtypes = [C.c_void_p]
        L.oracle_set_dt.argtypes = [C.c_void_p, C.c_double]
        L.oracle_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_post_init.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_evolve.restype = C.c_long
        L.oracle_evolve.argtypes = [C.c_void_p, C.c_double, C.c_long]
        L.oracle_pgen_blast.argtypes = [C.c_void_p] + [C.c_double] * 7 + [C.c_int, C.c_int]
        L.oracle_pgen_linear_wave.restype = C.c_double
        L.oracle_pgen_linear_wave.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double,
                                              C.c_int, C.c_int, C.c_int, C.c_double]
        L.oracle_pgen_advection.restype = C.c_double
        L.oracle_pgen_advection.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int,
                                            C.c_int, C.c_int, C.c_double]
        L.oracle_linear_wave_errors.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        L.oracle_advection_errors.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.oracle_history.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.oracle_plm.argtypes = [C.c_double] * 3 + [C.POINTER(C.c_double)] * 2
        L.oracle_ppm4.argtypes = [C.c_double] * 5 + [C.POINTER(C.c_double)] * 2
        L.oracle_riemann.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double)]
        d, vp, i = C.c_double, C.c_void_p, C.c_int
        L.oracle_set_gravity_uniform.argtypes = [vp, d, d, d]
        L.oracle_set_gravity_point.argtypes = [vp] + [d] * 7
        L.oracle_set_gravity_binary.argtypes = [vp, C.POINTER(d)]
        L.oracle_set_gravity_window.argtypes = [vp, d, d]
        L.oracle_set_gravity_nbody.argtypes = [vp, i, C.POINTER(d), i]
        L.oracle_set_gravity_gm.argtypes = [vp, d]
        L.oracle_nbody_force.argtypes = [vp, C.POINTER(d), i]
        L.oracle_set_rotating_frame.argtypes = [vp, d, d]
        L.oracle_set_drag.argtypes = [vp, i, i, d, d, C.POINTER(d), C.POINTER(d)]
        L.oracle_set_damping.argtypes = [vp, i, C.POINTER(d)]
        L.oracle_external_gravity.argtypes = [vp, d, d]
        L.oracle_rotating_frame_force.argtypes = [vp, d]
        L.oracle_drag_source.argtypes = [vp, d]
        L.oracle_pgen_constant.argtypes = [vp] + [d] * 9
        L.oracle_pgen_strat.argtypes = [vp] + [d] * 4
        L.oracle_set_diffusion.argtypes = [vp, i, i, i, C.POINTER(d)]
        for f in ("zero_diffusion_flux", "viscous_flux", "thermal_flux"):
            getattr(L, "oracle_" + f).argtypes = [vp]
        L.oracle_diffusion_update.argtypes = [vp, d]
        L.oracle_qflux.restype = C.POINTER(d)
        L.oracle_qflux.argtypes = [vp, i]
        L.oracle_pgen_gaussian_bump.argtypes = [vp, C.POINTER(d)] + [d] * 11
        L.oracle_pgen_conduction.argtypes = [vp] + [d] * 6
        L.oracle_restrict_average.argtypes = [vp, vp, C.POINTER(i)]
        L.oracle_prolongate_minmod.argtypes = [vp, vp, C.POINTER(i)]
        L.oracle_restrict_average_field.argtypes = [vp, vp, C.POINTER(i), i]
        L.oracle_prolongate_minmod_field.argtypes = [vp, vp, C.POINTER(i), i]
        L.oracle_amr_first_derivative.argtypes = [vp, i, d, C.POINTER(d)]
        L.oracle_amr_first_derivative.restype = i
        L.oracle_amr_magnitude.argtypes = [vp, i, d, d, C.POINTER(d)]
        L.oracle_amr_magnitude.restype = i
        L.oracle_face_areas.argtypes = [vp, i, vp]
        L.oracle_set_damp_to_visc.argtypes = [vp, i]
        L.oracle_set_damp_to_visc.restype = i
        L.oracle_set_cooling.argtypes = [vp, C.POINTER(d)]
        L.oracle_cooling_source.argtypes = [vp, d, d]
        L.oracle_pgen_disk.argtypes = [vp, C.POINTER(d)]
        L.oracle_pgen_disk.restype = i
        _lib = L
    return _lib


EXCH = C.CFUNCTYPE(None, C.c_void_p)


class Oracle:
    """One block of the CPU oracle.  Keyword arguments follow the reference's input-deck
    names (inputs/*/*.in): reconstruct, riemann, gamma, cfl, dfloor, siefloor, integrator."""

    def __init__(self, nx, xmin, xmax, ng=2, ns_gas=1, ns_dust=0, reconstruct="plm",
                 riemann="hllc", dust_reconstruct="plm", dust_riemann="hlle", gamma=1.66666666667,
                 dfloor=1.0e-20, siefloor=1.0e-20, de_switch=0.0, dust_dfloor=1.0e-20, cfl=0.8,
                 dust_cfl=0.8, bc=("periodic",) * 6, integrator="rk2", mesh_bounds=None,
                 coordinates="cartesian", nthreads=0):
        self.L = lib()
        c = Cfg()
        c.nx1, c.nx2, c.nx3 = nx
        c.ng = ng
        c.x1min, c.x2min, c.x3min = xmin
        c.x1max, c.x2max, c.x3max = xmax
        c.ns_gas, c.ns_dust = ns_gas, ns_dust
        c.recon_gas, c.riemann_gas = RC[reconstruct], RS[riemann]
        c.recon_dust, c.riemann_dust = RC[dust_reconstruct], RS[dust_riemann]
        c.gamma, c.dfloor_gas, c.siefloor_gas = gamma, dfloor, siefloor
        c.de_switch, c.dfloor_dust = de_switch, dust_dfloor
        c.cfl_gas, c.cfl_dust = cfl, dust_cfl
        for i, b in enumerate(bc):
            c.bc[i] = BC[b] if isinstance(b, str) else b
        c.integrator = INTEG[integrator]
        c.nthreads = nthreads  # > 0: omp_set_num_threads (process-wide); 0: leave the OpenMP default
        c.coords = coord_select(coordinates, sum(n > 1 for n in nx))
        self.cfg = c
        self.h = C.c_void_p(self.L.oracle_create(C.byref(c)))
        d = (C.c_int * 10)()
        self.L.oracle_dims(self.h, d)
        (self.ni, self.nj, self.nk, self.is_, self.ie, self.js, self.je, self.ks, self.ke,
         self.ndim) = list(d)
        self.N = self.ni * self.nj * self.nk
        if mesh_bounds is not None:
            self.L.oracle_set_mesh_bounds(self.h, *[float(x) for x in mesh_bounds])
        self._cb = None

    def __del__(self):
        try:
            self.L.oracle_destroy(self.h)
        except Exception:
            pass

    # numpy views [nvar, nk, nj, ni] of the oracle's own arrays (no copy)
    def field(self, which, nvar):
        p = self.L.oracle_field(self.h, which)
        a = np.ctypeslib.as_array(p, shape=(nvar * self.N,))
        return a.reshape(nvar, self.nk, self.nj, self.ni)

    @property
    def gprim(self): return self.field(F_GPRIM, 6 * self.cfg.ns_gas)
    @property
    def gu0(self): return self.field(F_GU0, 6 * self.cfg.ns_gas)
    @property
    def gu1(self): return self.field(F_GU1, 6 * self.cfg.ns_gas)
    def gflux(self, d): return self.field(F_GFLUX + d, 6 * self.cfg.ns_gas)
    def gpflux(self, d): return self.field(F_GPFLUX + d, self.cfg.ns_gas)
    def gvface(self, d): return self.field(F_GVFACE + d, self.cfg.ns_gas)
    @property
    def dprim(self): return self.field(F_DPRIM, 4 * self.cfg.ns_dust)
    @property
    def du0(self): return self.field(F_DU0, 4 * self.cfg.ns_dust)
    @property
    def du1(self): return self.field(F_DU1, 4 * self.cfg.ns_dust)
    def dflux(self, d): return self.field(F_DFLUX + d, 4 * self.cfg.ns_dust)

    def interior(self, a):
        return a[..., self.ks:self.ke + 1, self.js:self.je + 1, self.is_:self.ie + 1]

    # reference task names (artemis_driver.cpp:145-273)
    def CalculateFluxes(self, fluid=GAS, pcm=False): self.L.oracle_calculate_fluxes(self.h, fluid, int(pcm))
    def ApplyUpdate(self, gam0, gam1, beta_dt): self.L.oracle_apply_update(self.h, gam0, gam1, beta_dt)
    def FluxSource(self, dt, fluid=GAS): self.L.oracle_flux_source(self.h, fluid, dt)
    def SetAuxillaryFields(self): self.L.oracle_set_aux(self.h)
    def ConsToPrim(self): self.L.oracle_cons_to_prim(self.h)
    def PrimToCons(self): self.L.oracle_prim_to_cons(self.h)
    def DeepCopyConservedData(self): self.L.oracle_deep_copy(self.h)
    def EstimateTimestepMesh(self, fluid=GAS): return self.L.oracle_estimate_dt(self.h, fluid)
    def ApplyBoundaryConditions(self): self.L.oracle_apply_bcs(self.h)
    def new_dt(self): return self.L.oracle_new_dt(self.h)

    @property
    def time(self): return self.L.oracle_time(self.h)
    @property
    def dt(self): return self.L.oracle_dt(self.h)
    @dt.setter
    def dt(self, v): self.L.oracle_set_dt(self.h, v)
    @property
    def ncycle(self): return self.L.oracle_ncycle(self.h)

    def step(self, exchange=None):
        if exchange is None:
            self.L.oracle_step(self.h, None, None)
        else:
            cb = EXCH(lambda ctx: exchange())
            self.L.oracle_step(self.h, C.cast(cb, C.c_void_p), None)

    def evolve(self, tlim=-1.0, nlim=-1):
        return self.L.oracle_evolve(self.h, tlim, nlim)

    # problem generators (pgen/*.hpp)
    def post_init(self, exchange=None):
        """Mesh::Initialize's first boundary communication (ConsToPrim, ghosts, PrimToCons)."""
        if exchange is None:
            self.L.oracle_post_init(self.h, None, None)
        else:
            cb = EXCH(lambda ctx: exchange())
            self.L.oracle_post_init(self.h, C.cast(cb, C.c_void_p), None)

    # ---- optional source packages (deck block names in the docstrings) ------------------------
    def set_gravity_uniform(self, gx1, gx2, gx3):
        """<gravity/uniform> gx1, gx2, gx3"""
        self.L.oracle_set_gravity_uniform(self.h, gx1, gx2, gx3)

    def set_gravity_point(self, mass, soft=0.0, sink=0.0, sink_rate=0.0, x=0.0, y=0.0, z=0.0):
        """<gravity/point> mass, soft, sink, sink_rate, x, y, z"""
        self.L.oracle_set_gravity_point(self.h, mass, soft, sink, sink_rate, x, y, z)

    def set_gravity_binary(self, mass, q, a, e=0.0, i=0.0, omega=0.0, Omega=0.0, f=180.0, soft1=0.0, soft2=0.0,
                           sink1=0.0, sink2=0.0, sink_rate1=0.0, sink_rate2=0.0, x=0.0, y=0.0, z=0.0):
        """<gravity/binary> (angles in degrees like the deck; gravity.cpp:99-104 converts with M_PI / 180.)"""
        import math
        r = lambda deg: deg * math.pi / 180.
        self.L.oracle_set_gravity_binary(self.h, (C.c_double * 17)(
            mass, q, a, e, r(i), r(omega), r(Omega), r(f), soft1, soft2, sink1, sink2, sink_rate1, sink_rate2,
            x, y, z))

    def set_gravity_nbody(self, particles, frame_correction=True, gm=None):
        """<gravity/nbody> + the nbody package's particles (static: <nbody> integrator = none).  particles: dicts with
        GM, pos, vel, xf, vf, rs, racc, gamma, beta, spline, couple (nbody/particle_base.hpp:53-93)."""
        par = []
        for p in particles:
            par += [p["GM"], *p.get("pos", (0, 0, 0)), *p.get("vel", (0, 0, 0)), *p.get("xf", (0, 0, 0)),
                    *p.get("vf", (0, 0, 0)), p.get("rs", 0.0), p.get("racc", 0.0), p.get("gamma", 0.0),
                    p.get("beta", 0.0), float(p.get("spline", 0)), float(p.get("couple", 1)), 0.0]
        self._npart = len(particles)
        self.L.oracle_set_gravity_nbody(self.h, len(particles), (C.c_double * len(par))(*par), int(frame_correction))
        if gm is not None:  # nbody.cpp:109: G * mtot as read / summed before the rescale (default: the sum of the GM given)
            self.L.oracle_set_gravity_gm(self.h, C.c_double(gm))

    def nbody_force(self, reset=False):
        out = (C.c_double * (7 * self._npart))()
        self.L.oracle_nbody_force(self.h, out, int(reset))
        return np.array(out[:]).reshape(self._npart, 7)

    def set_gravity_window(self, tstart, tstop):
        self.L.oracle_set_gravity_window(self.h, tstart, tstop)

    def set_rotating_frame(self, omega, qshear=0.0):
        """<rotating_frame> omega, qshear"""
        self.L.oracle_set_rotating_frame(self.h, omega, qshear)

    def set_drag(self, type="simple_dust", model="constant", tau=None, scale=1.0, grain_density=1.0,
                 sizes=None):
        """<drag> type; <dust/stopping_time> type, tau, scale; <dust> sizes, grain_density"""
        nd = self.cfg.ns_dust
        arr = lambda v: (C.c_double * max(nd, 1))(*(list(v) if v is not None else [0.0] * nd))
        self.L.oracle_set_drag(self.h, {"simple_dust": 1, "self": 2}[type],
                               {"constant": 0, "stokes": 1}[model], scale, grain_density,
                               arr(tau), arr(sizes))

    def set_damping(self, fluid, inner=(-1.7976931348623157e308,) * 3, inner_rate=(0.0,) * 3,
                    outer=(1.7976931348623157e308,) * 3, outer_rate=(0.0,) * 3):
        """<gas/damping> / <dust/damping> inner_x*, inner_x*_rate, outer_x*, outer_x*_rate"""
        p = (C.c_double * 12)(*inner, *inner_rate, *outer, *outer_rate)
        self.L.oracle_set_damping(self.h, fluid, p)

    def ExternalGravity(self, time, dt): self.L.oracle_external_gravity(self.h, time, dt)
    def RotatingFrameForce(self, dt): self.L.oracle_rotating_frame_force(self.h, dt)
    def DragSource(self, dt): self.L.oracle_drag_source(self.h, dt)

    def pgen_constant(self, gas_rho=1.0, gas_v=(0.0, 0.0, 0.0), gas_temp=1.0, dust_rho=1.0,
                      dust_v=(0.0, 0.0, 0.0), post_init=True):
        self.L.oracle_pgen_constant(self.h, gas_rho, *gas_v, gas_temp, dust_rho, *dust_v)
        if post_init:
            self.post_init()

    def pgen_strat(self, rho0=1.0, dens_min=1.0e-5, h=1.0, dust_to_gas=0.01, post_init=True):
        self.L.oracle_pgen_strat(self.h, rho0, dens_min, h, dust_to_gas)
        if post_init:
            self.post_init()

    def set_viscosity(self, type="constant", nu=0.0, alpha=0.0, eta_bulk=0.0, r_exp=0.0, r0=1.0,
                      Omega0=0.0, averaging="arithmetic"):
        """<gas/viscosity> type = constant|powerlaw|alpha, nu|alpha, eta_bulk, r_exp, averaging"""
        t = {"constant": 1, "powerlaw": 1, "alpha": 2}[type]
        p = (C.c_double * 9)(nu if t == 1 else alpha, eta_bulk, r_exp, r0, Omega0, 0.0, 0.0, 1.0, 1.0)
        self.L.oracle_set_diffusion(self.h, 0, t, {"arithmetic": 0, "harmonic": 1}[averaging], p)

    def set_conductivity(self, type="conductivity", cond=0.0, kappa=0.0, temp_exp=0.0, rho_exp=0.0,
                         rho_ref=1.0, T_ref=1.0, averaging="arithmetic"):
        """<gas/conductivity> type = conductivity|diffusivity, cond|kappa, temp_exp, rho_exp, ..."""
        t = {"conductivity": 3, "diffusivity": 4}[type]
        p = (C.c_double * 9)(cond if t == 3 else kappa, 0.0, 0.0, 1.0, 0.0, temp_exp, rho_exp, rho_ref, T_ref)
        self.L.oracle_set_diffusion(self.h, 1, t, {"arithmetic": 0, "harmonic": 1}[averaging], p)

    def ZeroDiffusionFlux(self): self.L.oracle_zero_diffusion_flux(self.h)
    def ViscousFlux(self): self.L.oracle_viscous_flux(self.h)
    def ThermalFlux(self): self.L.oracle_thermal_flux(self.h)
    def DiffusionUpdate(self, dt): self.L.oracle_diffusion_update(self.h, dt)

    def qflux(self, d):
        """gas::diff::momentum (3n+c) and gas::diff::energy (3ns+n) face fluxes of direction d"""
        nv = 4 * self.cfg.ns_gas
        p = self.L.oracle_qflux(self.h, d)
        return np.ctypeslib.as_array(p, shape=(nv, self.nk, self.nj, self.ni))

    def pgen_gaussian_bump(self, sigma, centre=(0.0, 0.0, 0.0), density_bump=0.0, temperature_bump=0.0,
                           v_bump=(0.0, 0.0, 0.0), gas_rho=1.0, gas_v=(0.0, 0.0, 0.0), gas_pres=1.0,
                           post_init=True):
        xc = (C.c_double * 3)(*centre)
        self.L.oracle_pgen_gaussian_bump(self.h, xc, sigma, density_bump, temperature_bump, *v_bump,
                                         gas_rho, *gas_v, gas_pres)
        if post_init:
            self.post_init()

    def pgen_conduction(self, gas_rho=1.0, gas_v=(0.0, 0.0, 0.0), gas_temp=1.0, flux=0.0, post_init=True):
        self.L.oracle_pgen_conduction(self.h, gas_rho, *gas_v, gas_temp, flux)
        if post_init:
            self.post_init()

    REFINE_FIELDS = {"gas.prim": 0, "gas.cons": 1, "dust.prim": 2, "dust.cons": 3}

    def RestrictAverage(self, coarse, crange, corigin, forigin, field="gas.prim"):
        """RestrictAverage<GEOM> of this (fine) oracle's `field` arrays (default: the gas primitives) onto `coarse`'s:
        crange = (cis, cie, cjs, cje, cks, cke), corigin / forigin = the coarse / fine indices that coincide
        (cib.s <-> ib.s)."""
        self.L.oracle_restrict_average_field(self.h, coarse.h, (C.c_int * 12)(*crange, *corigin, *forigin),
                                             self.REFINE_FIELDS[field])

    def ProlongateSharedMinMod(self, coarse, crange, corigin, forigin, field="gas.prim"):
        self.L.oracle_prolongate_minmod_field(self.h, coarse.h, (C.c_int * 12)(*crange, *corigin, *forigin),
                                              self.REFINE_FIELDS[field])

    def face_areas(self, dir):
        """Lower-face areas GetFaceArea<dir> (dir = 1..3) or cell volumes (dir = 0), [nk, nj, ni]."""
        out = np.empty((self.nk, self.nj, self.ni))
        self.L.oracle_face_areas(self.h, dir, out.ctypes.data)
        return out

    def set_damp_to_visc(self, on=True):
        """<gas/damping> damp_to_visc (drag.hpp:101): the gas damping relaxes towards the viscous inflow
        velocity of the <gas/viscosity> set before (powerlaw / constant or alpha, drag.cpp:113-121)."""
        if self.L.oracle_set_damp_to_visc(self.h, int(on)):
            raise ValueError("The chosen viscosity model does not work with damping")

    def ScalarFirstDerivative(self, var, thr):
        """amr_criteria.hpp:28-132 on gas primitive component `var` (-1: pressure); returns (AmrTag, maxeps)"""
        m = C.c_double(0.0)
        tag = self.L.oracle_amr_first_derivative(self.h, var, thr, C.byref(m))
        return tag, m.value

    def ScalarMagnitude(self, var, refine_above, deref_below):
        """amr_criteria.hpp:137-168; returns (AmrTag, max)"""
        m = C.c_double(0.0)
        tag = self.L.oracle_amr_magnitude(self.h, var, refine_above, deref_below, C.byref(m))
        return tag, m.value

    def set_cooling(self, beta0, beta_min=1e-12, exp_scale=0.0, tfloor=0.0, tcyl=0.0, cyl_plaw=0.0, tsph=0.0,
                    sph_plaw=0.0):
        """<cooling> type = beta, tref = powerlaw (gas/cooling/cooling.cpp:34-63)"""
        self.L.oracle_set_cooling(self.h, (C.c_double * 8)(beta0, beta_min, exp_scale, tfloor, tcyl, cyl_plaw,
                                                          tsph, sph_plaw))

    def CoolingSource(self, time, dt): self.L.oracle_cooling_source(self.h, time, dt)

    def pgen_disk(self, r0=1.0, rho0=1.0, dslope=-2.25, h0=0.05, polytropic_index=None, dens_min=1.0e-5,
                  pres_min=1.0e-8, rexp=0.0, rcav=0.0, l0=0.0, dust_to_gas=0.01, temp_soft=0.0,
                  tslope=None, flare=None, quiet_start=False, mdot=None, post_init=True):
        """<problem> block of inputs/disk/*.in (pgen/disk.hpp:253-323); set gravity, the rotating
        frame and the viscosity first.  BC names: "ic" and "disk_extrap" (the deck's `extrap`)."""
        none = -1.7976931348623157e308  # -Big<Real>()
        par = (C.c_double * 16)(r0, rho0, dslope, h0,
                                self.cfg.gamma if polytropic_index is None else polytropic_index,
                                dens_min, pres_min, rexp, rcav, l0, dust_to_gas, temp_soft,
                                none if tslope is None else tslope, none if flare is None else flare,
                                1.0 if quiet_start else 0.0, -1.0 if mdot is None else mdot)
        rc = self.L.oracle_pgen_disk(self.h, par)
        if rc:
            raise ValueError({1: "problem/gamma needs to be >= 1", 2: "Set flare or tslope in <problem>",
                              3: "Set either flare or tslope in <problem> not both!"}[rc])
        if post_init:
            self.post_init()

    def pgen_blast(self, radius=1.0, internal_energy=1.0, p0=1.0, d0=1.0, x0=(0.0, 0.0, 0.0),
                   samples=-1, symmetry="spherical", post_init=True):
        self.L.oracle_pgen_blast(self.h, radius, internal_energy, p0, d0, x0[0], x0[1], x0[2],
                                 samples, 1 if symmetry == "spherical" else 2)
        if post_init:
            self.post_init()

    def pgen_linear_wave(self, wave_flag, amp, vflow=0.0, along=(False, False, False), nperiod=1.0,
                         post_init=True):
        t = self.L.oracle_pgen_linear_wave(self.h, wave_flag, amp, vflow, *map(int, along), nperiod)
        if post_init:
            self.post_init()
        return t

    def pgen_advection(self, amp, vflow=1.0, along=(False, False, False), nperiod=1.0, post_init=True):
        t = self.L.oracle_pgen_advection(self.h, amp, vflow, *map(int, along), nperiod)
        if post_init:
            self.post_init()
        return t

    def linear_wave_errors(self, do_rms=True):
        out = (C.c_double * 6)()
        self.L.oracle_linear_wave_errors(self.h, out, int(do_rms))
        return np.array(out[:])

    def advection_errors(self):
        out = (C.c_double * 16)()
        self.L.oracle_advection_errors(self.h, out)
        return np.array(out[:])

    def history(self):
        n = 6 + 4 * self.cfg.ns_dust
        out = (C.c_double * n)()
        self.L.oracle_history(self.h, out)
        return np.array(out[:])


def plm(qm, q, qp):
    a, b = C.c_double(), C.c_double()
    lib().oracle_plm(qm, q, qp, C.byref(a), C.byref(b))
    return a.value, b.value


def ppm4(qmm, qm, q, qp, qpp):
    a, b = C.c_double(), C.c_double()
    lib().oracle_ppm4(qmm, qm, q, qp, qpp, C.byref(a), C.byref(b))
    return a.value, b.value


def riemann(fluid, solver, gm1, wl, wr):
    wl = np.ascontiguousarray(wl, dtype=np.float64)
    wr = np.ascontiguousarray(wr, dtype=np.float64)
    out = np.zeros(8)
    P = C.POINTER(C.c_double)
    lib().oracle_riemann(fluid, RS[solver] if isinstance(solver, str) else solver, gm1,
                         wl.ctypes.data_as(P), wr.ctypes.data_as(P), out.ctypes.data_as(P))
    return out
