"""MeshBlockPack: device arrays + pointer tables for a set of equal-sized mesh blocks, with
methods named after the reference's Parthenon tasks (artemis_driver.cpp:145-273).  Every
method is a single call through the C ABI of libartemis_hip.so; torch only owns the HBM
allocations and the stream.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import capi


def _ptr_table(t):
    """Device int64 table of pointers to t[b, v] planes; t is [nb, nvar, nk, nj, ni]."""
    nb, nv = t.shape[0], t.shape[1]
    stride = t.stride(1) * t.element_size()
    bstride = t.stride(0) * t.element_size()
    base = t.data_ptr()
    host = np.array([base + b * bstride + v * stride for b in range(nb) for v in range(nv)],
                    dtype=np.int64)
    return torch.from_numpy(host).to(t.device)


class MeshBlockPack:
    def __init__(self, nblocks, nx, xmin, xmax, ng=2, ns_gas=1, ns_dust=0, reconstruct="plm",
                 riemann="hllc", dust_reconstruct="plm", dust_riemann="hlle", gamma=1.66666666667,
                 dfloor=1.0e-20, siefloor=1.0e-20, de_switch=0.0, dust_dfloor=1.0e-20,
                 device="cuda:0", with_fluxes=True, coordinates="cartesian", with_diffusion=False,
                 omega_frame=0.0):
        """xmin/xmax: per-block interior bounds, arrays of shape [nblocks, 3]."""
        self.L = capi.load()
        self.dev = torch.device(device)
        self.nb, self.ng = nblocks, ng
        self.nx = tuple(int(n) for n in nx)
        self.ndim = 3 if nx[2] > 1 else (2 if nx[1] > 1 else 1)
        g = [ng, ng if nx[1] > 1 else 0, ng if nx[2] > 1 else 0]
        self.ni, self.nj, self.nk = nx[0] + 2 * g[0], nx[1] + 2 * g[1], nx[2] + 2 * g[2]
        self.is_, self.js, self.ks = g
        self.ie, self.je, self.ke = g[0] + nx[0] - 1, g[1] + nx[1] - 1, g[2] + nx[2] - 1
        self.nsg, self.nsd = ns_gas, ns_dust
        xmin = np.asarray(xmin, dtype=np.float64).reshape(nblocks, 3)
        xmax = np.asarray(xmax, dtype=np.float64).reshape(nblocks, 3)
        geom = np.zeros((nblocks, 6))
        for d in range(3):
            dx = (xmax[:, d] - xmin[:, d]) / nx[d]  # parthenon UniformCartesian (upstream)
            geom[:, 2 * d] = xmin[:, d] - g[d] * dx
            geom[:, 2 * d + 1] = dx
        self.geom_host = geom
        self.geom = torch.from_numpy(geom).to(self.dev)
        shp = (self.nk, self.nj, self.ni)

        def alloc(nv):
            return torch.zeros((nblocks, max(nv, 0)) + shp, dtype=torch.float64, device=self.dev)

        self.gas_prim, self.gas_u0, self.gas_u1 = alloc(6 * ns_gas), alloc(6 * ns_gas), alloc(6 * ns_gas)
        self.dust_prim, self.dust_u0, self.dust_u1 = alloc(4 * ns_dust), alloc(4 * ns_dust), alloc(4 * ns_dust)
        nf = 3 if with_fluxes else 0
        self.gas_flux = [alloc(6 * ns_gas) for _ in range(nf)]
        self.gas_pflux = [alloc(ns_gas) for _ in range(nf)]
        self.gas_vface = [alloc(ns_gas) for _ in range(nf)]
        self.dust_flux = [alloc(4 * ns_dust) for _ in range(nf)]
        self.gas_diff_flux = [alloc(4 * ns_gas) for _ in range(3 if with_diffusion else 0)]
        self._tables = []

        def tab(t):
            if t.shape[1] == 0:
                return None
            tt = _ptr_table(t)
            self._tables.append(tt)
            return tt.data_ptr()

        p = capi.Pack()
        p.nblocks, p.nghost = nblocks, ng
        p.nx1, p.nx2, p.nx3 = self.nx
        p.coords = capi.coord_select(coordinates, self.ndim)
        p.gm1 = gamma - 1.0
        p.geom = self.geom.data_ptr()
        p.omega_frame = omega_frame  # rotating_frame/omega (FluxSource's frame velocity), 0 when off
        p.gas.nspecies, p.gas.recon, p.gas.riemann = ns_gas, capi.RECON[reconstruct], capi.RSOLVER[riemann]
        p.gas.dfloor, p.gas.siefloor, p.gas.de_switch = dfloor, siefloor, de_switch
        p.gas.prim, p.gas.cons0, p.gas.cons1 = tab(self.gas_prim), tab(self.gas_u0), tab(self.gas_u1)
        p.dust.nspecies, p.dust.recon, p.dust.riemann = ns_dust, capi.RECON[dust_reconstruct], capi.RSOLVER[dust_riemann]
        p.dust.dfloor = dust_dfloor
        p.dust.prim, p.dust.cons0, p.dust.cons1 = tab(self.dust_prim), tab(self.dust_u0), tab(self.dust_u1)
        for d in range(nf):
            p.gas.flux[d], p.gas.pflux[d], p.gas.vface[d] = tab(self.gas_flux[d]), tab(self.gas_pflux[d]), tab(self.gas_vface[d])
            p.dust.flux[d] = tab(self.dust_flux[d])
        for d in range(len(self.gas_diff_flux)):
            p.gas.diff_flux[d] = tab(self.gas_diff_flux[d])
        self.pack = p
        # x2 trigonometry tables of spherical2D/3D (host libm, see include/artemis_hip.h)
        nmetric = self.L.artemis_hip_metric_count(C.byref(p))
        if nmetric > 0:
            mt = np.zeros(nmetric)
            capi.check(self.L.artemis_hip_metric_fill(C.byref(p), geom.ctypes.data, mt.ctypes.data))
            self.metric_host = mt
            self.metric = torch.from_numpy(mt).to(self.dev)
            p.metric = self.metric.data_ptr()
        # PLM_G's geometric weights (artemis_hip_plm_table_fill): on by default for curvilinear packs, as the host driver
        # runs; ARTEMIS_NO_PLM_TABLE=1 (or set_plm_table(False)) makes the kernels form them per face
        self.plm_table = None
        if coordinates != "cartesian" and not os.environ.get("ARTEMIS_NO_PLM_TABLE"):
            self.set_plm_table(True)
        self.gas_prim_table = p.gas.prim
        self.dust_prim_table = p.dust.prim
        self._extra_prim = {}

    def set_plm_table(self, on):
        if not on:
            self.pack.plm_table = None
            return
        if self.plm_table is None:
            n = self.L.artemis_hip_plm_table_count(C.byref(self.pack))
            self.plm_table = torch.zeros(n, dtype=torch.float64, device=self.dev)
            capi.check(self.L.artemis_hip_plm_table_fill(C.byref(self.pack), C.c_void_p(self.plm_table.data_ptr()), None))
            torch.cuda.synchronize()
        self.pack.plm_table = self.plm_table.data_ptr()

    # ---- helpers -------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def _call(self, fn, *args):
        capi.check(fn(C.byref(self.pack), *args, self._stream()))

    def new_prim_buffer(self, name):
        """Extra gas-prim buffer (ping-pong target of the fused stage); returns its table."""
        t = torch.zeros_like(self.gas_prim)
        tt = _ptr_table(t)
        self._extra_prim[name] = (t, tt)
        return t, tt.data_ptr()

    # ---- reference task names --------------------------------------------------------------
    def CalculateFluxes(self, fluid=capi.GAS, pcm=False):
        self._call(self.L.artemis_hip_calculate_fluxes, fluid, int(pcm))

    def ApplyUpdate(self, gam0, gam1, beta_dt):
        self._call(self.L.artemis_hip_apply_update, gam0, gam1, beta_dt)

    def FluxSource(self, dt, fluid=capi.GAS):
        self._call(self.L.artemis_hip_flux_source, fluid, dt)

    def SetAuxillaryFields(self):
        self._call(self.L.artemis_hip_set_aux)

    def ConsToPrim(self):
        self._call(self.L.artemis_hip_cons_to_prim)

    def PrimToCons(self):
        self._call(self.L.artemis_hip_prim_to_cons)

    def DeepCopyConservedData(self):
        self._call(self.L.artemis_hip_deep_copy_conserved)

    def EstimateTimestepMesh(self, fluid=capi.GAS, cfl=1.0):
        out = C.c_double(0.0)
        self._call(self.L.artemis_hip_estimate_dt, fluid, cfl, C.byref(out))
        return out.value

    def ApplyBoundaryConditions(self, bc, strat=None, conductive=None, disk=None):
        """bc: per-block list of 6 names/flags (ix1, ox1, ix2, ox2, ix3, ox3); strat = (qshear,
        omega) when a block carries the strat problem's `extrap` / `inflow` conditions; disk =
        dict(ic_gas=table, ic_dust=table, omf=...) for the disk problem's `ic` / `disk_extrap`."""
        flat = []
        for row in bc:
            flat += [capi.BCS[x] if isinstance(x, str) else int(x) for x in row]
        arr = (C.c_int * len(flat))(*flat)
        par = None
        if strat is not None or conductive is not None or disk is not None:
            bp = capi.BcParams()
            if disk is not None:
                bp.ic_gas, bp.ic_dust = disk.get("ic_gas"), disk.get("ic_dust")
                bp.disk_omf = disk.get("omf", 0.0)
                bp.disk_nu0, bp.disk_nu_indx = disk.get("nu0", 0.0), disk.get("nu_indx", 0.0)
                bp.disk_r0, bp.disk_mdot = disk.get("r0", 1.0), disk.get("mdot", 0.0)
            if strat is not None:
                bp.qshear, bp.omega = strat
            if conductive is not None:  # dict(temp, flux, g=(gx1,gx2,gx3), coeff, cv, type)
                bp.cond_temp, bp.cond_flux = conductive["temp"], conductive["flux"]
                bp.cond_g[:] = list(conductive.get("g", (0.0, 0.0, 0.0)))
                bp.cond_coeff, bp.cond_cv = conductive["coeff"], conductive["cv"]
                bp.cond_type = conductive.get("type", capi.CONDUCTIVITY_PLAW)
                bp.cond_temp_exp, bp.cond_rho_exp = conductive.get("temp_exp", 0.0), conductive.get("rho_exp", 0.0)
                bp.cond_T_ref, bp.cond_rho_ref = conductive.get("T_ref", 1.0), conductive.get("rho_ref", 1.0)
            par = C.byref(bp)
        self._call(self.L.artemis_hip_apply_bc, arr, par)

    def new_dust_prim_buffer(self, name):
        t = torch.zeros_like(self.dust_prim)
        tt = _ptr_table(t) if t.shape[1] else None
        self._extra_prim["dust:" + name] = (t, tt)
        return t, (tt.data_ptr() if tt is not None else None)

    def stage_general(self, gam0, gam1, beta_dt, bdt, gas=(None, None, None), dust=(None, None, None),
                      pcm=False, time=0.0, gravity=None, rotating_frame=None, drag=None,
                      cfl=(0.0, 0.0), dt_dev=None, diffusion=None, cooling=None, diffusion_sums=None, nbody=None,
                      nbody_omf=0.0, defer_finish=False):
        """artemis_hip_stage_general: gas / dust = (in, u1, out) prim tables; diffusion = capi.Diffusion
        whose fluxes are already in gas_diff_flux for the `in` primitives -- or, with diffusion_sums (the table
        viscous_source returned for them), no flux array at all; cooling = capi.Cooling."""
        a = capi.StageGeneralArgs()
        a.gam0, a.gam1, a.beta_dt, a.bdt, a.pcm, a.time = gam0, gam1, beta_dt, bdt, int(pcm), time
        a.gas_in, a.gas_u1, a.gas_out = gas
        a.dust_in, a.dust_u1, a.dust_out = dust
        if gravity is not None:
            a.gravity = C.pointer(gravity)
        if rotating_frame is not None:
            a.rf_omega, a.rf_qshear = rotating_frame
        if drag is not None:
            a.drag = C.pointer(drag)
        a.cfl_gas, a.cfl_dust = cfl
        a.dt_dev = dt_dev
        if diffusion is not None:
            a.diffusion = C.pointer(diffusion)
        if cooling is not None:
            a.cooling = C.pointer(cooling)
        if diffusion_sums is not None:
            a.diffusion_sums = diffusion_sums
        if nbody is not None:  # (device array, count) of nbody_device(): Gravity::NBodyGravity inside the stage
            self._nbody_keep = nbody
            a.nbody_dev, a.nbody_n, a.nbody_omf = nbody[0].data_ptr(), nbody[1], nbody_omf
        a.defer_finish = int(defer_finish)  # (False / True, or the ABI's 0 / 1 / 2)
        self._last_general_args = a
        # which kernel this call takes (0 cell-centred, 1 2-D row march, 2 curvilinear streaming tile)
        self.last_stage_variant = self.L.artemis_hip_stage_general_variant(C.byref(self.pack), C.byref(a))
        self._call(self.L.artemis_hip_stage_general, C.byref(a))

    # ---- refined meshes on the one-kernel stages (include/artemis_hip.h "flux correction as a thin fix-up") ----
    def _device_records(self, records, ctype):
        arr = (ctype * len(records))(*records)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        return host.to(self.gas_prim.device if self.gas_prim.numel() else self.dust_prim.device)

    def ml_face_fluxes(self, boxes):
        """artemis_hip_ml_face_fluxes with the arguments of the last stage_general call; boxes = [(block, dir, lo, n)]."""
        recs = [capi.MlFaceBox(b, d, (C.c_int * 3)(*lo), (C.c_int * 3)(*n)) for b, d, lo, n in boxes]
        dev = self._device_records(recs, capi.MlFaceBox)
        self._call(self.L.artemis_hip_ml_face_fluxes, C.byref(self._last_general_args), C.c_void_p(dev.data_ptr()), len(recs))
        torch.cuda.synchronize()

    def ml_stage_fixup(self, cells):
        """artemis_hip_ml_stage_fixup with the arguments of the last stage_general call; cells = [(block, k, j, i, faces)]."""
        recs = [capi.MlFixCell(*c) for c in cells]
        dev = self._device_records(recs, capi.MlFixCell)
        self._call(self.L.artemis_hip_ml_stage_fixup, C.byref(self._last_general_args), C.c_void_p(dev.data_ptr()), len(recs))
        torch.cuda.synchronize()

    def stage_epilogue(self, gam0, gam1, beta_dt, bdt, time=0.0, gravity=None, rotating_frame=None,
                       diffusion=None, cooling=None):
        """artemis_hip_stage_epilogue: ApplyUpdate ... ConsToPrim in one pass over the stored fluxes."""
        a = capi.StageGeneralArgs()
        a.gam0, a.gam1, a.beta_dt, a.bdt, a.time = gam0, gam1, beta_dt, bdt, time
        if gravity is not None:
            a.gravity = C.pointer(gravity)
        if rotating_frame is not None:
            a.rf_omega, a.rf_qshear = rotating_frame
        if diffusion is not None:
            a.diffusion = C.pointer(diffusion)
        if cooling is not None:
            a.cooling = C.pointer(cooling)
        self._call(self.L.artemis_hip_stage_epilogue, C.byref(a))

    def stage_epilogue_cons(self, gam0, gam1, beta_dt, bdt, time=0.0, gravity=None, rotating_frame=None, diffusion=None):
        """artemis_hip_stage_epilogue_cons: ApplyUpdate ... RotatingFrameForce over the stored fluxes, state left in cons0."""
        a = capi.StageGeneralArgs()
        a.gam0, a.gam1, a.beta_dt, a.bdt, a.time = gam0, gam1, beta_dt, bdt, time
        if gravity is not None:
            a.gravity = C.pointer(gravity)
        if rotating_frame is not None:
            a.rf_omega, a.rf_qshear = rotating_frame
        if diffusion is not None:
            a.diffusion = C.pointer(diffusion)
        self._call(self.L.artemis_hip_stage_epilogue_cons, C.byref(a))

    def stage_finish(self, time, dt, drag=None):
        """artemis_hip_stage_finish: [DragSource] + SetAuxillaryFields + ConsToPrim of cons0 into the primitives."""
        self._call(self.L.artemis_hip_stage_finish, C.byref(drag) if drag is not None else None, time, dt)

    def stage_finish_cells(self, time, dt, drag, cells):
        """artemis_hip_stage_finish_cells: the same three tasks on the listed zones only; cells = [(block, k, j, i, faces)]."""
        recs = [capi.MlFixCell(*c) for c in cells]
        dev = self._device_records(recs, capi.MlFixCell)
        self._call(self.L.artemis_hip_stage_finish_cells, C.byref(drag) if drag is not None else None, time, dt,
                   C.c_void_p(dev.data_ptr()), len(recs))
        torch.cuda.synchronize()

    # ---- gas diffusion (artemis_driver.cpp:189-193, :218-221) -----------------------------------
    def ZeroDiffusionFlux(self):
        self._call(self.L.artemis_hip_zero_diffusion_flux)

    def ViscousFlux(self, diffusion):
        self._call(self.L.artemis_hip_viscous_flux, C.byref(diffusion))

    def ZeroViscousFlux(self, diffusion):
        """ZeroDiffusionFlux + ViscousFlux in one pass (artemis_hip_zero_viscous_flux)."""
        self._call(self.L.artemis_hip_zero_viscous_flux, C.byref(diffusion))

    def ThermalFlux(self, diffusion):
        self._call(self.L.artemis_hip_thermal_flux, C.byref(diffusion))

    def viscous_source_covers(self):
        return bool(self.L.artemis_hip_viscous_source_covers(C.byref(self.pack)))

    def viscous_source(self, diffusion, dt):
        """artemis_hip_viscous_source on gas_prim: returns (sums tensor [nb, 5, nk, nj, ni], its device table pointer)."""
        if getattr(self, "_vsums", None) is None:
            self._vsums = torch.zeros((self.gas_prim.shape[0], 5) + tuple(self.gas_prim.shape[2:]), dtype=torch.float64,
                                      device=self.gas_prim.device)
            self._vsums_tab = _ptr_table(self._vsums)
        self._call(self.L.artemis_hip_viscous_source, C.byref(diffusion), dt, None, self._vsums_tab.data_ptr())
        return self._vsums, self._vsums_tab.data_ptr()

    def DiffusionUpdate(self, diffusion, dt):
        self._call(self.L.artemis_hip_diffusion_update, C.byref(diffusion), dt)

    def DiffusionTimestep(self, diffusion, cfl):
        t = torch.full((1,), 1.7976931348623157e308, dtype=torch.float64, device=self.dev)
        self._call(self.L.artemis_hip_diffusion_dt, C.byref(diffusion), cfl, t.data_ptr())
        return t.item()

    def TimestepAll(self, cfl_gas, cfl_dust, diffusion=None):
        """artemis_hip_timestep_all: gas (hydro + diffusive limits) and dust timestep of the pack's primitives in one pass."""
        t = torch.full((1,), 1.7976931348623157e308, dtype=torch.float64, device=self.dev)
        self._call(self.L.artemis_hip_timestep_all, cfl_gas, cfl_dust, C.byref(diffusion) if diffusion is not None else None,
                   t.data_ptr())
        return t.item()

    # ---- source-term tasks (artemis_driver.cpp:222-241) ---------------------------------------
    def ExternalGravity(self, time, dt, gravity):
        """gravity: capi.Gravity (see gravity_point / gravity_uniform below)."""
        self._call(self.L.artemis_hip_external_gravity, C.byref(gravity), time, dt)

    @staticmethod
    def _particle_array(particles):
        n = len(particles)
        arr = (capi.NBodyParticle * n)()
        for q, p in zip(arr, particles):
            q.gm = p["GM"]
            for name in ("pos", "vel", "xf", "vf"):
                for d_, v in enumerate(p.get(name, (0.0, 0.0, 0.0))):
                    getattr(q, name)[d_] = v
            q.rs, q.racc, q.gamma, q.beta = p.get("rs", 0.0), p.get("racc", 0.0), p.get("gamma", 0.0), p.get("beta", 0.0)
            q.spline, q.couple = int(p.get("spline", 0)), int(p.get("couple", 1))
        return arr

    def nbody_device(self, particles):
        """The particle dicts of NBodyGravity as a DEVICE array for stage_general(nbody=...) / nbody_force_sums."""
        arr = self._particle_array(particles)
        return self._device_records(list(arr), capi.NBodyParticle), len(particles)

    def nbody_force_sums(self, nbody, omf, dt):
        """artemis_hip_nbody_force_sums of the pack's primitives: the [npart, 7] rows (device accumulators from zero)."""
        dev, n = nbody
        rows = self.L.artemis_hip_nbody_force_scratch(C.byref(self.pack))
        scratch = torch.zeros(7 * n * rows, dtype=torch.float64, device=dev.device)
        force = torch.zeros(7 * n, dtype=torch.float64, device=dev.device)
        capi.check(self.L.artemis_hip_nbody_force_sums(C.byref(self.pack), C.c_void_p(dev.data_ptr()), n, omf, dt, None,
                                                       C.c_void_p(scratch.data_ptr()), C.c_void_p(force.data_ptr()), self._stream()))
        return force.cpu().numpy().reshape(n, 7)

    def NBodyGravity(self, time, dt, particles, omf=0.0):
        """Gravity::NBodyGravity: particles = dicts with GM, pos, vel, xf, vf, rs, racc, gamma, beta, spline, couple.
        Returns the [npart, 7] back-reaction rows of this call."""
        n = len(particles)
        arr = self._particle_array(particles)
        force = (C.c_double * (7 * n))()
        capi.check(self.L.artemis_hip_nbody_gravity(C.byref(self.pack), arr, n, omf, time, dt, force, self._stream()))
        return np.array(force[:]).reshape(n, 7)

    def viscosity_radial_table(self, D):
        """Tabulate the radial factor of D.visc (powerlaw r_exp != 0, alpha) per cell with the host
        libm (artemis_hip_diffusion_radial_fill), upload it and point D.visc.radial at it."""
        n = self.pack.nblocks
        host = np.zeros((n,) + tuple(self.gas_prim.shape[2:]))
        mh = getattr(self, "metric_host", None)
        for b in range(n):
            capi.check(self.L.artemis_hip_diffusion_radial_fill(
                C.byref(self.pack), self.geom_host.ctypes.data, mh.ctypes.data if mh is not None else None,
                C.byref(D.visc), b, host[b].ctypes.data))
        self._radial = torch.from_numpy(host).to(self.dev)
        self._radial_tab = torch.tensor([self._radial[b].data_ptr() for b in range(n)], dtype=torch.int64,
                                        device=self.dev)
        D.visc.radial = self._radial_tab.data_ptr()
        return self._radial

    def distance_table(self, D):
        """Fill the static Coords::Distance table of this pack (artemis_hip_viscous_distance_fill) and point
        D.dist at it: the viscous / thermal flux tasks then look the distances up (same bits)."""
        n = int(self.L.artemis_hip_viscous_distance_count(C.byref(self.pack)))
        self._dist = torch.zeros(n, dtype=torch.float64, device=self.dev)
        self._call(self.L.artemis_hip_viscous_distance_fill, self._dist.data_ptr())
        D.dist = self._dist.data_ptr()
        return self._dist

    def cooling_params(self, gamma, gm, beta0, beta_min=1e-12, exp_scale=0.0, tfloor=0.0, tcyl=0.0, cyl_plaw=0.0,
                       tsph=0.0, sph_plaw=0.0, mu=1.0):
        """capi.Cooling from the <cooling> keys, with the Tref / beta tables filled on the host
        (artemis_hip_cooling_table_fill) and uploaded."""
        c = capi.Cooling()
        c.beta0, c.beta_min, c.exp_scale, c.tfloor = beta0, beta_min, exp_scale, tfloor
        c.tcyl, c.cyl_plaw, c.tsph, c.sph_plaw = tcyl, cyl_plaw, tsph, sph_plaw
        c.gm = gm if gm is not None else float("nan")
        c.cv = 1.0 / ((gamma - 1.0) * 1.0 * mu)
        n = self.pack.nblocks
        shape = (n,) + tuple(self.gas_prim.shape[2:])
        t0, be = np.zeros(shape), np.zeros(shape)
        mh = getattr(self, "metric_host", None)
        for b in range(n):
            capi.check(self.L.artemis_hip_cooling_table_fill(
                C.byref(self.pack), self.geom_host.ctypes.data, mh.ctypes.data if mh is not None else None,
                C.byref(c), b, t0[b].ctypes.data, be[b].ctypes.data))
        self._cool = (torch.from_numpy(t0).to(self.dev), torch.from_numpy(be).to(self.dev))
        self._cool_tab = tuple(torch.tensor([a[b].data_ptr() for b in range(n)], dtype=torch.int64, device=self.dev)
                               for a in self._cool)
        c.tref, c.beta = self._cool_tab[0].data_ptr(), self._cool_tab[1].data_ptr()
        return c

    def CoolingSource(self, time, dt, cooling):
        self._call(self.L.artemis_hip_cooling_source, C.byref(cooling), time, dt)

    def RotatingFrameForce(self, omega, qshear, time, dt):
        self._call(self.L.artemis_hip_rotating_frame_force, omega, qshear, time, dt)

    def DragSource(self, time, dt, drag):
        self._call(self.L.artemis_hip_drag_source, C.byref(drag), time, dt)

    def stage_fused(self, gam0, gam1, beta_dt, bdt, prim_in, prim_u1, prim_out, cons_out=None,
                    pcm=False, cfl=0.0, dt_dev=None, region=0, shell_faces=0, tiny_in=None, tiny_out=None,
                    tiny_clear=None, outflow_faces=0, outflow_faces_by_block=None):
        a = capi.StageArgs()
        a.gam0, a.gam1, a.beta_dt, a.bdt, a.pcm = gam0, gam1, beta_dt, bdt, int(pcm)
        a.prim_in, a.prim_u1, a.prim_out, a.cons_out = prim_in, prim_u1, prim_out, cons_out
        a.cfl, a.dt_dev, a.region, a.shell_faces = cfl, dt_dev, region, shell_faces
        a.tiny_in, a.tiny_out, a.tiny_clear = tiny_in, tiny_out, tiny_clear  # device words (int addresses) or None
        a.outflow_faces = outflow_faces  # bit f: do not read the ghost zones behind (outflow) face f
        if outflow_faces_by_block is not None:  # ... one mask per block (a host array)
            self._ofb = (C.c_ubyte * len(outflow_faces_by_block))(*outflow_faces_by_block)
            a.outflow_faces_by_block = C.cast(self._ofb, C.c_void_p)
        self._call(self.L.artemis_hip_stage_fused, C.byref(a))

    def halo_count(self, face):
        return self.L.artemis_hip_halo_count(C.byref(self.pack), face)

    def halo_pack(self, block, face, buf):
        self._call(self.L.artemis_hip_halo_pack, block, face, C.c_void_p(buf.data_ptr()))

    def halo_unpack(self, block, face, buf):
        self._call(self.L.artemis_hip_halo_unpack, block, face, C.c_void_p(buf.data_ptr()))

    def pack_with_prim(self, prim_table):
        """Shallow copy of the C pack whose gas.prim table points at another buffer (the
        fused stage ping-pongs primitives; BCs and PrimToCons then act on that buffer)."""
        p = capi.Pack()
        C.memmove(C.byref(p), C.byref(self.pack), C.sizeof(capi.Pack))
        p.gas.prim = prim_table
        return p

    def call_on(self, pack, fn, *args):
        capi.check(fn(C.byref(pack), *args, self._stream()))


def gravity_uniform(gx1, gx2, gx3):
    g = capi.Gravity()
    g.type = capi.GRAVITY_UNIFORM
    g.g[:] = [gx1, gx2, gx3]
    g.tstart, g.tstop = -1.7976931348623157e308, 1.7976931348623157e308
    return g


def gravity_point(mass, soft=0.0, sink=0.0, sink_rate=0.0, pos=(0.0, 0.0, 0.0), G=1.0):
    g = capi.Gravity()
    g.type = capi.GRAVITY_POINT
    g.gm, g.soft, g.sink, g.sink_rate = G * mass, soft, sink, sink_rate
    g.pos[:] = list(pos)
    g.tstart, g.tstop = -1.7976931348623157e308, 1.7976931348623157e308
    return g


def binary_orbit(gm, a, e=0.0, i=0.0, omega=0.0, Omega=0.0, f=180.0):
    """Orbit (gravity.hpp:30-94): returns solve(t, omf) -> separation vector rb of the pair; angles in degrees."""
    import math
    r = lambda deg: deg * math.pi / 180.
    n = math.sqrt(gm / (a * a * a))
    coso, sino, cosI, sinI = math.cos(r(omega)), math.sin(r(omega)), math.cos(r(i)), math.sin(r(i))
    cosO, sinO, cosf0, sinf0 = math.cos(r(Omega)), math.sin(r(Omega)), math.cos(r(f)), math.sin(r(f))

    def solve(t, omf=0.0):
        sint, cost = math.sin(t * (n - omf)), math.cos(t * (n - omf))
        cosf = cosf0 * cost - sinf0 * sint
        sinf = cosf0 * sint + sinf0 * cost
        rb = a * (1.0 - e * e) / (1.0 + e * cosf)
        xb, yb = rb * cosf, rb * sinf
        cosf = xb * coso - sino * yb
        sinf = xb * sino + coso * yb
        return ((cosO * cosf - sinO * sinf * cosI), (sinO * cosf + cosO * sinf * cosI), sinf * sinI)
    return solve


def gravity_binary(mass, q, rb, soft1=0.0, soft2=0.0, sink1=0.0, sink2=0.0, sink_rate1=0.0, sink_rate2=0.0,
                   com=(0.0, 0.0, 0.0), G=1.0):
    """capi.Gravity of type BINARY for a separation vector rb (binary_mass.cpp:56-70)."""
    g = capi.Gravity()
    g.type = capi.GRAVITY_BINARY
    g.gm, g.q = G * mass, q
    mu1, mu2 = 1. / (1.0 + q), q / (1.0 + q)
    g.pos[:] = [com[n] - mu2 * rb[n] for n in range(3)]
    g.pos2[:] = [com[n] + mu1 * rb[n] for n in range(3)]
    g.soft, g.soft2, g.sink, g.sink2, g.sink_rate, g.sink_rate2 = soft1, soft2, sink1, sink2, sink_rate1, sink_rate2
    g.tstart, g.tstop = -1.7976931348623157e308, 1.7976931348623157e308
    return g


def drag_params(type="simple_dust", model="constant", tau=(), scale=1.0, grain_density=1.0, sizes=(),
                mesh_min=(0.0, 0.0, 0.0), mesh_max=(1.0, 1.0, 1.0), gas_damping=None, dust_damping=None,
                damp_visc=None):
    """capi.Drag from deck-style values; damping = dict(inner, inner_rate, outer, outer_rate);
    damp_visc = the gas viscosity (capi.DiffCoeff with its radial table) for <gas/damping> damp_to_visc."""
    d = capi.Drag()
    if damp_visc is not None:
        d._keep = damp_visc  # the C struct holds a pointer to it
        d.damp_visc = C.addressof(damp_visc)
    d.type = {"simple_dust": capi.DRAG_SIMPLE_DUST, "self": capi.DRAG_SELF}[type]
    d.model = {"constant": capi.DRAG_CONSTANT, "stokes": capi.DRAG_STOKES}[model]
    d.scale, d.grain_density = scale, grain_density
    for n, t in enumerate(tau):
        d.tau[n] = scale * t if model == "constant" else scale  # drag.hpp:129-147
    if model == "stokes":
        for n in range(len(sizes)):
            d.tau[n] = scale
    for n, sz in enumerate(sizes):
        d.sizes[n] = sz
    big = 1.7976931348623157e308
    for dst, src in ((d.gas, gas_damping), (d.dust, dust_damping)):
        src = src or {}
        dst.ix[:] = list(src.get("inner", (-big,) * 3))
        dst.ox[:] = list(src.get("outer", (big,) * 3))
        dst.irate[:] = list(src.get("inner_rate", (0.0,) * 3))
        dst.orate[:] = list(src.get("outer_rate", (0.0,) * 3))
    d.xmin[:] = list(mesh_min)
    d.xmax[:] = list(mesh_max)
    return d


def diffusion_params(gamma, viscosity=None, conductivity=None, mu=1.0):
    """capi.Diffusion from deck-style dicts: viscosity = dict(type="constant", nu=..., eta_bulk=0,
    r_exp=0, averaging="arithmetic"), conductivity = dict(type="conductivity"|"diffusivity", cond= |
    kappa=, temp_exp=0, rho_exp=0, ...).  cv = kB/((gamma-1) amu mu) in scale-free units."""
    d = capi.Diffusion()
    d.cv = 1.0 / ((gamma - 1.0) * 1.0 * mu)
    if viscosity:
        t = {"constant": capi.VISCOSITY_PLAW, "powerlaw": capi.VISCOSITY_PLAW, "alpha": capi.VISCOSITY_ALPHA}[viscosity.get("type", "constant")]
        d.visc.type, d.visc.avg = t, {"arithmetic": 0, "harmonic": 1}[viscosity.get("averaging", "arithmetic")]
        d.visc.coeff = viscosity.get("nu", viscosity.get("alpha", 0.0))
        d.visc.eta, d.visc.r_exp, d.visc.r0 = viscosity.get("eta_bulk", 0.0), viscosity.get("r_exp", 0.0), viscosity.get("r0", 1.0)
        d.visc.omega0 = viscosity.get("Omega0", 0.0)  # sqrt(gm / r0^3), diffusion_coeff.hpp:117-119
        d.visc.rho_ref = d.visc.T_ref = 1.0
    if conductivity:
        t = {"conductivity": capi.CONDUCTIVITY_PLAW, "diffusivity": capi.THERMALDIFF_PLAW}[conductivity.get("type", "conductivity")]
        d.cond.type, d.cond.avg = t, {"arithmetic": 0, "harmonic": 1}[conductivity.get("averaging", "arithmetic")]
        d.cond.coeff = conductivity.get("cond", conductivity.get("kappa", 0.0))
        d.cond.temp_exp, d.cond.rho_exp = conductivity.get("temp_exp", 0.0), conductivity.get("rho_exp", 0.0)
        d.cond.rho_ref, d.cond.T_ref, d.cond.r0 = conductivity.get("rho_ref", 1.0), conductivity.get("T_ref", 1.0), 1.0
    return d
