// TEST DOUBLE of libartemis_hip.so for machines without a GPU.
//
// It exports the C ABI of include/artemis_hip.h and include/artemis_rt.h on HOST memory, each
// entry point implemented with the CPU oracle (oracle/artemis_oracle.cpp is compiled into this
// library).  Linked with the UNMODIFIED product driver sources (artemis_amd/csrc/driver/*.cpp)
// it lets the `-m "not gpu"` tests exercise the host logic -- deck parsing, mesh-block
// decomposition, ghost-slab links, message tags, the dt all-reduce, the step loop -- under
// torch.distributed/gloo with world_size > 1.  It lives under tests/ and is never built into,
// loaded by, or shipped with the product.
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../oracle/artemis_oracle.cpp"

#include "../../include/artemis_hip.h"
#include "../../include/artemis_rt.h"

namespace {
thread_local std::string d_err;

oracle_cfg cfg_of(const artemis_pack_t *p) {
  oracle_cfg c;
  std::memset(&c, 0, sizeof c);
  c.nx1 = p->nx1, c.nx2 = p->nx2, c.nx3 = p->nx3, c.ng = p->nghost;
  c.ns_gas = p->gas.nspecies, c.ns_dust = p->dust.nspecies;
  c.recon_gas = p->gas.recon, c.riemann_gas = p->gas.riemann;
  c.recon_dust = p->dust.recon, c.riemann_dust = p->dust.riemann;
  c.gamma = p->gm1 + 1.0;
  c.dfloor_gas = p->gas.dfloor, c.siefloor_gas = p->gas.siefloor, c.de_switch = p->gas.de_switch;
  c.dfloor_dust = p->dust.dfloor;
  c.cfl_gas = 1.0, c.cfl_dust = 1.0;
  for (int i = 0; i < 6; ++i) c.bc[i] = BC_NONE;
  c.integrator = INT_RK2;
  c.coords = p->coords;
  return c;
}

// An oracle Sim whose geometry is block b's; arrays are copied in/out per call.
struct Bound {
  Sim *s;
  const artemis_pack_t *p;
  int b;
  Bound(const artemis_pack_t *p_, int b_) : p(p_), b(b_) {
    oracle_cfg c = cfg_of(p);
    // block bounds from the geom table: f0 = xmin - g*dx
    const double *g = p->geom + 6 * b;
    const int nx[3] = {p->nx1, p->nx2, p->nx3};
    const int ndim = (p->nx3 > 1) ? 3 : ((p->nx2 > 1) ? 2 : 1);
    double lo[3], hi[3];
    for (int d = 0; d < 3; ++d) {
      const int gh = (d < ndim) ? p->nghost : 0;
      lo[d] = g[2 * d] + gh * g[2 * d + 1];
      hi[d] = lo[d] + nx[d] * g[2 * d + 1];
    }
    c.x1min = lo[0], c.x1max = hi[0], c.x2min = lo[1], c.x2max = hi[1], c.x3min = lo[2], c.x3max = hi[2];
    s = static_cast<Sim *>(oracle_create(&c));
    // use the table's f0/dx verbatim so cell edges match the product's kernels bit for bit
    for (int d = 0; d < 3; ++d) s->f0[d] = g[2 * d], s->dx[d] = g[2 * d + 1];
    if (p->omega_frame != 0.0) s->rframe.on = true, s->rframe.omega = p->omega_frame; // FluxSource's vf
  }
  ~Bound() { oracle_destroy(s); }
  void in(RVec &dst, double *const *tab, int nvar) {
    if (!tab) return;
    for (int v = 0; v < nvar; ++v) std::memcpy(dst.data() + v * s->N, tab[b * nvar + v], s->N * sizeof(Real));
  }
  void out(const RVec &src, double *const *tab, int nvar) {
    if (!tab) return;
    for (int v = 0; v < nvar; ++v) std::memcpy(tab[b * nvar + v], src.data() + v * s->N, s->N * sizeof(Real));
  }
  void load_state() {
    in(s->gprim, p->gas.prim, s->nvg), in(s->gu0, p->gas.cons0, s->nvg), in(s->gu1, p->gas.cons1, s->nvg);
    in(s->dprim, p->dust.prim, s->nvd), in(s->du0, p->dust.cons0, s->nvd), in(s->du1, p->dust.cons1, s->nvd);
  }
  void load_fluxes() {
    for (int d = 0; d < 3; ++d) {
      in(s->gflux[d], p->gas.flux[d], s->nvg), in(s->gpflux[d], p->gas.pflux[d], s->c.ns_gas);
      in(s->gvface[d], p->gas.vface[d], s->c.ns_gas), in(s->dflux[d], p->dust.flux[d], s->nvd);
    }
  }
};
int bad(const char *m) {
  d_err = m;
  return ARTEMIS_HIP_EINVAL;
}
// abi.hip validate_registers: the same contract on the host
int need_registers(const artemis_pack_t *p, const char *task) {
  for (const artemis_fluid_pack_t *f : {&p->gas, &p->dust})
    if (f->nspecies > 0 && (!f->cons0 || !f->cons1))
      return bad((std::string(task) + " needs the cons0 (u0) and cons1 (u1) tables of every fluid in the pack").c_str());
  return 0;
}
} // namespace

extern "C" {
const char *artemis_hip_last_error(void) { return d_err.c_str(); }
const char *artemis_hip_version(void) { return "artemis_hip CPU TEST DOUBLE (tests/cpu_double)"; }
int artemis_hip_device_count(void) { return 0; }

int artemis_hip_calculate_fluxes(const artemis_pack_t *p, int fluid, int pcm, void *) {
  if (!p) return bad("null pack");
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state();
    calculate_fluxes(*B.s, fluid, pcm != 0);
    for (int d = 0; d < B.s->ndim; ++d) {
      if (fluid == FL_GAS) {
        B.out(B.s->gflux[d], p->gas.flux[d], B.s->nvg), B.out(B.s->gpflux[d], p->gas.pflux[d], p->gas.nspecies);
        B.out(B.s->gvface[d], p->gas.vface[d], p->gas.nspecies);
      } else {
        B.out(B.s->dflux[d], p->dust.flux[d], B.s->nvd);
      }
    }
  }
  return 0;
}
int artemis_hip_apply_update(const artemis_pack_t *p, double g0, double g1, double bdt, void *) {
  if (int rc = need_registers(p, "ApplyUpdate")) return rc;
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state(), B.load_fluxes();
    apply_update(*B.s, g0, g1, bdt);
    B.out(B.s->gu0, p->gas.cons0, B.s->nvg), B.out(B.s->du0, p->dust.cons0, B.s->nvd);
  }
  return 0;
}
int artemis_hip_flux_source(const artemis_pack_t *p, int fluid, double dt, void *) {
  const bool gas = (fluid == FL_GAS);
  if ((gas ? p->gas.nspecies : p->dust.nspecies) == 0) return 0;
  if (!gas && p->coords == CO_CART) return 0;
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state(), B.load_fluxes();
    // interior only, like the product (the oracle's literal [is-2, ie+1] range only scribbles
    // on ghost cells that PrimToCons overwrites)
    Sim &s = *B.s;
    RVec &u0 = gas ? s.gu0 : s.du0;
    const int nv = gas ? s.nvg : s.nvd;
    RVec before = u0;
    flux_source(s, fluid, dt);
    for (int v = 0; v < nv; ++v)
      for (int k = 0; k < s.nk; ++k)
        for (int j = 0; j < s.nj; ++j)
          for (int i = 0; i < s.ni; ++i)
            if (i < s.is || i > s.ie) u0[v * s.N + IDX(s, k, j, i)] = before[v * s.N + IDX(s, k, j, i)];
    B.out(u0, gas ? p->gas.cons0 : p->dust.cons0, nv);
  }
  return 0;
}
// The double evaluates the metric with the oracle's own per-cell libm calls, but the driver's
// host-side problem generators read the tables, so fill them the way the product does.
long artemis_hip_metric_count(const artemis_pack_t *p) {
  if (p->coords == CO_CART || p->coords == CO_SPH1D) return 0;
  const int nj = p->nx2 + ((p->nx2 > 1) ? 2 * p->nghost : 0);
  const int nk = p->nx3 + ((p->nx3 > 1) ? 2 * p->nghost : 0);
  return static_cast<long>(p->nblocks) * (6L * (nj + 1) + 2L * (nk + 1));
}
// (the double evaluates PLM_G through the oracle: the table is accepted and ignored)
long artemis_hip_plm_table_count(const artemis_pack_t *p) {
  const int g1 = p->nghost, g2 = (p->nx2 > 1) ? p->nghost : 0, g3 = (p->nx3 > 1) ? p->nghost : 0;
  const int ni = p->nx1 + 2 * g1, nj = p->nx2 + 2 * g2, nk = p->nx3 + 2 * g3;
  return static_cast<long>(p->nblocks) * 3 * 9 * std::max(ni, std::max(nj, nk));
}
int artemis_hip_plm_table_fill(const artemis_pack_t *p, double *table, void *) {
  std::memset(table, 0, sizeof(double) * artemis_hip_plm_table_count(p));
  return 0;
}
int artemis_hip_metric_fill(const artemis_pack_t *p, const double *g, double *out) {
  if (artemis_hip_metric_count(p) == 0) return 0;
  const int nj = p->nx2 + ((p->nx2 > 1) ? 2 * p->nghost : 0), st = nj + 1;
  const int nk = p->nx3 + ((p->nx3 > 1) ? 2 * p->nghost : 0), st3 = nk + 1;
  const long stride = 6L * st + 2L * st3;
  for (int b = 0; b < p->nblocks; ++b) {
    double *m = out + b * stride;
    for (long q = 0; q < stride; ++q) m[q] = 0.0;
    if (p->coords == CO_SPH2D || p->coords == CO_SPH3D) {
      for (int j = 0; j <= nj; ++j) {
        const double xf = g[6 * b + 2] + j * g[6 * b + 3];
        m[j] = std::cos(xf), m[st + j] = std::sin(xf);
      }
      for (int j = 0; j < nj; ++j) {
        BBox bb{};
        bb.x2[0] = g[6 * b + 2] + j * g[6 * b + 3], bb.x2[1] = g[6 * b + 2] + (j + 1) * g[6 * b + 3];
        const Real ctm = std::cos(bb.x2[0]), ctp = std::cos(bb.x2[1]);
        const Real dst = std::sin(bb.x2[1]) - std::sin(bb.x2[0]);
        const Real x2v = (dst - bb.x2[1] * ctp + bb.x2[0] * ctm) / std::abs(ctm - ctp);
        m[2 * st + j] = x2v, m[3 * st + j] = std::sin(x2v), m[4 * st + j] = std::sin(0.5 * (bb.x2[0] + bb.x2[1]));
        m[5 * st + j] = std::cos(x2v);
      }
    } else if (p->coords == CO_CYL) {
      for (int j = 0; j < nj; ++j) {
        const Real x2v = 0.5 * ((g[6 * b + 2] + j * g[6 * b + 3]) + (g[6 * b + 2] + (j + 1) * g[6 * b + 3]));
        m[2 * st + j] = x2v, m[3 * st + j] = std::sin(x2v), m[5 * st + j] = std::cos(x2v);
      }
    }
    if (p->coords == CO_SPH3D || p->coords == CO_AXI)
      for (int k = 0; k < nk; ++k) {
        const Real x3v = 0.5 * ((g[6 * b + 4] + k * g[6 * b + 5]) + (g[6 * b + 4] + (k + 1) * g[6 * b + 5]));
        m[6L * st + k] = std::cos(x3v), m[6L * st + st3 + k] = std::sin(x3v);
      }
  }
  return 0;
}
int artemis_hip_set_aux(const artemis_pack_t *p, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state();
    set_aux(*B.s);
    B.out(B.s->gu0, p->gas.cons0, B.s->nvg);
  }
  return 0;
}
int artemis_hip_cons_to_prim(const artemis_pack_t *p, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state();
    cons_to_prim(*B.s);
    B.out(B.s->gprim, p->gas.prim, B.s->nvg), B.out(B.s->dprim, p->dust.prim, B.s->nvd);
  }
  return 0;
}
int artemis_hip_prim_to_cons_ghosts(const artemis_pack_t *p, void *s) { return artemis_hip_prim_to_cons(p, s); } // (idempotent on the interior)
int artemis_hip_prim_to_cons(const artemis_pack_t *p, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state();
    prim_to_cons(*B.s);
    B.out(B.s->gprim, p->gas.prim, B.s->nvg), B.out(B.s->dprim, p->dust.prim, B.s->nvd);
    B.out(B.s->gu0, p->gas.cons0, B.s->nvg), B.out(B.s->du0, p->dust.cons0, B.s->nvd);
  }
  return 0;
}
int artemis_hip_deep_copy_conserved(const artemis_pack_t *p, void *) {
  if (int rc = need_registers(p, "DeepCopyConservedData")) return rc;
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state();
    B.out(B.s->gu0, p->gas.cons1, B.s->nvg), B.out(B.s->du0, p->dust.cons1, B.s->nvd);
  }
  return 0;
}
int artemis_hip_estimate_dt_async(const artemis_pack_t *p, int fluid, double cfl, double *dt_dev, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state();
    if ((fluid == FL_GAS ? p->gas.nspecies : p->dust.nspecies) == 0) continue;
    *dt_dev = std::min(*dt_dev, cfl * estimate_dt(*B.s, fluid)); // Bound's cfl is 1
  }
  return 0;
}
int artemis_hip_estimate_dt(const artemis_pack_t *p, int fluid, double cfl, double *out, void *s) {
  *out = DBL_MAX;
  return artemis_hip_estimate_dt_async(p, fluid, cfl, out, s);
}
int artemis_hip_apply_bc(const artemis_pack_t *p, const int *bc, const artemis_bc_params_t *par, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state();
    for (int f = 0; f < 6; ++f) B.s->c.bc[f] = bc[6 * b + f];
    if (par) {
      Sim &s = *B.s;
      s.strat.q = par->qshear, s.strat.Om0 = par->omega;
      s.condbc.g_temp = par->cond_temp, s.condbc.flux = par->cond_flux;
      s.grav.type = 1;
      for (int d = 0; d < 3; ++d) s.grav.g[d] = par->cond_g[d];
      s.cond.type = par->cond_type, s.cond.hcond_0 = s.cond.kappa_0 = par->cond_coeff;
      s.cond.temp_exp = par->cond_temp_exp, s.cond.rho_exp = par->cond_rho_exp;
      s.cond.T0 = (par->cond_T_ref > 0.0) ? par->cond_T_ref : 1.0, s.cond.d0 = (par->cond_rho_ref > 0.0) ? par->cond_rho_ref : 1.0;
      if (par->cond_cv > 0.0) s.cv = par->cond_cv;
      s.disk.omf = par->disk_omf, s.disk.nu0 = par->disk_nu0, s.disk.nu_indx = par->disk_nu_indx;
      s.disk.r0 = par->disk_r0, s.disk.mdot = par->disk_mdot;
      if (par->ic_gas) s.ic_g.resize(s.gprim.size()), B.in(s.ic_g, par->ic_gas, s.nvg);
      if (par->ic_dust) s.ic_d.resize(s.dprim.size()), B.in(s.ic_d, par->ic_dust, s.nvd);
    }
    apply_bcs(*B.s);
    if (par && par->floor_ghosts) { // PrimToCons's primitive floors on the ghost zones
      Sim &s = *B.s;
      for (int k = 0; k < s.nk; ++k)
        for (int j = 0; j < s.nj; ++j)
          for (int i = 0; i < s.ni; ++i) {
            if (i >= s.is && i <= s.ie && j >= s.js && j <= s.je && k >= s.ks && k <= s.ke) continue;
            const size_t c = IDX(s, k, j, i);
            for (int n = 0; n < s.c.ns_gas; ++n) {
              Real &w_d = s.gprim[n * s.N + c], &w_s = s.gprim[(5 * s.c.ns_gas + n) * s.N + c];
              w_d = (w_d > s.c.dfloor_gas) ? w_d : s.c.dfloor_gas;
              w_s = (w_s > s.c.siefloor_gas) ? w_s : s.c.siefloor_gas;
            }
            for (int n = 0; n < s.c.ns_dust; ++n) {
              Real &w_d = s.dprim[n * s.N + c];
              w_d = (w_d > s.c.dfloor_dust) ? w_d : s.c.dfloor_dust;
            }
          }
    }
    B.out(B.s->gprim, p->gas.prim, B.s->nvg), B.out(B.s->dprim, p->dust.prim, B.s->nvd);
  }
  return 0;
}
int artemis_hip_external_gravity(const artemis_pack_t *p, const artemis_gravity_t *g, double time,
                                 double dt, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state();
    Sim &s = *B.s;
    s.grav.type = g->type, s.grav.gm = g->gm, s.grav.soft = g->soft, s.grav.sink = g->sink;
    s.grav.sink_rate = g->sink_rate, s.grav.tstart = g->tstart, s.grav.tstop = g->tstop;
    for (int d = 0; d < 3; ++d) s.grav.g[d] = g->g[d], s.grav.pos[d] = g->pos[d], s.grav.pos2[d] = g->pos2[d];
    s.grav.q = g->q, s.grav.soft2 = g->soft2, s.grav.sink2 = g->sink2, s.grav.sink_rate2 = g->sink_rate2;
    s.grav.given_pos = true;
    external_gravity(s, time, dt);
    B.out(s.gu0, p->gas.cons0, s.nvg), B.out(s.du0, p->dust.cons0, s.nvd);
  }
  return 0;
}
static void set_nbody(Sim &s, const artemis_nbody_particle_t *pl, int npart, double omf) {
  s.grav.type = 4;
  s.nbody.assign(npart, Sim::NBodyParticle());
  for (int n = 0; n < npart; ++n) {
    Sim::NBodyParticle &q = s.nbody[n];
    q.GM = pl[n].gm, q.rs = pl[n].rs, q.racc = pl[n].racc, q.gamma = pl[n].gamma, q.beta = pl[n].beta;
    q.spline = pl[n].spline, q.couple = pl[n].couple;
    for (int d = 0; d < 3; ++d) q.pos[d] = pl[n].pos[d], q.vel[d] = pl[n].vel[d], q.xf[d] = pl[n].xf[d], q.vf[d] = pl[n].vf[d];
  }
  s.rframe.on = (omf != 0.0), s.rframe.omega = omf, s.nbody_frame_correction = true;
  s.pforce.assign(static_cast<size_t>(7) * npart, 0.0);
}
// A stage kernel reads the primitives as the caller left them; the oracle's PrimToCons floors them in place.  Where a
// ghost zone arrives below a floor -- a caller that left out the floors behind a refined-mesh fill
// (artemis_hip_ml_floor_ghosts) -- the double must not repair that silently: the supplied value goes back in, so that
// the run parts from the reference's the way the device's would (sane values only: never-written corner zones stay
// floored, nothing reads them into an active zone).
static void prim_to_cons_as_supplied(Sim &s) {
  const int nsg = s.c.ns_gas, nsd = s.c.ns_dust;
  RVec rho(s.gprim.begin(), s.gprim.begin() + static_cast<size_t>(nsg) * s.N);
  RVec sie(s.gprim.begin() + static_cast<size_t>(5 * nsg) * s.N, s.gprim.begin() + static_cast<size_t>(6 * nsg) * s.N);
  RVec drho(s.dprim.begin(), s.dprim.begin() + static_cast<size_t>(nsd) * s.N);
  prim_to_cons(s);
  auto back = [](Real raw, Real &now) {
    if (std::isfinite(raw) && raw > 0.0 && raw < now) now = raw;
  };
  for (int k = 0; k < s.nk; ++k)
    for (int j = 0; j < s.nj; ++j)
      for (int i = 0; i < s.ni; ++i) {
        if (i >= s.is && i <= s.ie && j >= s.js && j <= s.je && k >= s.ks && k <= s.ke) continue;
        const size_t c = IDX(s, k, j, i);
        for (int n = 0; n < nsg; ++n) back(rho[n * s.N + c], s.gprim[n * s.N + c]), back(sie[n * s.N + c], s.gprim[(5 * nsg + n) * s.N + c]);
        for (int n = 0; n < nsd; ++n) back(drho[n * s.N + c], s.dprim[n * s.N + c]);
      }
}
int artemis_hip_nbody_force_scratch(const artemis_pack_t *) { return 1; }
// the seven sums alone: the task on a copy of the block (its fluid update is discarded)
int artemis_hip_nbody_force_sums(const artemis_pack_t *p, const artemis_nbody_particle_t *pl, int npart, double omf, double dt,
                                 const double *dt_dev, double *, double *force, void *) {
  if (dt_dev) dt = *dt_dev;
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    Sim &s = *B.s;
    B.in(s.gprim, p->gas.prim, s.nvg), B.in(s.dprim, p->dust.prim, s.nvd);
    prim_to_cons(s);
    set_nbody(s, pl, npart, omf);
    s.grav.tstart = -1e300, s.grav.tstop = 1e300;
    nbody_gravity(s, 0.0, dt);
    for (int q = 0; q < 7 * npart; ++q) force[q] += s.pforce[q];
  }
  return 0;
}
int artemis_hip_nbody_gravity(const artemis_pack_t *p, const artemis_nbody_particle_t *pl, int npart, double omf, double time,
                              double dt, double *force, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    Sim &s = *B.s;
    B.load_state();
    set_nbody(s, pl, npart, omf);
    nbody_gravity(s, time, dt);
    for (int q = 0; q < 7 * npart; ++q) force[q] += s.pforce[q];
    B.out(s.gu0, p->gas.cons0, s.nvg), B.out(s.du0, p->dust.cons0, s.nvd);
  }
  return 0;
}
int artemis_hip_rotating_frame_force(const artemis_pack_t *p, double omega, double qshear, double,
                                     double dt, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state();
    B.s->rframe.on = true, B.s->rframe.omega = omega, B.s->rframe.qshear = qshear;
    if (p->coords != ARTEMIS_CARTESIAN) B.load_fluxes(); // RotatingFrameImpl reads the mass fluxes
    rotating_frame_force(*B.s, dt);
    B.out(B.s->gu0, p->gas.cons0, B.s->nvg), B.out(B.s->du0, p->dust.cons0, B.s->nvd);
  }
  return 0;
}
static void set_cooling(Sim &s, const artemis_cooling_t *c) {
  s.cool.on = true, s.cool.beta0 = c->beta0, s.cool.beta_min = c->beta_min, s.cool.escale = c->exp_scale;
  s.cool.tfloor = c->tfloor, s.cool.tcyl = c->tcyl, s.cool.cyl_plaw = c->cyl_plaw, s.cool.tsph = c->tsph;
  s.cool.sph_plaw = c->sph_plaw, s.cv = c->cv;
  s.grav.type = (c->gm == c->gm) ? 2 : 0, s.grav.gm = c->gm;
}
int artemis_hip_cooling_table_fill(const artemis_pack_t *, const double *, const double *, const artemis_cooling_t *,
                                   int, double *, double *) {
  return 0; // the oracle evaluates Tref and beta per cell itself
}
int artemis_hip_cooling_source(const artemis_pack_t *p, const artemis_cooling_t *c, double time, double dt, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state();
    set_cooling(*B.s, c);
    cooling_source(*B.s, time, dt);
    B.out(B.s->gu0, p->gas.cons0, B.s->nvg);
  }
  return 0;
}
static void set_damp_visc(Sim &s, const artemis_drag_t *d) { // <gas/damping> damp_to_visc
  s.drag.damp_to_visc = d->damp_visc != nullptr;
  if (!d->damp_visc) return;
  const artemis_diffcoeff_t &c = *d->damp_visc;
  s.visc.type = c.type, s.visc.avg = c.avg, s.visc.nu_s = s.visc.alpha = c.coeff;
  s.visc.eta = c.eta, s.visc.r_exp = c.r_exp, s.visc.R0 = c.r0, s.visc.Omega0 = c.omega0;
}
int artemis_hip_drag_source(const artemis_pack_t *p, const artemis_drag_t *d, double, double dt, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state();
    Sim &s = *B.s;
    s.drag.type = d->type, s.drag.model = d->model, s.drag.scale = d->scale;
    s.drag.grain_density = d->grain_density;
    s.drag.tau.assign(d->tau, d->tau + s.c.ns_dust), s.drag.sizes.assign(d->sizes, d->sizes + s.c.ns_dust);
    for (int i = 0; i < 3; ++i) {
      s.drag.gas.ix[i] = d->gas.ix[i], s.drag.gas.ox[i] = d->gas.ox[i];
      s.drag.gas.irate[i] = d->gas.irate[i], s.drag.gas.orate[i] = d->gas.orate[i];
      s.drag.dust.ix[i] = d->dust.ix[i], s.drag.dust.ox[i] = d->dust.ox[i];
      s.drag.dust.irate[i] = d->dust.irate[i], s.drag.dust.orate[i] = d->dust.orate[i];
    }
    s.gx1min = d->xmin[0], s.gx2min = d->xmin[1], s.gx3min = d->xmin[2];
    s.gx1max = d->xmax[0], s.gx2max = d->xmax[1], s.gx3max = d->xmax[2];
    set_damp_visc(s, d);
    drag_source(s, dt);
    B.out(s.gu0, p->gas.cons0, s.nvg), B.out(s.du0, p->dust.cons0, s.nvd);
  }
  return 0;
}

// The fused-stage CONTRACT restated with the unfused oracle chain: u0 := PrimToCons(prim_in),
// u1 := PrimToCons(prim_u1), then the reference's task order; prim_out receives the interior.
size_t artemis_hip_redo_scratch_bytes(const artemis_pack_t *) { return 64; }
int artemis_hip_stage_fused_redo_shell(const artemis_pack_t *, const artemis_stage_args_t *, void *) { return 0; } // (the double is exact everywhere)
int artemis_hip_stage_fused(const artemis_pack_t *p, const artemis_stage_args_t *a, void *) {
  if (p->gas.nspecies != 1 || p->dust.nspecies != 0) {
    d_err = "fused stage: one gas species, no dust";
    return ARTEMIS_HIP_EUNSUPPORTED;
  }
  if (p->gas.recon == RC_PPM && !a->pcm) {
    d_err = "fused stage: pcm|plm";
    return ARTEMIS_HIP_EUNSUPPORTED;
  }
  if (a->prim_in == a->prim_out) return bad("prim_out must not alias prim_in");
  // region 1 (shell, "rounded out") is the whole block here, so region 2 (the rest) is empty
  if (a->region == 2) return 0;
  if (a->shell_done) { // synchronous double: everything is done when the call returns
    *a->shell_done = 1u;
    if (a->shell_target) *a->shell_target = 1u;
  }
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    Sim &s = *B.s;
    B.in(s.gprim, a->prim_u1, 6);
    prim_to_cons(s);
    s.gu1 = s.gu0;
    B.in(s.gprim, a->prim_in, 6);
    const int outflow = a->outflow_faces_by_block ? a->outflow_faces_by_block[b] : a->outflow_faces;
    if (outflow) { // the ghost zones behind these faces are not read: they are the edge zones (outflow)
      for (int f = 0; f < 6; ++f) s.c.bc[f] = ((outflow >> f) & 1) ? BC_OUTFLOW : BC_NONE;
      apply_bcs(s);
      for (int f = 0; f < 6; ++f) s.c.bc[f] = BC_NONE;
    }
    prim_to_cons_as_supplied(s);
    calculate_fluxes(s, FL_GAS, a->pcm != 0);
    const double beta_dt = a->beta_dt_dev ? *a->beta_dt_dev : a->beta_dt;
    const double bdt = a->beta_dt_dev ? *a->beta_dt_dev : a->bdt;
    apply_update(s, a->gam0, a->gam1, beta_dt);
    flux_source(s, FL_GAS, bdt);
    set_aux(s);
    cons_to_prim(s);
    // pressure of the interior cells as PrimToCons would set it
    const Real gm1 = p->gm1;
    for (int k = s.ks; k <= s.ke; ++k)
      for (int j = s.js; j <= s.je; ++j)
        for (int i = s.is; i <= s.ie; ++i) {
          const size_t c = IDX(s, k, j, i);
          s.gprim[4 * s.N + c] = std::max(0.0, gm1 * s.gprim[0 * s.N + c] * s.gprim[5 * s.N + c]);
        }
    for (int v = 0; v < 6; ++v)
      for (int k = s.ks; k <= s.ke; ++k)
        for (int j = s.js; j <= s.je; ++j)
          std::memcpy(a->prim_out[b * 6 + v] + IDX(s, k, j, s.is), s.gprim.data() + v * s.N + IDX(s, k, j, s.is),
                      (s.ie - s.is + 1) * sizeof(Real));
    if (a->cons_out) {
      prim_to_cons(s);
      for (int v = 0; v < 6; ++v)
        for (int k = s.ks; k <= s.ke; ++k)
          for (int j = s.js; j <= s.je; ++j)
            std::memcpy(a->cons_out[b * 6 + v] + IDX(s, k, j, s.is), s.gu0.data() + v * s.N + IDX(s, k, j, s.is),
                        (s.ie - s.is + 1) * sizeof(Real));
    }
    if (a->dt_dev) *a->dt_dev = std::min(*a->dt_dev, a->cfl * estimate_dt(s, FL_GAS));
  }
  return 0;
}

// The general-stage CONTRACT restated with the oracle's task chain (both fluids, every source
// package): u1 := PrimToCons(*_u1), u0 := PrimToCons(*_in), the reference's task order, interior
// of the new primitives (rho, v, sie) to *_out.
// gas diffusion through the oracle's restatement
static void set_diffusion(Sim &s, const artemis_diffusion_t *d) {
  auto cp = [](Sim::DiffCoeff &o, const artemis_diffcoeff_t &c) {
    o.type = c.type, o.avg = c.avg;
    o.nu_s = o.alpha = o.hcond_0 = o.kappa_0 = c.coeff;
    o.eta = c.eta, o.r_exp = c.r_exp, o.R0 = c.r0, o.Omega0 = c.omega0;
    o.temp_exp = c.temp_exp, o.rho_exp = c.rho_exp, o.d0 = c.rho_ref, o.T0 = c.T_ref;
  };
  cp(s.visc, d->visc), cp(s.cond, d->cond);
  s.cv = d->cv;
}
static void dflux_io(Bound &B, const artemis_pack_t *p, bool out) {
  for (int d = 0; d < 3; ++d) {
    if (!p->gas.diff_flux[d]) continue;
    if (out) B.out(B.s->qflux[d], p->gas.diff_flux[d], 4 * B.s->c.ns_gas);
    else B.in(B.s->qflux[d], p->gas.diff_flux[d], 4 * B.s->c.ns_gas);
  }
}
// the cell-local remainder of a stage over the stored fluxes, through the oracle's tasks
static int stage_epilogue_common(const artemis_pack_t *p, const artemis_stage_general_args_t *a, bool to_cons) {
  if (a->drag) return bad("stage epilogue: drag is not part of it");
  if (to_cons && a->cooling) return bad("stage epilogue (cons): cooling acts after drag");
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    Sim &s = *B.s;
    B.load_state(), B.load_fluxes();
    apply_update(s, a->gam0, a->gam1, a->beta_dt);
    flux_source(s, FL_GAS, a->bdt), flux_source(s, FL_DUST, a->bdt);
    if (a->diffusion) {
      dflux_io(B, p, false), set_diffusion(s, a->diffusion);
      diffusion_update(s, a->bdt);
    }
    if (a->gravity) {
      const artemis_gravity_t *g = a->gravity;
      s.grav.type = g->type, s.grav.gm = g->gm, s.grav.soft = g->soft, s.grav.sink = g->sink;
      s.grav.sink_rate = g->sink_rate, s.grav.tstart = g->tstart, s.grav.tstop = g->tstop;
      for (int d = 0; d < 3; ++d) s.grav.g[d] = g->g[d], s.grav.pos[d] = g->pos[d], s.grav.pos2[d] = g->pos2[d];
      s.grav.q = g->q, s.grav.soft2 = g->soft2, s.grav.sink2 = g->sink2, s.grav.sink_rate2 = g->sink_rate2;
      s.grav.given_pos = true;
      external_gravity(s, a->time, a->bdt);
    }
    if (a->rf_omega != 0.0) {
      s.rframe.on = true, s.rframe.omega = a->rf_omega, s.rframe.qshear = a->rf_qshear;
      rotating_frame_force(s, a->bdt);
    }
    if (a->cooling) {
      const int gtype = s.grav.type;
      set_cooling(s, a->cooling);
      cooling_source(s, a->time, a->bdt);
      s.grav.type = gtype;
    }
    if (to_cons) { // artemis_hip_stage_epilogue_cons: the state stays conserved, in cons0
      B.out(s.gu0, p->gas.cons0, s.nvg), B.out(s.du0, p->dust.cons0, s.nvd);
      continue;
    }
    set_aux(s);
    cons_to_prim(s);
    B.out(s.gprim, p->gas.prim, s.nvg), B.out(s.dprim, p->dust.prim, s.nvd);
    B.out(s.gu0, p->gas.cons0, s.nvg), B.out(s.du0, p->dust.cons0, s.nvd);
  }
  return 0;
}
int artemis_hip_stage_epilogue(const artemis_pack_t *p, const artemis_stage_general_args_t *a, void *) {
  return stage_epilogue_common(p, a, false);
}
int artemis_hip_stage_epilogue_cons(const artemis_pack_t *p, const artemis_stage_general_args_t *a, void *) {
  return stage_epilogue_common(p, a, true);
}
int artemis_hip_stage_finish(const artemis_pack_t *p, const artemis_drag_t *drag, double time, double dt, void *st) {
  if (drag)
    if (int rc = artemis_hip_drag_source(p, drag, time, dt, st)) return rc;
  if (p->gas.nspecies)
    if (int rc = artemis_hip_set_aux(p, st)) return rc;
  return artemis_hip_cons_to_prim(p, st);
}
// ... of the listed zones only (defer_finish = 2: the stage has finished every zone, the fix-up has rewritten the
// conserved state of its listed zones): the block's cons0 through the same three tasks, the listed zones' primitives out
int artemis_hip_stage_finish_cells(const artemis_pack_t *p, const artemis_drag_t *d, double, double dt, const artemis_ml_fix_cell_t *cells,
                                   int ncells, void *) {
  if (!d) return bad("stage finish of listed zones: drag is required");
  std::map<int, std::vector<const artemis_ml_fix_cell_t *>> per_block;
  for (int q = 0; q < ncells; ++q) per_block[cells[q].block].push_back(&cells[q]);
  for (auto &kv : per_block) {
    const int b = kv.first;
    Bound B(p, b);
    Sim &s = *B.s;
    B.in(s.gu0, p->gas.cons0, s.nvg), B.in(s.du0, p->dust.cons0, s.nvd);
    // (zones outside the list hold whatever the tables held: the tasks are pointwise and only listed zones are copied out;
    //  give them a valid state so that nothing traps on the way)
    {
      std::vector<char> listed(s.N, 0);
      for (const artemis_ml_fix_cell_t *fc : kv.second) listed[IDX(s, fc->k, fc->j, fc->i)] = 1;
      for (long c = 0; c < static_cast<long>(s.N); ++c) {
        if (listed[c]) continue;
        for (int v = 0; v < s.nvg; ++v) s.gu0[v * s.N + c] = (v < s.c.ns_gas || v >= 4 * s.c.ns_gas) ? 1.0 : 0.0;
        for (int v = 0; v < s.nvd; ++v) s.du0[v * s.N + c] = (v < s.c.ns_dust) ? 1.0 : 0.0;
      }
    }
    s.drag.type = d->type, s.drag.model = d->model, s.drag.scale = d->scale;
    s.drag.grain_density = d->grain_density;
    s.drag.tau.assign(d->tau, d->tau + s.c.ns_dust), s.drag.sizes.assign(d->sizes, d->sizes + s.c.ns_dust);
    for (int i = 0; i < 3; ++i) {
      s.drag.gas.ix[i] = d->gas.ix[i], s.drag.gas.ox[i] = d->gas.ox[i];
      s.drag.gas.irate[i] = d->gas.irate[i], s.drag.gas.orate[i] = d->gas.orate[i];
      s.drag.dust.ix[i] = d->dust.ix[i], s.drag.dust.ox[i] = d->dust.ox[i];
      s.drag.dust.irate[i] = d->dust.irate[i], s.drag.dust.orate[i] = d->dust.orate[i];
    }
    s.gx1min = d->xmin[0], s.gx2min = d->xmin[1], s.gx3min = d->xmin[2];
    s.gx1max = d->xmax[0], s.gx2max = d->xmax[1], s.gx3max = d->xmax[2];
    set_damp_visc(s, d);
    drag_source(s, dt);
    set_aux(s);
    cons_to_prim(s);
    for (const artemis_ml_fix_cell_t *fc : kv.second) {
      const long c = IDX(s, fc->k, fc->j, fc->i);
      for (int v = 0; v < s.nvg; ++v) {
        if (v >= 4 * s.c.ns_gas && v < 5 * s.c.ns_gas) continue; // P is not an output
        p->gas.prim[b * s.nvg + v][c] = s.gprim[v * s.N + c];
      }
      for (int v = 0; v < s.nvd; ++v) p->dust.prim[b * s.nvd + v][c] = s.dprim[v * s.N + c];
    }
  }
  return 0;
}
int artemis_hip_stage_general_variant(const artemis_pack_t *, const artemis_stage_general_args_t *) { return 0; }
// One block through the stage (shared by artemis_hip_stage_general and the refined-mesh fix-up, which redoes the
// listed zones of the block with the corrected fluxes of their flagged faces)
static void stage_block(const artemis_pack_t *p, const artemis_stage_general_args_t *a, int b,
                        const std::vector<const artemis_ml_fix_cell_t *> *fix) {
    Bound B(p, b);
    Sim &s = *B.s;
    B.in(s.gprim, a->gas_u1, s.nvg), B.in(s.dprim, a->dust_u1, s.nvd);
    prim_to_cons(s);
    s.gu1 = s.gu0, s.du1 = s.du0;
    B.in(s.gprim, a->gas_in, s.nvg), B.in(s.dprim, a->dust_in, s.nvd);
    prim_to_cons_as_supplied(s);
    calculate_fluxes(s, FL_GAS, a->pcm != 0), calculate_fluxes(s, FL_DUST, a->pcm != 0);
    if (fix) { // refined meshes: the flagged faces take the corrected flux fields from the pack's arrays
      for (const artemis_ml_fix_cell_t *fc : *fix)
        for (int d = 0; d < 3; ++d)
          for (int side = 0; side < 2; ++side) {
            if (!((fc->faces >> (2 * d + side)) & 1u)) continue;
            const long st = (d == 0) ? 1 : ((d == 1) ? s.ni : static_cast<long>(s.ni) * s.nj);
            const long cf = IDX(s, fc->k, fc->j, fc->i) + side * st;
            for (int v = 0; v < s.nvg; ++v) s.gflux[d][v * s.N + cf] = p->gas.flux[d][b * s.nvg + v][cf];
            for (int v = 0; v < s.c.ns_gas; ++v) s.gpflux[d][v * s.N + cf] = p->gas.pflux[d][b * s.c.ns_gas + v][cf];
            for (int v = 0; v < s.nvd; ++v) s.dflux[d][v * s.N + cf] = p->dust.flux[d][b * s.nvd + v][cf];
          }
    }
    apply_update(s, a->gam0, a->gam1, a->beta_dt);
    flux_source(s, FL_GAS, a->bdt), flux_source(s, FL_DUST, a->bdt);
    if (a->diffusion) {
      dflux_io(B, p, false), set_diffusion(s, a->diffusion);
      diffusion_update(s, a->bdt);
    }
    if (a->gravity) {
      const artemis_gravity_t *g = a->gravity;
      s.grav.type = g->type, s.grav.gm = g->gm, s.grav.soft = g->soft, s.grav.sink = g->sink;
      s.grav.sink_rate = g->sink_rate, s.grav.tstart = g->tstart, s.grav.tstop = g->tstop;
      for (int d = 0; d < 3; ++d) s.grav.g[d] = g->g[d], s.grav.pos[d] = g->pos[d], s.grav.pos2[d] = g->pos2[d];
      s.grav.q = g->q, s.grav.soft2 = g->soft2, s.grav.sink2 = g->sink2, s.grav.sink_rate2 = g->sink_rate2;
      s.grav.given_pos = true;
      external_gravity(s, a->time, a->bdt);
    }
    if (a->nbody_n) { // Gravity::NBodyGravity in the gravity task's slot ("device" memory is host memory here)
      set_nbody(s, a->nbody_dev, a->nbody_n, a->nbody_omf);
      s.grav.tstart = -1e300, s.grav.tstop = 1e300;
      nbody_gravity(s, a->time, a->bdt);
      s.rframe.on = (p->omega_frame != 0.0), s.rframe.omega = p->omega_frame;
    }
    if (a->rf_omega != 0.0) {
      s.rframe.on = true, s.rframe.omega = a->rf_omega, s.rframe.qshear = a->rf_qshear;
      rotating_frame_force(s, a->bdt);
    }
    if (a->defer_finish == 1 || (a->defer_finish && fix)) { // the conserved state of the zones, for artemis_hip_stage_finish (_cells)
      auto putc = [&](const RVec &src, double *const *tab, int nvar) {
        for (int v = 0; v < nvar; ++v) {
          if (fix) {
            for (const artemis_ml_fix_cell_t *fc : *fix) tab[b * nvar + v][IDX(s, fc->k, fc->j, fc->i)] = src[v * s.N + IDX(s, fc->k, fc->j, fc->i)];
            continue;
          }
          for (int k = s.ks; k <= s.ke; ++k)
            for (int j = s.js; j <= s.je; ++j)
              std::memcpy(tab[b * nvar + v] + IDX(s, k, j, s.is), src.data() + v * s.N + IDX(s, k, j, s.is),
                          (s.ie - s.is + 1) * sizeof(Real));
        }
      };
      if (s.c.ns_gas) putc(s.gu0, p->gas.cons0, s.nvg);
      if (s.c.ns_dust) putc(s.du0, p->dust.cons0, s.nvd);
      return;
    }
    if (a->drag) {
      const artemis_drag_t *d = a->drag;
      s.drag.type = d->type, s.drag.model = d->model, s.drag.scale = d->scale;
      s.drag.grain_density = d->grain_density;
      s.drag.tau.assign(d->tau, d->tau + s.c.ns_dust), s.drag.sizes.assign(d->sizes, d->sizes + s.c.ns_dust);
      for (int i = 0; i < 3; ++i) {
        s.drag.gas.ix[i] = d->gas.ix[i], s.drag.gas.ox[i] = d->gas.ox[i];
        s.drag.gas.irate[i] = d->gas.irate[i], s.drag.gas.orate[i] = d->gas.orate[i];
        s.drag.dust.ix[i] = d->dust.ix[i], s.drag.dust.ox[i] = d->dust.ox[i];
        s.drag.dust.irate[i] = d->dust.irate[i], s.drag.dust.orate[i] = d->dust.orate[i];
      }
      s.gx1min = d->xmin[0], s.gx2min = d->xmin[1], s.gx3min = d->xmin[2];
      s.gx1max = d->xmax[0], s.gx2max = d->xmax[1], s.gx3max = d->xmax[2];
      set_damp_visc(s, d);
      drag_source(s, a->bdt);
    }
    if (a->cooling) {
      const int gtype = s.grav.type; // set_cooling overwrites the gravity type: keep it for nothing below
      set_cooling(s, a->cooling);
      cooling_source(s, a->time, a->bdt);
      s.grav.type = gtype;
    }
    set_aux(s);
    cons_to_prim(s);
    auto put = [&](const RVec &src, double *const *tab, int nvar, bool gas) {
      for (int v = 0; v < nvar; ++v) {
        if (gas && v >= 4 * s.c.ns_gas && v < 5 * s.c.ns_gas) continue; // P is not an output
        if (fix) { // only the listed zones
          for (const artemis_ml_fix_cell_t *fc : *fix) tab[b * nvar + v][IDX(s, fc->k, fc->j, fc->i)] = src[v * s.N + IDX(s, fc->k, fc->j, fc->i)];
          continue;
        }
        for (int k = s.ks; k <= s.ke; ++k)
          for (int j = s.js; j <= s.je; ++j)
            std::memcpy(tab[b * nvar + v] + IDX(s, k, j, s.is), src.data() + v * s.N + IDX(s, k, j, s.is),
                        (s.ie - s.is + 1) * sizeof(Real));
      }
    };
    if (s.c.ns_gas) put(s.gprim, a->gas_out, s.nvg, true);
    if (s.c.ns_dust) put(s.dprim, a->dust_out, s.nvd, false);
    if (a->dt_dev) {
      if (s.c.ns_gas) *a->dt_dev = std::min(*a->dt_dev, a->cfl_gas * estimate_dt(s, FL_GAS));
      if (s.c.ns_dust) *a->dt_dev = std::min(*a->dt_dev, a->cfl_dust * estimate_dt(s, FL_DUST));
    }
}
int artemis_hip_stage_general(const artemis_pack_t *p, const artemis_stage_general_args_t *a_in, void *) {
  artemis_stage_general_args_t args = *a_in;
  if (args.beta_dt_dev) args.beta_dt = args.bdt = *args.beta_dt_dev; // "device" memory is host memory here
  const artemis_stage_general_args_t *a = &args;
  if ((p->gas.nspecies && a->gas_in == a->gas_out) || (p->dust.nspecies && a->dust_in == a->dust_out))
    return bad("*_out must not alias *_in");
  for (int b = 0; b < p->nblocks; ++b) stage_block(p, a, b, nullptr);
  return 0;
}
// include/artemis_hip.h "flux correction as a thin fix-up": the fine side's faces ...
int artemis_hip_ml_face_fluxes(const artemis_pack_t *p, const artemis_stage_general_args_t *a, const artemis_ml_face_box_t *boxes,
                               int nboxes, void *) {
  std::map<int, std::vector<const artemis_ml_face_box_t *>> per_block;
  for (int q = 0; q < nboxes; ++q) per_block[boxes[q].block].push_back(&boxes[q]);
  for (auto &kv : per_block) {
    const int b = kv.first;
    Bound B(p, b);
    Sim &s = *B.s;
    B.in(s.gprim, a->gas_in, s.nvg), B.in(s.dprim, a->dust_in, s.nvd);
    prim_to_cons_as_supplied(s); // (refreshes the pressure of every zone, ghosts included)
    calculate_fluxes(s, FL_GAS, a->pcm != 0), calculate_fluxes(s, FL_DUST, a->pcm != 0);
    for (const artemis_ml_face_box_t *bx : kv.second) {
      const int d = bx->dir;
      for (int k = bx->lo[2]; k < bx->lo[2] + bx->n[2]; ++k)
        for (int j = bx->lo[1]; j < bx->lo[1] + bx->n[1]; ++j)
          for (int i = bx->lo[0]; i < bx->lo[0] + bx->n[0]; ++i) {
            const long c = IDX(s, k, j, i);
            for (int v = 0; v < s.nvg; ++v) p->gas.flux[d][b * s.nvg + v][c] = s.gflux[d][v * s.N + c];
            for (int v = 0; v < s.c.ns_gas; ++v) p->gas.pflux[d][b * s.c.ns_gas + v][c] = s.gpflux[d][v * s.N + c];
            for (int v = 0; v < s.c.ns_gas; ++v) p->gas.vface[d][b * s.c.ns_gas + v][c] = s.gvface[d][v * s.N + c];
            for (int v = 0; v < s.nvd; ++v) p->dust.flux[d][b * s.nvd + v][c] = s.dflux[d][v * s.N + c];
          }
    }
  }
  return 0;
}
// ... and the coarse zones that touch them, redone
int artemis_hip_ml_stage_fixup(const artemis_pack_t *p, const artemis_stage_general_args_t *a_in, const artemis_ml_fix_cell_t *cells,
                               int ncells, void *) {
  artemis_stage_general_args_t args = *a_in;
  if (args.beta_dt_dev) args.beta_dt = args.bdt = *args.beta_dt_dev;
  args.dt_dev = nullptr;
  if (args.drag && !args.defer_finish) return bad("refined-mesh fix-up: drag needs defer_finish");
  std::map<int, std::vector<const artemis_ml_fix_cell_t *>> per_block;
  for (int q = 0; q < ncells; ++q) per_block[cells[q].block].push_back(&cells[q]);
  for (auto &kv : per_block) stage_block(p, &args, kv.first, &kv.second);
  return 0;
}

static void slab(const artemis_pack_t *p, int face, int unpack, int lo[3], int n[3], int extended = 0) {
  const int ndim = (p->nx3 > 1) ? 3 : ((p->nx2 > 1) ? 2 : 1);
  const int nx[3] = {p->nx1, p->nx2, p->nx3};
  const int d = face / 2, side = face % 2, ng = p->nghost;
  for (int q = 0; q < 3; ++q) {
    lo[q] = (q < ndim) ? ng : 0, n[q] = nx[q];
    if (extended && q < d) lo[q] = 0, n[q] = nx[q] + 2 * ng; // dims below d are active: entire extent
  }
  n[d] = ng;
  const int st = lo[d], en = st + nx[d] - 1;
  if (!unpack) lo[d] = side ? en - ng + 1 : st;
  else lo[d] = side ? en + 1 : st - ng;
}
long artemis_hip_halo_count_ext(const artemis_pack_t *p, int face, int extended) {
  int lo[3], n[3];
  slab(p, face, 0, lo, n, extended);
  return static_cast<long>(n[0]) * n[1] * n[2] * (5 * p->gas.nspecies + 4 * p->dust.nspecies);
}
long artemis_hip_halo_count(const artemis_pack_t *p, int face) { return artemis_hip_halo_count_ext(p, face, 0); }
static int halo(const artemis_pack_t *p, int b, int face, double *buf, int unpack, int extended = 0) {
  int lo[3], n[3];
  slab(p, face, unpack, lo, n, extended);
  const int ndim = (p->nx3 > 1) ? 3 : ((p->nx2 > 1) ? 2 : 1);
  const int ni = p->nx1 + 2 * p->nghost, nj = p->nx2 + (ndim > 1 ? 2 * p->nghost : 0);
  const long ncell = static_cast<long>(n[0]) * n[1] * n[2];
  const int nsg = p->gas.nspecies, nsd = p->dust.nspecies;
  int v = 0;
  auto doit = [&](double *q) {
    long t = 0;
    for (int k = 0; k < n[2]; ++k)
      for (int j = 0; j < n[1]; ++j)
        for (int i = 0; i < n[0]; ++i, ++t) {
          const size_t c = (static_cast<size_t>(lo[2] + k) * nj + lo[1] + j) * ni + lo[0] + i;
          if (unpack) q[c] = buf[v * ncell + t];
          else buf[v * ncell + t] = q[c];
        }
    ++v;
  };
  for (int s = 0; s < 6 * nsg; ++s)
    if (!(s >= 4 * nsg && s < 5 * nsg)) doit(p->gas.prim[b * 6 * nsg + s]);
  for (int s = 0; s < 4 * nsd; ++s) doit(p->dust.prim[b * 4 * nsd + s]);
  return 0;
}
int artemis_hip_halo_pack(const artemis_pack_t *p, int b, int face, double *buf, void *) {
  return halo(p, b, face, buf, 0);
}
int artemis_hip_halo_unpack(const artemis_pack_t *p, int b, int face, const double *buf, void *) {
  return halo(p, b, face, const_cast<double *>(buf), 1);
}
int artemis_hip_halo_pack_ext(const artemis_pack_t *p, int b, int face, int extended, double *buf, void *) {
  return halo(p, b, face, buf, 0, extended);
}
int artemis_hip_halo_unpack_ext(const artemis_pack_t *p, int b, int face, int extended, const double *buf, void *) {
  return halo(p, b, face, const_cast<double *>(buf), 1, extended);
}
int artemis_hip_diffusion_radial_fill(const artemis_pack_t *p, const double *geom_host, const double *,
                                      const artemis_diffcoeff_t *c, int block, double *out) {
  artemis_pack_t q = *p;
  q.geom = geom_host;
  Bound B(&q, block);
  Sim &s = *B.s;
  for (int k = 0; k < s.nk; ++k)
    for (int j = 0; j < s.nj; ++j)
      for (int i = 0; i < s.ni; ++i) {
        const Coords co(s, k, j, i);
        const Real xv[3] = {co.x1v(), co.x2v(), co.x3v()};
        out[IDX(s, k, j, i)] = (c->type == ARTEMIS_VISCOSITY_PLAW)
                                   ? std::pow(to_cyl_with_vec(co, xv).R / c->r0, c->r_exp)
                                   : c->omega0 * std::pow(to_sph_radius(co, xv) / c->r0, -1.5);
      }
  return 0;
}
// the distance table is an optimisation of the device kernels; the double evaluates Coords::Distance directly
size_t artemis_hip_viscous_distance_count(const artemis_pack_t *p) {
  const int g = p->nghost;
  return 6 * static_cast<size_t>(p->nblocks) * (p->nx1 + 2 * g) * (p->nx2 > 1 ? p->nx2 + 2 * g : 1) * (p->nx3 > 1 ? p->nx3 + 2 * g : 1);
}
int artemis_hip_viscous_distance_fill(const artemis_pack_t *, double *, void *) { return 0; }
int artemis_hip_zero_diffusion_flux(const artemis_pack_t *p, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    zero_diffusion_flux(*B.s);
    dflux_io(B, p, true);
  }
  return 0;
}
int artemis_hip_viscous_flux(const artemis_pack_t *p, const artemis_diffusion_t *d, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state(), dflux_io(B, p, false), set_diffusion(*B.s, d);
    viscous_flux(*B.s);
    dflux_io(B, p, true);
  }
  return 0;
}
int artemis_hip_zero_viscous_flux(const artemis_pack_t *p, const artemis_diffusion_t *d, void *s) {
  if (int rc = artemis_hip_zero_diffusion_flux(p, s)) return rc;
  return (d->visc.type == ARTEMIS_DIFF_OFF) ? 0 : artemis_hip_viscous_flux(p, d, s);
}
// the viscous-source march is a device-side reorganisation of the three tasks above: this double keeps the tasks
int artemis_hip_viscous_source_covers(const artemis_pack_t *) { return 0; }
int artemis_hip_viscous_source(const artemis_pack_t *, const artemis_diffusion_t *, double, const double *, double *const *, void *) {
  return ARTEMIS_HIP_EUNSUPPORTED;
}
int artemis_hip_ml_viscous_faces(const artemis_pack_t *, const artemis_diffusion_t *, const artemis_ml_face_box_t *, int,
                                 const artemis_ml_fix_cell_t *, int, void *) {
  return ARTEMIS_HIP_EUNSUPPORTED; // (never reached: artemis_hip_viscous_source_covers is 0 on this double)
}
int artemis_hip_thermal_flux(const artemis_pack_t *p, const artemis_diffusion_t *d, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state(), dflux_io(B, p, false), set_diffusion(*B.s, d);
    thermal_flux(*B.s);
    dflux_io(B, p, true);
  }
  return 0;
}
int artemis_hip_diffusion_update(const artemis_pack_t *p, const artemis_diffusion_t *d, double dt, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state(), dflux_io(B, p, false), set_diffusion(*B.s, d);
    diffusion_update(*B.s, dt);
    B.out(B.s->gu0, p->gas.cons0, B.s->nvg);
  }
  return 0;
}
int artemis_hip_diffusion_dt(const artemis_pack_t *p, const artemis_diffusion_t *d, double cfl,
                             double *dt_dev, void *) {
  for (int b = 0; b < p->nblocks; ++b) {
    Bound B(p, b);
    B.load_state(), set_diffusion(*B.s, d);
    *dt_dev = std::min(*dt_dev, cfl * std::min(diffusion_dt(*B.s, B.s->visc), diffusion_dt(*B.s, B.s->cond)));
  }
  return 0;
}
int artemis_hip_timestep_all(const artemis_pack_t *p, double cfl_gas, double cfl_dust, const artemis_diffusion_t *d, double *dt_dev,
                             void *s) {
  if (p->gas.nspecies)
    if (int rc = artemis_hip_estimate_dt_async(p, ARTEMIS_GAS, cfl_gas, dt_dev, s)) return rc;
  if (p->dust.nspecies)
    if (int rc = artemis_hip_estimate_dt_async(p, ARTEMIS_DUST, cfl_dust, dt_dev, s)) return rc;
  if (d && p->gas.nspecies) return artemis_hip_diffusion_dt(p, d, cfl_gas, dt_dev, s);
  return 0;
}
int artemis_hip_wait_counter(unsigned *, unsigned, unsigned *, void *) { return 0; }
int artemis_hip_advance_dt(double *st, double tlim, int nstages, const double *beta, void *) {
  double time = st[0], dt = st[1];
  const double est = st[2];
  time += dt;
  double ndt = dt;
  if (ndt < 0.1 * DBL_MAX) ndt *= 2.0;
  ndt = std::min(ndt, est);
  if (tlim > 0.0 && time < tlim && (tlim - time) < ndt) ndt = tlim - time;
  st[0] = time, st[1] = ndt, st[2] = DBL_MAX;
  for (int q = 0; q < nstages; ++q) st[3 + q] = beta[q] * ndt;
  return 0;
}
int artemis_hip_selftest_divsqrt(long, const double *, const double *, double *, double *, double *,
                                 double *, void *) {
  return ARTEMIS_HIP_EUNSUPPORTED;
}

// ---- runtime shim on host memory ----------------------------------------------------------
// the refinement operators are exercised against the oracle on the GPU only; the host stand-in has no use for them
// ---- multilevel block-graph data path on host memory (include/artemis_hip.h), with the oracle's Coords --------
} // extern "C"
namespace {
struct MlHost {
  const artemis_pack_t *p;
  const artemis_ml_pack_t *ml;
  int ndim, ni, nj, nk, cni, cnj, cnk, s[3], cn[3];
  std::vector<std::unique_ptr<Bound>> fine, coarse;
  artemis_pack_t pc;
  MlHost(const artemis_pack_t *p_, const artemis_ml_pack_t *ml_) : p(p_), ml(ml_) {
    ndim = (p->nx3 > 1) ? 3 : ((p->nx2 > 1) ? 2 : 1);
    const int g = p->nghost, nx[3] = {p->nx1, p->nx2, p->nx3};
    for (int d = 0; d < 3; ++d) s[d] = (d < ndim) ? g : 0, cn[d] = (d < ndim) ? nx[d] / 2 : 1;
    ni = nx[0] + 2 * s[0], nj = nx[1] + 2 * s[1], nk = nx[2] + 2 * s[2];
    cni = cn[0] + 2 * s[0], cnj = cn[1] + 2 * s[1], cnk = cn[2] + 2 * s[2];
    fine.resize(p->nblocks), coarse.resize(p->nblocks);
    pc = *p;
    pc.nx1 = cn[0], pc.nx2 = cn[1], pc.nx3 = cn[2];
    if (ml) pc.geom = ml->cgeom, pc.metric = ml->cmetric;
  }
  Sim &F(int b) {
    if (!fine[b]) fine[b].reset(new Bound(p, b));
    return *fine[b]->s;
  }
  Sim &C(int b) {
    if (!coarse[b]) coarse[b].reset(new Bound(&pc, b));
    return *coarse[b]->s;
  }
  int nfill() const { return 5 * p->gas.nspecies + 4 * p->dust.nspecies; }
  double *var(bool crs, int b, int v) const {
    const int nsg = p->gas.nspecies, nsd = p->dust.nspecies;
    if (v < 5 * nsg) {
      const int slot = (v < 4 * nsg) ? v : v + nsg;
      return (crs ? ml->gas_coarse : p->gas.prim)[b * 6 * nsg + slot];
    }
    return (crs ? ml->dust_coarse : p->dust.prim)[b * 4 * nsd + (v - 5 * nsg)];
  }
  size_t fidx(int k, int j, int i) const { return (static_cast<size_t>(k) * nj + j) * ni + i; }
  size_t cidx(int k, int j, int i) const { return (static_cast<size_t>(k) * cnj + j) * cni + i; }
  // RestrictAverage (restriction.hpp:73-112) of the fine zones at (fk, fj, fi) of block b, weights w (volumes / areas)
  template <class W>
  double restrict8(const double *q, int fk, int fj, int fi, bool I1, bool I2, bool I3, W weight) const {
    Real w[2][2][2], t[2][2][2];
    for (int ok = 0; ok < 2; ++ok)
      for (int oj = 0; oj < 2; ++oj)
        for (int oi = 0; oi < 2; ++oi) w[ok][oj][oi] = t[ok][oj][oi] = 0;
    for (int ok = 0; ok < 1 + I3; ++ok)
      for (int oj = 0; oj < 1 + I2; ++oj)
        for (int oi = 0; oi < 1 + I1; ++oi) {
          w[ok][oj][oi] = weight(fk + ok, fj + oj, fi + oi);
          t[ok][oj][oi] = w[ok][oj][oi] * q[fidx(fk + ok, fj + oj, fi + oi)];
        }
    const Real tw = ((w[0][0][0] + w[0][1][0]) + (w[0][0][1] + w[0][1][1])) + ((w[1][0][0] + w[1][1][0]) + (w[1][0][1] + w[1][1][1]));
    return (((t[0][0][0] + t[0][1][0]) + (t[0][0][1] + t[0][1][1])) + ((t[1][0][0] + t[1][1][0]) + (t[1][0][1] + t[1][1][1]))) / tw;
  }
};
} // namespace
extern "C" {
int artemis_hip_ml_exchange(const artemis_pack_t *p, const artemis_ml_pack_t *ml, const artemis_ml_op_t *ops, int nops,
                            double *sbuf, const double *rbuf, void *) {
  MlHost H(p, ml);
  const int nf = H.nfill();
  for (int q = 0; q < nops; ++q) {
    const artemis_ml_op_t &op = ops[q];
    const bool crs = (op.kind == ARTEMIS_ML_FROM_COARSER);
    const long ncell = static_cast<long>(op.n[0]) * op.n[1] * op.n[2];
    long t = 0;
    for (int q2 = 0; q2 < op.n[2]; ++q2)
      for (int q1 = 0; q1 < op.n[1]; ++q1)
        for (int q0 = 0; q0 < op.n[0]; ++q0, ++t) {
          const int i = op.lo[0] + q0, j = op.lo[1] + q1, k = op.lo[2] + q2;
          const size_t dc = crs ? H.cidx(k, j, i) : H.fidx(k, j, i);
          for (int v = 0; v < nf; ++v) {
            double val;
            if (op.src_block < 0) {
              val = rbuf[op.buf + v * ncell + t];
            } else if (op.kind == ARTEMIS_ML_FROM_FINER) {
              const int fi = 2 * i + op.off[0], fj = (H.ndim > 1) ? 2 * j + op.off[1] : j, fk = (H.ndim > 2) ? 2 * k + op.off[2] : k;
              Sim &fs = H.F(op.src_block);
              val = H.restrict8(H.var(false, op.src_block, v), fk, fj, fi, true, H.ndim > 1, H.ndim > 2,
                                [&](int kk, int jj, int ii) { return Coords(fs, kk, jj, ii).Volume(); });
            } else {
              val = H.var(false, op.src_block, v)[H.fidx(k + op.off[2], j + op.off[1], i + op.off[0])];
            }
            if (op.dst_block < 0) sbuf[op.buf + v * ncell + t] = val;
            else H.var(crs, op.dst_block, v)[dc] = val;
          }
        }
  }
  return 0;
}
int artemis_hip_ml_restrict_halos(const artemis_pack_t *p, const artemis_ml_pack_t *ml, const int *blocks, int nblocks, void *) {
  MlHost H(p, ml);
  const int h = p->nghost / 2, nf = H.nfill();
  for (int q = 0; q < nblocks; ++q) {
    const int b = blocks[q];
    Sim &fs = H.F(b);
    const int lo[3] = {H.s[0] - h, H.ndim > 1 ? H.s[1] - h : 0, H.ndim > 2 ? H.s[2] - h : 0};
    const int hi[3] = {H.s[0] + H.cn[0] + h, H.ndim > 1 ? H.s[1] + H.cn[1] + h : 1, H.ndim > 2 ? H.s[2] + H.cn[2] + h : 1};
    for (int ck = lo[2]; ck < hi[2]; ++ck)
      for (int cj = lo[1]; cj < hi[1]; ++cj)
        for (int ci = lo[0]; ci < hi[0]; ++ci) {
          const int fi = (ci - H.s[0]) * 2 + H.s[0], fj = (H.ndim > 1) ? (cj - H.s[1]) * 2 + H.s[1] : 0;
          const int fk = (H.ndim > 2) ? (ck - H.s[2]) * 2 + H.s[2] : 0;
          for (int v = 0; v < nf; ++v)
            H.var(true, b, v)[H.cidx(ck, cj, ci)] =
                H.restrict8(H.var(false, b, v), fk, fj, fi, true, H.ndim > 1, H.ndim > 2,
                            [&](int kk, int jj, int ii) { return Coords(fs, kk, jj, ii).Volume(); });
        }
  }
  return 0;
}
int artemis_hip_ml_prolongate(const artemis_pack_t *p, const artemis_ml_pack_t *ml, const artemis_ml_box_t *boxes, int nboxes, void *) {
  MlHost H(p, ml);
  const int nf = H.nfill(), DIM = H.ndim;
  const bool X1 = true, X2 = DIM > 1, X3 = DIM > 2;
  auto sign = [](Real a) { return (a < 0.) ? -1. : 1.; };
  auto centre = [](const Coords &co, int d) { return d == 1 ? co.x1v() : (d == 2 ? co.x2v() : co.x3v()); };
  for (int q = 0; q < nboxes; ++q) {
    const artemis_ml_box_t &bx = boxes[q];
    Sim &f = H.F(bx.block), &c = H.C(bx.block);
    for (int k = bx.lo[2]; k < bx.lo[2] + bx.n[2]; ++k)
      for (int j = bx.lo[1]; j < bx.lo[1] + bx.n[1]; ++j)
        for (int i = bx.lo[0]; i < bx.lo[0] + bx.n[0]; ++i) {
          const int fi = (i - H.s[0]) * 2 + H.s[0], fj = X2 ? (j - H.s[1]) * 2 + H.s[1] : 0, fk = X3 ? (k - H.s[2]) * 2 + H.s[2] : 0;
          for (int v = 0; v < nf; ++v) { // prolongation.hpp:83-184
            const Real *qc = H.var(true, bx.block, v);
            const Real fc = qc[H.cidx(k, j, i)];
            Real dxfm[3] = {0, 0, 0}, dxfp[3] = {0, 0, 0}, g[3] = {0, 0, 0};
            for (int d = 1; d <= DIM; ++d) {
              const int dk = (d == 3), dj = (d == 2), di = (d == 1);
              const Real xm = centre(Coords(c, k - dk, j - dj, i - di), d), xc = centre(Coords(c, k, j, i), d);
              const Real xp = centre(Coords(c, k + dk, j + dj, i + di), d);
              const Real fxm = centre(Coords(f, fk, fj, fi), d), fxp = centre(Coords(f, fk + dk, fj + dj, fi + di), d);
              const Real dxm = xc - xm, dxp = xp - xc;
              dxfm[d - 1] = xc - fxm, dxfp[d - 1] = fxp - xc;
              const Real gxm = (fc - qc[H.cidx(k - dk, j - dj, i - di)]) / dxm;
              const Real gxp = (qc[H.cidx(k + dk, j + dj, i + di)] - fc) / dxp;
              g[d - 1] = 0.5 * (sign(gxm) + sign(gxp)) * std::min(std::abs(gxm), std::abs(gxp));
            }
            const Real gx1m = g[0], gx1p = g[0], gx2m = g[1], gx2p = g[1], gx3m = g[2], gx3p = g[2];
            const Real dx1fm = dxfm[0], dx1fp = dxfp[0], dx2fm = dxfm[1], dx2fp = dxfp[1], dx3fm = dxfm[2], dx3fp = dxfp[2];
            Real *o = H.var(false, bx.block, v);
            o[H.fidx(fk, fj, fi)] = fc - (gx1m * dx1fm + gx2m * dx2fm + gx3m * dx3fm);
            if (X1) o[H.fidx(fk, fj, fi + 1)] = fc + (gx1p * dx1fp - gx2m * dx2fm - gx3m * dx3fm);
            if (X2) o[H.fidx(fk, fj + 1, fi)] = fc - (gx1m * dx1fm - gx2p * dx2fp + gx3m * dx3fm);
            if (X2 && X1) o[H.fidx(fk, fj + 1, fi + 1)] = fc + (gx1p * dx1fp + gx2p * dx2fp - gx3m * dx3fm);
            if (X3) o[H.fidx(fk + 1, fj, fi)] = fc - (gx1m * dx1fm + gx2m * dx2fm - gx3p * dx3fp);
            if (X3 && X1) o[H.fidx(fk + 1, fj, fi + 1)] = fc + (gx1p * dx1fp - gx2m * dx2fm + gx3p * dx3fp);
            if (X3 && X2) o[H.fidx(fk + 1, fj + 1, fi)] = fc - (gx1m * dx1fm - gx2p * dx2fp - gx3p * dx3fp);
            if (X3 && X2 && X1) o[H.fidx(fk + 1, fj + 1, fi + 1)] = fc + (gx1p * dx1fp + gx2p * dx2fp + gx3p * dx3fp);
          }
        }
  }
  return 0;
}
int artemis_hip_ml_floor_ghosts(const artemis_pack_t *p, const int *blocks, int nblocks, void *) {
  // PrimToCons's primitive floors (fill_derived.cpp:227, :245, :262) on the ghost zones of the listed blocks
  const int ndim = (p->nx3 > 1) ? 3 : ((p->nx2 > 1) ? 2 : 1), g = p->nghost;
  const int ni = p->nx1 + 2 * g, nj = (ndim > 1) ? p->nx2 + 2 * g : 1, nk = (ndim > 2) ? p->nx3 + 2 * g : 1;
  const int lo[3] = {g, ndim > 1 ? g : 0, ndim > 2 ? g : 0}, hi[3] = {g + p->nx1 - 1, lo[1] + p->nx2 - 1, lo[2] + p->nx3 - 1};
  const int nsg = p->gas.nspecies, nsd = p->dust.nspecies;
  for (int q = 0; q < nblocks; ++q) {
    const int b = blocks[q];
    for (int k = 0; k < nk; ++k)
      for (int j = 0; j < nj; ++j)
        for (int i = 0; i < ni; ++i) {
          if (i >= lo[0] && i <= hi[0] && j >= lo[1] && j <= hi[1] && k >= lo[2] && k <= hi[2]) continue;
          const size_t c = (static_cast<size_t>(k) * nj + j) * ni + i;
          for (int n = 0; n < nsg; ++n) {
            Real &w_d = p->gas.prim[b * 6 * nsg + n][c], &w_s = p->gas.prim[b * 6 * nsg + 5 * nsg + n][c];
            w_d = (w_d > p->gas.dfloor) ? w_d : p->gas.dfloor;
            w_s = (w_s > p->gas.siefloor) ? w_s : p->gas.siefloor;
          }
          for (int n = 0; n < nsd; ++n) {
            Real &w_d = p->dust.prim[b * 4 * nsd + n][c];
            w_d = (w_d > p->dust.dfloor) ? w_d : p->dust.dfloor;
          }
        }
  }
  return 0;
}
int artemis_hip_ml_flux_correction(const artemis_pack_t *p, const artemis_ml_op_t *ops, int nops, double *sbuf, const double *rbuf,
                                   void *) {
  MlHost H(p, nullptr);
  const int nsg = p->gas.nspecies, nsd = p->dust.nspecies;
  for (int q = 0; q < nops; ++q) {
    const artemis_ml_op_t &op = ops[q];
    const int d = op.dir;
    const bool diff = nsg > 0 && p->gas.diff_flux[0] != nullptr;
    std::vector<std::pair<double *const *, int>> sets; // (table, vars per block)
    sets.push_back({p->gas.flux[d], 6 * nsg}), sets.push_back({p->gas.pflux[d], nsg});
    if (diff) sets.push_back({p->gas.diff_flux[d], 4 * nsg});
    sets.push_back({p->dust.flux[d], 4 * nsd});
    const long ncell = static_cast<long>(op.n[0]) * op.n[1] * op.n[2];
    const bool I1 = d != 0, I2 = (H.ndim > 1) && d != 1, I3 = (H.ndim > 2) && d != 2;
    long t = 0;
    for (int q2 = 0; q2 < op.n[2]; ++q2)
      for (int q1 = 0; q1 < op.n[1]; ++q1)
        for (int q0 = 0; q0 < op.n[0]; ++q0, ++t) {
          const int i = op.lo[0] + q0, j = op.lo[1] + q1, k = op.lo[2] + q2;
          const int fi = (d == 0) ? op.off[0] : 2 * i + op.off[0];
          const int fj = (d == 1) ? op.off[1] : ((H.ndim > 1) ? 2 * j + op.off[1] : j);
          const int fk = (d == 2) ? op.off[2] : ((H.ndim > 2) ? 2 * k + op.off[2] : k);
          int v = 0;
          for (auto &st : sets)
            for (int n = 0; n < st.second; ++n, ++v) {
              double val;
              if (op.src_block < 0) {
                val = rbuf[op.buf + v * ncell + t];
              } else {
                Sim &fs = H.F(op.src_block);
                val = H.restrict8(st.first[op.src_block * st.second + n], fk, fj, fi, I1, I2, I3, [&](int kk, int jj, int ii) {
                  Real a[2];
                  Coords co(fs, kk, jj, ii);
                  if (d == 0) co.GetFaceAreaX1(a);
                  else if (d == 1) co.GetFaceAreaX2(a);
                  else co.GetFaceAreaX3(a);
                  return a[0];
                });
              }
              if (op.dst_block < 0) sbuf[op.buf + v * ncell + t] = val;
              else st.first[op.dst_block * st.second + n][H.fidx(k, j, i)] = val;
            }
        }
  }
  return 0;
}

} // extern "C" (helpers with C++ linkage)
namespace {
// An oracle Sim on an index space given by its edge table {x1f0, dx1, ...} and array extents (ghosts included in
// the active dimensions), with room for `nvar` cell-centred arrays in its gas-primitive slots.
Sim *sim_on(int coords, int ndim, int ng, int ni, int nj, int nk, const double *g, int nvar) {
  oracle_cfg c;
  std::memset(&c, 0, sizeof c);
  c.ng = ng;
  c.nx1 = ni - 2 * ng, c.nx2 = (ndim > 1) ? nj - 2 * ng : 1, c.nx3 = (ndim > 2) ? nk - 2 * ng : 1;
  c.ns_gas = (nvar + 5) / 6, c.ns_dust = 0;
  c.gamma = 1.4, c.dfloor_gas = 1e-300, c.siefloor_gas = 1e-300, c.cfl_gas = 1.0, c.cfl_dust = 1.0;
  for (int i = 0; i < 6; ++i) c.bc[i] = BC_NONE;
  c.integrator = INT_RK2;
  c.coords = coords;
  const int gh[3] = {ng, ndim > 1 ? ng : 0, ndim > 2 ? ng : 0};
  const int nx[3] = {c.nx1, c.nx2, c.nx3};
  double lo[3], hi[3];
  for (int d = 0; d < 3; ++d) lo[d] = g[2 * d] + gh[d] * g[2 * d + 1], hi[d] = lo[d] + nx[d] * g[2 * d + 1];
  c.x1min = lo[0], c.x1max = hi[0], c.x2min = lo[1], c.x2max = hi[1], c.x3min = lo[2], c.x3max = hi[2];
  Sim *s = static_cast<Sim *>(oracle_create(&c));
  for (int d = 0; d < 3; ++d) s->f0[d] = g[2 * d], s->dx[d] = g[2 * d + 1];
  return s;
}
int refine_op(const artemis_refine_t *r, bool prolong) {
  // ghost depth: the first active index of the fine / coarse arrays is where interior starts (fib / cib anchors are
  // given relative to it by the callers of this test double: the driver passes is, js, ks)
  const int gf = r->fib;
  Sim *f = sim_on(r->coords, r->ndim, gf, r->fni, r->fnj, r->fnk, r->fgeom, r->nvar);
  Sim *c = sim_on(r->coords, r->ndim, gf, r->cni, r->cnj, r->cnk, r->cgeom, r->nvar);
  for (int v = 0; v < r->nvar; ++v) {
    std::memcpy(f->gprim.data() + static_cast<size_t>(v) * f->N, r->fine[v], f->N * sizeof(Real));
    std::memcpy(c->gprim.data() + static_cast<size_t>(v) * c->N, r->coarse[v], c->N * sizeof(Real));
  }
  const int rr[12] = {r->cis, r->cie, r->cjs, r->cje, r->cks, r->cke, r->cib, r->cjb, r->ckb, r->fib, r->fjb, r->fkb};
  if (prolong) {
    oracle_prolongate_minmod(f, c, rr);
    for (int v = 0; v < r->nvar; ++v) std::memcpy(r->fine[v], f->gprim.data() + static_cast<size_t>(v) * f->N, f->N * sizeof(Real));
  } else {
    oracle_restrict_average(f, c, rr);
    for (int v = 0; v < r->nvar; ++v) std::memcpy(r->coarse[v], c->gprim.data() + static_cast<size_t>(v) * c->N, c->N * sizeof(Real));
  }
  oracle_destroy(f), oracle_destroy(c);
  return 0;
}
int criterion(const artemis_amr_criterion_t *a, int *tag, double *maxval, bool magnitude) {
  Sim *s = sim_on(a->coords, a->ndim, a->is, a->ni, a->nj, a->nk, a->geom, 1);
  std::memcpy(s->gprim.data(), a->field, s->N * sizeof(Real));
  double m = 0.0;
  const int t = magnitude ? oracle_amr_magnitude(s, 0, a->refine_thr, a->deref_thr, &m) : oracle_amr_first_derivative(s, 0, a->refine_thr, &m);
  oracle_destroy(s);
  if (tag) *tag = t;
  if (maxval) *maxval = m;
  return 0;
}
} // namespace
extern "C" {
int artemis_hip_restrict_average(const artemis_refine_t *r, void *) { return refine_op(r, false); }
int artemis_hip_prolongate_minmod(const artemis_refine_t *r, void *) { return refine_op(r, true); }
int artemis_hip_amr_first_derivative(const artemis_amr_criterion_t *a, int *tag, double *mx, void *) { return criterion(a, tag, mx, false); }
int artemis_hip_amr_magnitude(const artemis_amr_criterion_t *a, int *tag, double *mx, void *) { return criterion(a, tag, mx, true); }
int artemis_hip_amr_block_maxima(const artemis_pack_t *p, int field, int magnitude, double *maxima, void *) {
  const int ndim = (p->nx3 > 1) ? 3 : ((p->nx2 > 1) ? 2 : 1), g = p->nghost;
  for (int b = 0; b < p->nblocks; ++b) {
    artemis_amr_criterion_t a;
    std::memset(&a, 0, sizeof a);
    a.coords = p->coords, a.ndim = ndim;
    a.ni = p->nx1 + 2 * g, a.nj = (ndim > 1) ? p->nx2 + 2 * g : 1, a.nk = (ndim > 2) ? p->nx3 + 2 * g : 1;
    a.geom = p->geom + 6 * b;
    a.field = p->gas.prim[b * 6 * p->gas.nspecies + (field == 0 ? 0 : 4 * p->gas.nspecies)];
    std::vector<double> pres;
    if (field == 2) { // the pressure recomputed from rho and sie (fill_derived.cpp:247)
      const size_t n = static_cast<size_t>(a.ni) * a.nj * a.nk;
      const double *rho = p->gas.prim[b * 6 * p->gas.nspecies], *sie = p->gas.prim[b * 6 * p->gas.nspecies + 5 * p->gas.nspecies];
      pres.resize(n);
      for (size_t c = 0; c < n; ++c) pres[c] = std::max(0.0, p->gm1 * rho[c] * sie[c]);
      a.field = pres.data();
    }
    a.is = g, a.ie = g + p->nx1 - 1, a.js = (ndim > 1) ? g : 0, a.je = a.js + p->nx2 - 1, a.ks = (ndim > 2) ? g : 0, a.ke = a.ks + p->nx3 - 1;
    a.refine_thr = 1e300, a.deref_thr = -1e300;
    int tag;
    criterion(&a, &tag, maxima + b, magnitude != 0);
  }
  return 0;
}
int artemis_rt_set_device(int) { return 0; }
void *artemis_rt_malloc(size_t n) { return std::calloc(1, n ? n : 8); }
void artemis_rt_device_bytes(size_t *current, size_t *peak, int) {
  if (current) *current = 0;
  if (peak) *peak = 0;
}
void artemis_rt_free(void *p) { std::free(p); }
void artemis_rt_pool_trim(size_t) {}
size_t artemis_rt_pool_bytes(void) { return 0; }
void artemis_rt_pool_limit(size_t) {}
void *artemis_rt_malloc_host(size_t n) { return std::calloc(1, n ? n : 8); }
void artemis_rt_free_host(void *p) { std::free(p); }
int artemis_rt_memcpy_h2d(void *d, const void *s, size_t n, void *) { std::memcpy(d, s, n); return 0; }
int artemis_rt_memcpy_d2h(void *d, const void *s, size_t n, void *) { std::memcpy(d, s, n); return 0; }
int artemis_rt_memcpy_d2d(void *d, const void *s, size_t n, void *) { std::memmove(d, s, n); return 0; }
int artemis_rt_memset(void *d, int v, size_t n, void *) { std::memset(d, v, n); return 0; }
void *artemis_rt_stream_create(void) { return std::malloc(8); }
void artemis_rt_stream_destroy(void *s) { std::free(s); }
int artemis_rt_stream_sync(void *) { return 0; }
int artemis_rt_device_sync(void) { return 0; }
void *artemis_rt_event_create(void) { return std::malloc(8); }
void artemis_rt_event_destroy(void *e) { std::free(e); }
int artemis_rt_event_record(void *, void *) { return 0; }
int artemis_rt_stream_wait_event(void *, void *) { return 0; }
int artemis_rt_event_sync(void *) { return 0; }
double artemis_rt_event_elapsed_ms(void *, void *) { return 0.0; }
void artemis_rt_tables_changed(void) {}
int artemis_rt_capture_begin(void *) { return 1; } // no graphs on the host stand-in: the driver falls back
void *artemis_rt_capture_end(void *) { return nullptr; }
int artemis_rt_graph_launch(void *, void *) { return 1; }
void artemis_rt_graph_destroy(void *) {}
}
