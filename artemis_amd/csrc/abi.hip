// extern "C" surface of libartemis_hip.so: validates arguments the way the reference's
// PARTHENON_REQUIRE/FAIL guards do, builds the by-value kernel view and enqueues kernels.
// No CPU fallback: without a HIP device every compute entry point returns
// ARTEMIS_HIP_EDEVICE.
#include <cfloat>
#include <cstring>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>

#include "../../include/artemis_hip.h"
#include "../../include/artemis_rt.h"
#include <vector>

#include "kernels.hpp"
#include "options.hpp"
#include "geometry_core.hpp"

namespace {
thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
int check_hip(hipError_t e, const char *what) {
  if (e == hipSuccess) return 0;
  return fail(ARTEMIS_HIP_EDEVICE, "%s: %s", what, hipGetErrorString(e));
}
int device_ready() {
  static int ndev = -1;
  if (ndev < 0) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    ndev = n;
  }
  if (ndev <= 0)
    return fail(ARTEMIS_HIP_EDEVICE,
                "no HIP device visible: libartemis_hip has no CPU fallback (gfx950 required)");
  return 0;
}

// Guards shared by every entry point (gas.cpp:59-96, dust.cpp:50-86, fluid_fluxes.hpp:290).
int validate(const artemis_pack_t *p, bool need_gas_cons = false) {
  if (!p) return fail(ARTEMIS_HIP_EINVAL, "null pack");
  if (p->nblocks < 1 || p->nx1 < 1 || p->nx2 < 1 || p->nx3 < 1)
    return fail(ARTEMIS_HIP_EINVAL, "bad pack extents nblocks=%d nx=(%d,%d,%d)", p->nblocks, p->nx1,
                p->nx2, p->nx3);
  if (p->nx3 > 1 && p->nx2 == 1)
    return fail(ARTEMIS_HIP_EINVAL, "nx3 > 1 requires nx2 > 1");
  if (p->coords < ARTEMIS_CARTESIAN || p->coords > ARTEMIS_AXISYMMETRIC)
    return fail(ARTEMIS_HIP_EINVAL, "Coordinate type not recognized!");
  {
    // geometry::CoordSelect (geometry.hpp:38-56) ties the spherical variant to the dimension
    const int ndim = (p->nx3 > 1) ? 3 : ((p->nx2 > 1) ? 2 : 1);
    const int want = (ndim == 1) ? ARTEMIS_SPHERICAL1D
                                 : ((ndim == 2) ? ARTEMIS_SPHERICAL2D : ARTEMIS_SPHERICAL3D);
    const bool sph = p->coords >= ARTEMIS_SPHERICAL1D && p->coords <= ARTEMIS_SPHERICAL3D;
    if (sph && p->coords != want)
      return fail(ARTEMIS_HIP_EINVAL, "spherical%dD coordinates on a %d-D block", p->coords - 1, ndim);
    if (sph && ndim > 1 && !p->metric)
      return fail(ARTEMIS_HIP_EINVAL,
                  "spherical 2-D/3-D needs the x2 metric tables (artemis_hip_metric_fill)");
  }
  if (!p->geom) return fail(ARTEMIS_HIP_EINVAL, "null geom table");
  if (p->gas.nspecies < 0 || p->dust.nspecies < 0 || (p->gas.nspecies == 0 && p->dust.nspecies == 0))
    return fail(ARTEMIS_HIP_EINVAL, "no fluid species in pack");
  if (p->gas.nspecies > 0 && !(p->gm1 > 0.0))
    return fail(ARTEMIS_HIP_EINVAL, "gm1 must be positive (ideal gas)");
  (void)need_gas_cons;
  return device_ready();
}
int validate_fluid(const artemis_pack_t *p, int fluid, int pcm) {
  if (fluid != ARTEMIS_GAS && fluid != ARTEMIS_DUST)
    return fail(ARTEMIS_HIP_EINVAL, "Fluid type not recognized!");
  const artemis_fluid_pack_t &f = (fluid == ARTEMIS_GAS) ? p->gas : p->dust;
  if (f.nspecies == 0) return 0;
  if (f.recon < ARTEMIS_PCM || f.recon > ARTEMIS_PPM)
    return fail(ARTEMIS_HIP_EINVAL, "Reconstruction method not recognized!");
  if (f.riemann < ARTEMIS_HLLC || f.riemann > ARTEMIS_LLF)
    return fail(ARTEMIS_HIP_EINVAL, "Riemann solver not recognized!");
  if (fluid == ARTEMIS_DUST && f.riemann == ARTEMIS_HLLC)
    return fail(ARTEMIS_HIP_EINVAL, "Riemann solver (dust) not recognized."); // dust.cpp:77-85
  const int recon = pcm ? ARTEMIS_PCM : f.recon;
  const int need = (recon == ARTEMIS_PCM) ? 1 : ((recon == ARTEMIS_PLM) ? 2 : 3);
  if (p->nghost < need)
    return fail(ARTEMIS_HIP_EINVAL, "%s requires at least %d ghost cells.",
                recon == ARTEMIS_PCM ? "PCM" : (recon == ARTEMIS_PLM ? "PLM" : "PPM"), need);
  return 0;
}
// ApplyUpdate / DeepCopyConservedData read BOTH registers: a pack whose cons1 tables were never filled from the u1
// MeshData (artemis_driver.cpp:137-139) is an error here, not a device fault
int validate_registers(const artemis_pack_t *p, const char *task) {
  for (const artemis_fluid_pack_t *f : {&p->gas, &p->dust})
    if (f->nspecies > 0 && (!f->cons0 || !f->cons1))
      return fail(ARTEMIS_HIP_EINVAL, "%s needs the cons0 (u0) and cons1 (u1) tables of every fluid in the pack", task);
  return 0;
}
inline hipStream_t S(void *s) { return static_cast<hipStream_t>(s); }
int after_launch(const char *what) { return check_hip(hipGetLastError(), what); }

// device scalar + pinned mirror for the synchronous dt entry point
struct DtScratch {
  double *dev = nullptr;
  double *host = nullptr;
};
thread_local DtScratch g_dt;
int ensure_dt_scratch() {
  if (g_dt.dev) return 0;
  if (int rc = check_hip(hipMalloc(reinterpret_cast<void **>(&g_dt.dev), sizeof(double)), "hipMalloc"))
    return rc;
  return check_hip(hipHostMalloc(reinterpret_cast<void **>(&g_dt.host), sizeof(double)),
                   "hipHostMalloc");
}
} // namespace

extern "C" {

const char *artemis_hip_last_error(void) { return g_err.c_str(); }
const char *artemis_hip_version(void) { return "artemis_hip 0.1 (gfx950, fp64, -ffp-contract=off)"; }
int artemis_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int artemis_hip_calculate_fluxes(const artemis_pack_t *p, int fluid, int pcm, void *stream) {
  if (int rc = validate(p)) return rc;
  if (int rc = validate_fluid(p, fluid, pcm)) return rc;
  const artemis_fluid_pack_t &f = (fluid == ARTEMIS_GAS) ? p->gas : p->dust;
  if (f.nspecies == 0) return 0;
  const artemis::PackView P = artemis::make_pack_view(*p);
  const int recon = pcm ? ARTEMIS_PCM : f.recon;
  // gas on Cartesian blocks at least a tile wide: the LDS-staged march of the fused stage with the task's stores
  if (fluid == ARTEMIS_GAS && artemis::fused_flux_covers(P, recon) && p->gas.flux[0] && p->gas.pflux[0] && p->gas.vface[0]) {
    if (artemis::launch_flux_fused(P, f.riemann, recon, S(stream)) == 0) return after_launch("CalculateFluxes (tile march)");
  }
  artemis::launch_calculate_fluxes(P, fluid, f.riemann, recon, S(stream));
  return after_launch("CalculateFluxes");
}

int artemis_hip_apply_update(const artemis_pack_t *p, double gam0, double gam1, double beta_dt,
                             void *stream) {
  if (int rc = validate(p)) return rc;
  if (int rc = validate_registers(p, "ApplyUpdate")) return rc;
  artemis::launch_apply_update(artemis::make_pack_view(*p), gam0, gam1, beta_dt, S(stream));
  return after_launch("ApplyUpdate");
}

int artemis_hip_flux_source(const artemis_pack_t *p, int fluid, double dt, void *stream) {
  if (int rc = validate(p)) return rc;
  if (fluid != ARTEMIS_GAS && fluid != ARTEMIS_DUST)
    return fail(ARTEMIS_HIP_EINVAL, "Fluid type not recognized!");
  // Dust is pressureless: Dust::FluxSource returns immediately for Cartesian (dust.cpp:310-323).
  if ((fluid == ARTEMIS_DUST ? p->dust.nspecies : p->gas.nspecies) == 0) return 0;
  if (fluid == ARTEMIS_DUST && p->coords == ARTEMIS_CARTESIAN) return 0;
  if (p->nghost < 1) return fail(ARTEMIS_HIP_EINVAL, "FluxSource needs >= 1 ghost cell");
  artemis::launch_flux_source(artemis::make_pack_view(*p), fluid, dt, S(stream));
  return after_launch("FluxSource");
}

int artemis_hip_set_aux(const artemis_pack_t *p, void *stream) {
  if (int rc = validate(p)) return rc;
  if (p->gas.nspecies == 0) return 0; // fill_derived.cpp:37-38
  artemis::launch_set_aux(artemis::make_pack_view(*p), S(stream));
  return after_launch("SetAuxillaryFields");
}

int artemis_hip_cons_to_prim(const artemis_pack_t *p, void *stream) {
  if (int rc = validate(p)) return rc;
  artemis::launch_cons_to_prim(artemis::make_pack_view(*p), S(stream));
  return after_launch("ConsToPrim");
}

int artemis_hip_prim_to_cons(const artemis_pack_t *p, void *stream) {
  if (int rc = validate(p)) return rc;
  artemis::launch_prim_to_cons(artemis::make_pack_view(*p), S(stream));
  return after_launch("PrimToCons");
}

int artemis_hip_prim_to_cons_ghosts(const artemis_pack_t *p, void *stream) {
  if (int rc = validate(p)) return rc;
  artemis::launch_prim_to_cons(artemis::make_pack_view(*p), S(stream), true);
  return after_launch("PrimToCons (ghost zones)");
}

int artemis_hip_deep_copy_conserved(const artemis_pack_t *p, void *stream) {
  if (int rc = validate(p)) return rc;
  if (int rc = validate_registers(p, "DeepCopyConservedData")) return rc;
  artemis::launch_deep_copy(artemis::make_pack_view(*p), S(stream));
  return after_launch("DeepCopyConservedData");
}

int artemis_hip_estimate_dt_async(const artemis_pack_t *p, int fluid, double cfl, double *dt_dev,
                                  void *stream) {
  if (int rc = validate(p)) return rc;
  if (fluid != ARTEMIS_GAS && fluid != ARTEMIS_DUST)
    return fail(ARTEMIS_HIP_EINVAL, "Fluid type not recognized!");
  if (!dt_dev) return fail(ARTEMIS_HIP_EINVAL, "null dt_dev");
  const artemis_fluid_pack_t &f = (fluid == ARTEMIS_GAS) ? p->gas : p->dust;
  if (f.nspecies == 0) return 0;
  artemis::launch_estimate_dt(artemis::make_pack_view(*p), fluid, cfl, dt_dev, S(stream));
  return after_launch("EstimateTimestepMesh");
}

int artemis_hip_estimate_dt(const artemis_pack_t *p, int fluid, double cfl, double *dt_out,
                            void *stream) {
  if (!dt_out) return fail(ARTEMIS_HIP_EINVAL, "null dt_out");
  if (int rc = validate(p)) return rc;
  if (int rc = ensure_dt_scratch()) return rc;
  *g_dt.host = DBL_MAX;
  if (int rc = check_hip(hipMemcpyAsync(g_dt.dev, g_dt.host, sizeof(double), hipMemcpyHostToDevice,
                                        S(stream)), "hipMemcpyAsync"))
    return rc;
  if (int rc = artemis_hip_estimate_dt_async(p, fluid, cfl, g_dt.dev, stream)) return rc;
  if (int rc = check_hip(hipMemcpyAsync(g_dt.host, g_dt.dev, sizeof(double), hipMemcpyDeviceToHost,
                                        S(stream)), "hipMemcpyAsync"))
    return rc;
  if (int rc = check_hip(hipStreamSynchronize(S(stream)), "hipStreamSynchronize")) return rc;
  *dt_out = *g_dt.host;
  return 0;
}

int artemis_hip_apply_bc(const artemis_pack_t *p, const int *bc, const artemis_bc_params_t *params,
                         void *stream) {
  if (int rc = validate(p)) return rc;
  if (!bc) return fail(ARTEMIS_HIP_EINVAL, "null bc array");
  for (int i = 0; i < 6 * p->nblocks; ++i) {
    if (bc[i] < ARTEMIS_BC_PERIODIC || bc[i] > ARTEMIS_BC_DISK_VISC)
      return fail(ARTEMIS_HIP_EINVAL, "unknown boundary flag %d", bc[i]);
    const int d = (i % 6) / 2;
    if (bc[i] == ARTEMIS_BC_DISK_VISC) { // disk.hpp:420-424, :452-455
      if (d != 0)
        return fail(ARTEMIS_HIP_EINVAL, "Viscous boundary conditions only work for the inner or outer radial boundary");
      if (p->coords == ARTEMIS_CARTESIAN)
        return fail(ARTEMIS_HIP_EINVAL, "Viscous boundary conditions only work with spherical/cylindrical radial boundaries");
    }
    if (bc[i] == ARTEMIS_BC_IC || bc[i] == ARTEMIS_BC_DISK_EXTRAP || bc[i] == ARTEMIS_BC_DISK_VISC) {
      if (!params) return fail(ARTEMIS_HIP_EINVAL, "disk conditions need artemis_bc_params_t");
      if (bc[i] == ARTEMIS_BC_IC && ((p->gas.nspecies && !params->ic_gas) || (p->dust.nspecies && !params->ic_dust)))
        return fail(ARTEMIS_HIP_EINVAL, "ic condition: ic_gas / ic_dust tables are required");
      const int nxd[3] = {p->nx1, p->nx2, p->nx3};
      if (bc[i] != ARTEMIS_BC_IC && nxd[d] < 2)
        return fail(ARTEMIS_HIP_EINVAL, "disk extrap condition needs two active zones along the face normal");
    }
    if (bc[i] == ARTEMIS_BC_CONDUCTIVE) {
      if ((p->coords == ARTEMIS_CYLINDRICAL || p->coords == ARTEMIS_AXISYMMETRIC) && !p->metric)
        return fail(ARTEMIS_HIP_EINVAL, // Coords::Distance needs cos/sin of the azimuth
                    "conductive boundary condition: cylindrical / axisymmetric blocks need the metric tables");
      if (!params) return fail(ARTEMIS_HIP_EINVAL, "conductive conditions need artemis_bc_params_t");
      if (params->cond_type != ARTEMIS_CONDUCTIVITY_PLAW && params->cond_type != ARTEMIS_THERMALDIFF_PLAW)
        return fail(ARTEMIS_HIP_EINVAL, // conduction.hpp:246-248
                    "Chosen conductivity type is not compatible with conductivity boundaries");
      if (!(params->cond_cv > 0.0)) return fail(ARTEMIS_HIP_EINVAL, "conductive conditions: cv must be positive");
      if (p->gas.nspecies > 1) return fail(ARTEMIS_HIP_EINVAL, "Cond pgen requires a single gas species.");
    }
    if (bc[i] == ARTEMIS_BC_STRAT_EXTRAP || bc[i] == ARTEMIS_BC_STRAT_INFLOW) {
      const bool extrap = (bc[i] == ARTEMIS_BC_STRAT_EXTRAP);
      if ((extrap && d == 1) || (!extrap && d != 1)) // problem_modifier.hpp:117-128
        return fail(ARTEMIS_HIP_EINVAL, "strat conditions: extrap belongs to x1/x3, inflow to x2 faces");
      if (extrap && d == 2 && p->nx3 < 2)
        return fail(ARTEMIS_HIP_EINVAL, "strat extrap condition on x3 faces needs two active zones in x3");
      if (p->coords != ARTEMIS_CARTESIAN)
        return fail(ARTEMIS_HIP_EINVAL, "problem = strat only works for Cartesian Coordinates!");
      if (p->gas.nspecies > 1)
        return fail(ARTEMIS_HIP_EINVAL, "strat conditions fill gas species 0 only (strat.hpp:188-199)");
      if (!params) return fail(ARTEMIS_HIP_EINVAL, "strat conditions need artemis_bc_params_t");
    }
  }
  const int rc = artemis::launch_apply_bc(artemis::make_pack_view(*p), bc, params, S(stream));
  if (rc == 1) return fail(ARTEMIS_HIP_EDEVICE, "could not read pointer tables from the device");
  if (rc == 2) return fail(ARTEMIS_HIP_EUNSUPPORTED, "too many FillGhost variables (max 64)");
  if (rc == 3) return fail(ARTEMIS_HIP_EUNSUPPORTED, "boundary conditions: a block's ghost shell holds 2^31 zones or more");
  return after_launch("ApplyBoundaryConditions");
}

int artemis_hip_external_gravity(const artemis_pack_t *p, const artemis_gravity_t *g, double time,
                                 double dt, void *stream) {
  if (int rc = validate(p)) return rc;
  if (!g) return fail(ARTEMIS_HIP_EINVAL, "null gravity parameters");
  if (g->type != ARTEMIS_GRAVITY_UNIFORM && g->type != ARTEMIS_GRAVITY_POINT && g->type != ARTEMIS_GRAVITY_BINARY)
    return fail(ARTEMIS_HIP_EUNSUPPORTED, "gravity type %d (nbody) is not built", g->type);
  if (g->type == ARTEMIS_GRAVITY_BINARY &&
      (p->coords == ARTEMIS_AXISYMMETRIC || p->coords == ARTEMIS_SPHERICAL1D || p->coords == ARTEMIS_SPHERICAL2D))
    return fail(ARTEMIS_HIP_EINVAL, "Binary gravity is not compatable with axisymmetric coordinates!"); // gravity.cpp:82-83
  if (g->type == ARTEMIS_GRAVITY_BINARY && p->coords == ARTEMIS_CYLINDRICAL && !p->metric)
    return fail(ARTEMIS_HIP_EINVAL, "binary gravity on cylindrical blocks needs the metric tables");
  if (g->type == ARTEMIS_GRAVITY_POINT) {
    if (p->coords == ARTEMIS_CYLINDRICAL && !p->metric) // ConvertToCartWithVec: cos / sin of x2v
      return fail(ARTEMIS_HIP_EINVAL, "point-mass gravity on cylindrical blocks needs the metric tables");
    const bool axi = p->coords == ARTEMIS_AXISYMMETRIC || p->coords == ARTEMIS_SPHERICAL1D ||
                     p->coords == ARTEMIS_SPHERICAL2D;
    if (axi && !(g->pos[0] == 0.0 && g->pos[1] == 0.0 && g->pos[2] == 0.0)) // gravity.cpp:66-70
      return fail(ARTEMIS_HIP_EINVAL,
                  "In axisymmetric coordinates, the point mass must be at the origin!");
  }
  if (!((time >= g->tstart) && (time < g->tstop))) return 0; // gravity.cpp:134
  artemis::launch_external_gravity(artemis::make_pack_view(*p), *g, dt, S(stream));
  return after_launch("ExternalGravity");
}

int artemis_hip_nbody_gravity(const artemis_pack_t *p, const artemis_nbody_particle_t *particles, int npart, double omf,
                              double time, double dt, double *force, void *stream) {
  (void)time;
  if (int rc = validate(p)) return rc;
  if (npart < 0 || (npart > 0 && (!particles || !force))) return fail(ARTEMIS_HIP_EINVAL, "nbody gravity: null particles / force");
  if (p->coords != ARTEMIS_CARTESIAN && p->coords != ARTEMIS_CYLINDRICAL && p->coords != ARTEMIS_SPHERICAL3D)
    return fail(ARTEMIS_HIP_EINVAL, "NBody does not work with axisymmetric coordinates!"); // nbody.cpp:61-62
  if (p->coords != ARTEMIS_CARTESIAN && !p->metric)
    return fail(ARTEMIS_HIP_EINVAL, "nbody gravity on curvilinear blocks needs the metric tables");
  if (!(dt != 0.0)) return fail(ARTEMIS_HIP_EINVAL, "nbody gravity: dt must be non-zero (forces are per unit time)");
  if (npart == 0) return 0;
  const artemis::PackView P = artemis::make_pack_view(*p);
  const int grid = artemis::nbody_grid(P);
  const size_t pbytes = sizeof(artemis_nbody_particle_t) * npart, rbytes = sizeof(double) * 7 * static_cast<size_t>(npart) * grid;
  void *dev = nullptr;
  if (int rc = check_hip(hipMalloc(&dev, pbytes + rbytes + 64), "hipMalloc")) return rc;
  artemis_nbody_particle_t *pl_dev = static_cast<artemis_nbody_particle_t *>(dev);
  double *partial = reinterpret_cast<double *>(static_cast<char *>(dev) + ((pbytes + 63) / 64) * 64);
  std::vector<double> host(static_cast<size_t>(7) * npart * grid);
  int rc = check_hip(hipMemcpyAsync(pl_dev, particles, pbytes, hipMemcpyHostToDevice, S(stream)), "h2d particles");
  if (!rc) {
    artemis::launch_nbody_gravity(P, pl_dev, npart, omf, dt, partial, S(stream));
    rc = after_launch("NBodyGravity");
  }
  if (!rc) rc = check_hip(hipMemcpyAsync(host.data(), partial, rbytes, hipMemcpyDeviceToHost, S(stream)), "d2h partials");
  if (!rc) rc = check_hip(hipStreamSynchronize(S(stream)), "sync");
  (void)hipFree(dev);
  if (rc) return rc;
  for (int n = 0; n < npart; ++n) {
    if (!particles[n].couple) continue;
    for (int q = 0; q < 7; ++q) {
      double sum = 0.0;
      for (int w = 0; w < grid; ++w) sum += host[(static_cast<size_t>(n) * grid + w) * 7 + q];
      force[7 * n + q] += sum;
    }
  }
  return 0;
}

static int validate_nbody_coords(const artemis_pack_t *p) {
  if (p->coords != ARTEMIS_CARTESIAN && p->coords != ARTEMIS_CYLINDRICAL && p->coords != ARTEMIS_SPHERICAL3D)
    return fail(ARTEMIS_HIP_EINVAL, "NBody does not work with axisymmetric coordinates!"); // nbody.cpp:61-62
  if (p->coords != ARTEMIS_CARTESIAN && !p->metric)
    return fail(ARTEMIS_HIP_EINVAL, "nbody gravity on curvilinear blocks needs the metric tables");
  return 0;
}
int artemis_hip_nbody_force_scratch(const artemis_pack_t *p) {
  if (validate(p)) return -1;
  return artemis::nbody_grid(artemis::make_pack_view(*p));
}
int artemis_hip_nbody_force_sums(const artemis_pack_t *p, const artemis_nbody_particle_t *particles_dev, int npart, double omf,
                                 double dt, const double *dt_dev, double *scratch_dev, double *force_dev, void *stream) {
  if (int rc = validate(p)) return rc;
  if (npart < 0 || npart > 128 || (npart > 0 && (!particles_dev || !scratch_dev || !force_dev)))
    return fail(ARTEMIS_HIP_EINVAL, "nbody force sums: 0 <= npart <= 128, particles / scratch / force are DEVICE arrays");
  if (int rc = validate_nbody_coords(p)) return rc;
  if (!dt_dev && !(dt != 0.0)) return fail(ARTEMIS_HIP_EINVAL, "nbody gravity: dt must be non-zero (forces are per unit time)");
  if (npart == 0) return 0;
  const artemis::PackView P = artemis::make_pack_view(*p);
  if (!artemis::nbody_force_sums_covers(P))
    return fail(ARTEMIS_HIP_EUNSUPPORTED, "nbody force sums: at most one species per fluid (use artemis_hip_nbody_gravity)");
  artemis::launch_nbody_force_sums(P, particles_dev, npart, omf, dt, dt_dev, scratch_dev, force_dev, S(stream));
  return after_launch("nbody force sums");
}

int artemis_hip_rotating_frame_force(const artemis_pack_t *p, double omega, double qshear,
                                     double time, double dt, void *stream) {
  (void)time;
  if (int rc = validate(p)) return rc;
  if (omega == 0.0) // rotating_frame.cpp:31-32
    return fail(ARTEMIS_HIP_EINVAL, "rotating_frame/omega cannot be zero!");
  if (p->coords != ARTEMIS_CARTESIAN) {
    if (qshear != 0.0) // rotating_frame.cpp:34-38
      return fail(ARTEMIS_HIP_EINVAL,
                  "rotating_frame/qshear must be zero for non-Cartesian coordinate systems!");
    // RotatingFrameImpl reads the mass fluxes CalculateFluxes left in flux[d] (rotating_frame_impl.hpp:107-111)
    const int ndim = (p->nx3 > 1) ? 3 : ((p->nx2 > 1) ? 2 : 1);
    for (int d = 0; d < ndim; ++d)
      if ((p->gas.nspecies && !p->gas.flux[d]) || (p->dust.nspecies && !p->dust.flux[d]))
        return fail(ARTEMIS_HIP_EINVAL, "rotating frame (curvilinear): flux tables are required");
    artemis::launch_rotating_frame(artemis::make_pack_view(*p), omega, dt, S(stream));
    return after_launch("RotatingFrameForce");
  }
  artemis::launch_shearing_box(artemis::make_pack_view(*p), omega, qshear, dt, S(stream));
  return after_launch("RotatingFrameForce");
}

int artemis_hip_cooling_table_fill(const artemis_pack_t *p, const double *geom_host, const double *metric_host,
                                   const artemis_cooling_t *c, int block, double *tref_host, double *beta_host) {
  if (!p || !geom_host || !c || !tref_host || !beta_host) return fail(ARTEMIS_HIP_EINVAL, "null argument");
  if (block < 0 || block >= p->nblocks) return fail(ARTEMIS_HIP_EINVAL, "bad block index %d", block);
  if (!metric_host && (p->coords == ARTEMIS_SPHERICAL2D || p->coords == ARTEMIS_SPHERICAL3D))
    return fail(ARTEMIS_HIP_EINVAL, "cooling table: spherical 2-D/3-D blocks need the host metric table");
  const artemis::PackView P = artemis::make_pack_view(*p);
  const double *m = metric_host ? metric_host + block * artemis::metric_block_stride(P.nj, P.nk) : nullptr;
  for (int k = 0; k < P.nk; ++k)
    for (int j = 0; j < P.nj; ++j)
      for (int i = 0; i < P.ni; ++i) {
        const artemis::DCoords co = artemis::coords_of(p->coords, geom_host + 6 * block, m, P.nj, P.nk, k, j, i);
        double xv[3];
        co.centre(xv);
        const artemis::Frame fr = artemis::cyl_frame(co.sys, xv, co.cv, co.sv);
        // TemperatureProfile<GEOM, powerlaw> (cooling.hpp:47-58), beta (beta_cooling.cpp:98-99)
        const double T0 = c->tfloor + c->tcyl * std::pow(fr.x[0], c->cyl_plaw) +
                          c->tsph * std::pow(artemis::sph_radius(co.sys, xv), c->sph_plaw);
        const double efac = (T0 > 0.) ? std::exp(-c->exp_scale * fr.x[2] * fr.x[2] / T0) : 1.;
        const long q = (static_cast<long>(k) * P.nj + j) * P.ni + i;
        tref_host[q] = T0, beta_host[q] = c->beta_min + c->beta0 * efac;
      }
  return 0;
}
int artemis_hip_cooling_source(const artemis_pack_t *p, const artemis_cooling_t *c, double time, double dt,
                               void *stream) {
  (void)time;
  if (int rc = validate(p)) return rc;
  if (!c) return fail(ARTEMIS_HIP_EINVAL, "null cooling parameters");
  if (p->gas.nspecies < 1) return 0;
  if (!c->tref || !c->beta)
    return fail(ARTEMIS_HIP_EINVAL, "cooling: tref / beta tables are required (artemis_hip_cooling_table_fill)");
  if (!(c->cv > 0.0)) return fail(ARTEMIS_HIP_EINVAL, "cooling: specific heat cv must be positive");
  if (!p->gas.cons0) return fail(ARTEMIS_HIP_EINVAL, "cooling: gas.cons0 table is required");
  artemis::launch_cooling(artemis::make_pack_view(*p), *c, dt, S(stream));
  return after_launch("CoolingSource");
}

static int validate_damp_visc(const artemis_drag_t *d) { // drag.cpp:113-121, 137-157
  if (!d->damp_visc) return ARTEMIS_HIP_OK;
  const int t = d->damp_visc->type;
  if (t != ARTEMIS_VISCOSITY_PLAW && t != ARTEMIS_VISCOSITY_ALPHA)
    return fail(ARTEMIS_HIP_EINVAL, "The chosen viscosity model does not work with damping");
  if (t == ARTEMIS_VISCOSITY_ALPHA && !d->damp_visc->radial)
    return fail(ARTEMIS_HIP_EINVAL, "damp_to_visc: alpha viscosity needs its radial table (artemis_hip_diffusion_radial_fill)");
  return ARTEMIS_HIP_OK;
}
int artemis_hip_drag_source(const artemis_pack_t *p, const artemis_drag_t *d, double time, double dt,
                            void *stream) {
  (void)time;
  if (int rc = validate(p)) return rc;
  if (!d) return fail(ARTEMIS_HIP_EINVAL, "null drag parameters");
  if (d->type != ARTEMIS_DRAG_SIMPLE_DUST && d->type != ARTEMIS_DRAG_SELF)
    return fail(ARTEMIS_HIP_EINVAL, "Bad choice of drag type"); // drag.hpp:66
  if (d->type == ARTEMIS_DRAG_SIMPLE_DUST) {
    if (p->gas.nspecies < 1 || p->dust.nspecies < 1) // drag.cpp:71-72
      return fail(ARTEMIS_HIP_EINVAL, "drag type simple_dust requires do_gas = do_dust = true");
    if (p->dust.nspecies > ARTEMIS_MAX_DUST_SPECIES)
      return fail(ARTEMIS_HIP_EUNSUPPORTED, "simple_dust drag: more than %d dust species",
                  ARTEMIS_MAX_DUST_SPECIES);
    if (d->model != ARTEMIS_DRAG_CONSTANT && d->model != ARTEMIS_DRAG_STOKES)
      return fail(ARTEMIS_HIP_EINVAL, "bad type for stopping time model"); // drag.hpp:150
  }
  for (int i = 0; i < 3; ++i) // drag.hpp:110-115
    if (d->gas.irate[i] < 0.0 || d->dust.irate[i] < 0.0 || d->gas.ix[i] > d->gas.ox[i] ||
        d->dust.ix[i] > d->dust.ox[i])
      return fail(ARTEMIS_HIP_EINVAL, "bad damping bounds / rates");
  if (int rc = validate_damp_visc(d)) return rc;
  artemis::launch_drag_source(artemis::make_pack_view(*p), *d, dt, nullptr, S(stream));
  return after_launch("DragSource");
}

// Host-side metric tables (see include/artemis_hip.h and csrc/geometry_core.hpp: per block six x2
// rows of nj+1 doubles, then two x3 rows of nk+1).  Expressions follow spherical.hpp:61-68 (x2v),
// :53-55, :88-104 (the sine arguments) and the ConvertCoordsToCart of each system.
long artemis_hip_metric_count(const artemis_pack_t *p) {
  if (!p) return -1;
  if (p->coords == ARTEMIS_CARTESIAN || p->coords == ARTEMIS_SPHERICAL1D) return 0;
  const int nj = p->nx2 + ((p->nx2 > 1) ? 2 * p->nghost : 0);
  const int nk = p->nx3 + ((p->nx3 > 1) ? 2 * p->nghost : 0);
  return static_cast<long>(p->nblocks) * (6L * (nj + 1) + 2L * (nk + 1));
}
int artemis_hip_metric_fill(const artemis_pack_t *p, const double *geom_host, double *out_host) {
  if (!p || !geom_host || !out_host) return fail(ARTEMIS_HIP_EINVAL, "null argument");
  if (artemis_hip_metric_count(p) == 0) return 0;
  const int nj = p->nx2 + ((p->nx2 > 1) ? 2 * p->nghost : 0);
  const int nk = p->nx3 + ((p->nx3 > 1) ? 2 * p->nghost : 0);
  const int st = nj + 1, st3 = nk + 1;
  const long stride = 6L * st + 2L * st3;
  const bool sph23 = (p->coords == ARTEMIS_SPHERICAL2D || p->coords == ARTEMIS_SPHERICAL3D);
  for (int b = 0; b < p->nblocks; ++b) {
    const double f0 = geom_host[6 * b + 2], dx = geom_host[6 * b + 3];
    double *m = out_host + b * stride;
    for (long q = 0; q < stride; ++q) m[q] = 0.0;
    if (sph23) {
      for (int j = 0; j <= nj; ++j) {
        const double xf = f0 + j * dx;
        m[0 * st + j] = std::cos(xf);
        m[1 * st + j] = std::sin(xf);
      }
      for (int j = 0; j < nj; ++j) {
        const double x0 = f0 + j * dx, x1 = f0 + (j + 1) * dx;
        const double ctm = m[0 * st + j], ctp = m[0 * st + j + 1];
        const double dst = m[1 * st + j + 1] - m[1 * st + j];
        const double x2v = (dst - x1 * ctp + x0 * ctm) / std::abs(ctm - ctp);
        m[2 * st + j] = x2v;
        m[3 * st + j] = std::sin(x2v);
        m[4 * st + j] = std::sin(0.5 * (x0 + x1));
        m[5 * st + j] = std::cos(x2v);
      }
    } else if (p->coords == ARTEMIS_CYLINDRICAL) { // azimuth of the cell centre (cylindrical.hpp:88-92)
      for (int j = 0; j < nj; ++j) {
        const double x2v = 0.5 * ((f0 + j * dx) + (f0 + (j + 1) * dx));
        m[2 * st + j] = x2v, m[3 * st + j] = std::sin(x2v), m[5 * st + j] = std::cos(x2v);
      }
    }
    if (p->coords == ARTEMIS_SPHERICAL3D || p->coords == ARTEMIS_AXISYMMETRIC) {
      const double g0 = geom_host[6 * b + 4], dz = geom_host[6 * b + 5];
      double *m3 = m + 6L * st;
      for (int k = 0; k < nk; ++k) {
        const double x3v = 0.5 * ((g0 + k * dz) + (g0 + (k + 1) * dz));
        m3[0 * st3 + k] = std::cos(x3v), m3[1 * st3 + k] = std::sin(x3v);
      }
    }
  }
  return 0;
}

long artemis_hip_halo_count_ext(const artemis_pack_t *p, int face, int extended) {
  if (!p || face < 0 || face > 5) return -1;
  return artemis::halo_count(artemis::make_pack_view(*p), face, extended ? 1 : 0);
}
long artemis_hip_halo_count(const artemis_pack_t *p, int face) { return artemis_hip_halo_count_ext(p, face, 0); }
static int halo_common(const artemis_pack_t *p, int block, int face, double *buf, int unpack,
                       int extended, void *stream) {
  if (int rc = validate(p)) return rc;
  if (block < 0 || block >= p->nblocks) return fail(ARTEMIS_HIP_EINVAL, "bad block %d", block);
  if (face < 0 || face > 5) return fail(ARTEMIS_HIP_EINVAL, "bad face %d", face);
  const artemis::PackView P = artemis::make_pack_view(*p);
  if (face / 2 >= P.ndim) return fail(ARTEMIS_HIP_EINVAL, "face %d is not an active direction", face);
  if (!buf) return fail(ARTEMIS_HIP_EINVAL, "null halo buffer");
  const int rc = artemis::launch_halo(P, block, face, buf, unpack, extended ? 1 : 0, S(stream));
  if (rc == 1) return fail(ARTEMIS_HIP_EDEVICE, "could not read pointer tables from the device");
  if (rc == 2) return fail(ARTEMIS_HIP_EUNSUPPORTED, "too many FillGhost variables (max 64)");
  return after_launch(unpack ? "halo unpack" : "halo pack");
}
int artemis_hip_halo_pack(const artemis_pack_t *p, int block, int face, double *buf, void *stream) {
  return halo_common(p, block, face, buf, 0, 0, stream);
}
int artemis_hip_halo_unpack(const artemis_pack_t *p, int block, int face, const double *buf,
                            void *stream) {
  return halo_common(p, block, face, const_cast<double *>(buf), 1, 0, stream);
}
int artemis_hip_halo_pack_ext(const artemis_pack_t *p, int block, int face, int extended, double *buf,
                              void *stream) {
  return halo_common(p, block, face, buf, 0, extended, stream);
}
int artemis_hip_halo_unpack_ext(const artemis_pack_t *p, int block, int face, int extended,
                                const double *buf, void *stream) {
  return halo_common(p, block, face, const_cast<double *>(buf), 1, extended, stream);
}

// ---- multilevel block-graph data path (kernels_amr.hip) ------------------------------------------------
static int validate_ml(const artemis_pack_t *p, const artemis_ml_pack_t *ml) {
  if (int rc = validate(p)) return rc;
  if (!ml || !ml->cgeom) return fail(ARTEMIS_HIP_EINVAL, "multilevel: null coarse-buffer pack / edge table");
  if ((p->gas.nspecies && !ml->gas_coarse) || (p->dust.nspecies && !ml->dust_coarse))
    return fail(ARTEMIS_HIP_EINVAL, "multilevel: coarse-buffer tables are required");
  if (p->nghost % 2 != 0) return fail(ARTEMIS_HIP_EINVAL, "multilevel meshes need an even number of ghost zones");
  if (p->nx1 % 2 != 0 || (p->nx2 > 1 && p->nx2 % 2 != 0) || (p->nx3 > 1 && p->nx3 % 2 != 0))
    return fail(ARTEMIS_HIP_EINVAL, "multilevel meshes need an even number of zones per mesh block");
  if ((p->coords == ARTEMIS_SPHERICAL2D || p->coords == ARTEMIS_SPHERICAL3D) && !ml->cmetric)
    return fail(ARTEMIS_HIP_EINVAL, "multilevel: spherical coarse buffers need their metric tables");
  return 0;
}
int artemis_hip_ml_exchange(const artemis_pack_t *p, const artemis_ml_pack_t *ml, const artemis_ml_op_t *ops_dev, int nops,
                            double *sendbuf, const double *recvbuf, void *stream) {
  if (int rc = validate_ml(p, ml)) return rc;
  if (nops < 0 || (nops > 0 && !ops_dev)) return fail(ARTEMIS_HIP_EINVAL, "multilevel: bad operation list");
  artemis::launch_ml_exchange(artemis::make_pack_view(*p), *ml, ops_dev, nops, sendbuf, recvbuf, S(stream));
  return after_launch("ml_exchange");
}
int artemis_hip_ml_flux_correction(const artemis_pack_t *p, const artemis_ml_op_t *ops_dev, int nops, double *sendbuf,
                                   const double *recvbuf, void *stream) {
  if (int rc = validate(p)) return rc;
  if (nops < 0 || (nops > 0 && !ops_dev)) return fail(ARTEMIS_HIP_EINVAL, "multilevel: bad operation list");
  const int ndim = (p->nx3 > 1) ? 3 : ((p->nx2 > 1) ? 2 : 1);
  for (int d = 0; d < ndim; ++d)
    if ((p->gas.nspecies && (!p->gas.flux[d] || !p->gas.pflux[d])) || (p->dust.nspecies && !p->dust.flux[d]))
      return fail(ARTEMIS_HIP_EINVAL, "flux correction: flux tables are required");
  artemis::launch_ml_flux_correction(artemis::make_pack_view(*p), ops_dev, nops, sendbuf, recvbuf, S(stream));
  return after_launch("ml_flux_correction");
}
int artemis_hip_ml_restrict_halos(const artemis_pack_t *p, const artemis_ml_pack_t *ml, const int *blocks_dev, int nblocks,
                                  void *stream) {
  if (int rc = validate_ml(p, ml)) return rc;
  if (nblocks < 0 || (nblocks > 0 && !blocks_dev)) return fail(ARTEMIS_HIP_EINVAL, "multilevel: bad block list");
  artemis::launch_ml_restrict_halos(artemis::make_pack_view(*p), *ml, blocks_dev, nblocks, S(stream));
  return after_launch("ml_restrict_halos");
}
int artemis_hip_ml_prolongate(const artemis_pack_t *p, const artemis_ml_pack_t *ml, const artemis_ml_box_t *boxes_dev,
                              int nboxes, void *stream) {
  if (int rc = validate_ml(p, ml)) return rc;
  if (nboxes < 0 || (nboxes > 0 && !boxes_dev)) return fail(ARTEMIS_HIP_EINVAL, "multilevel: bad box list");
  artemis::launch_ml_prolongate(artemis::make_pack_view(*p), *ml, boxes_dev, nboxes, S(stream));
  return after_launch("ml_prolongate");
}
int artemis_hip_ml_floor_ghosts(const artemis_pack_t *p, const int *blocks_dev, int nblocks, void *stream) {
  if (int rc = validate(p)) return rc;
  if (nblocks < 0 || (nblocks > 0 && !blocks_dev)) return fail(ARTEMIS_HIP_EINVAL, "multilevel: bad block list");
  if ((p->gas.nspecies && !p->gas.prim) || (p->dust.nspecies && !p->dust.prim))
    return fail(ARTEMIS_HIP_EINVAL, "multilevel: primitive tables are required");
  artemis::launch_ml_floor_ghosts(artemis::make_pack_view(*p), blocks_dev, nblocks, S(stream));
  return after_launch("ml_floor_ghosts");
}

long artemis_hip_plm_table_count(const artemis_pack_t *p) {
  if (!p) return 0;
  return artemis::plm_table_count(artemis::make_pack_view(*p));
}
int artemis_hip_plm_table_fill(const artemis_pack_t *p, double *table_dev, void *stream) {
  // geometry only: shape, coordinate system, edge table (no fluid needs to be in the pack)
  if (!p || p->nblocks < 1 || p->nx1 < 1 || p->nx2 < 1 || p->nx3 < 1 || p->nghost < 1 || !p->geom)
    return fail(ARTEMIS_HIP_EINVAL, "plm table: the pack needs its shape and edge table");
  if (p->coords < ARTEMIS_CARTESIAN || p->coords > ARTEMIS_AXISYMMETRIC) return fail(ARTEMIS_HIP_EINVAL, "Invalid artemis/coordinate system!");
  if (device_ready()) return fail(ARTEMIS_HIP_EDEVICE, "no HIP device");
  if (!table_dev) return fail(ARTEMIS_HIP_EINVAL, "plm table: null table");
  if ((p->coords == ARTEMIS_SPHERICAL2D || p->coords == ARTEMIS_SPHERICAL3D) && !p->metric)
    return fail(ARTEMIS_HIP_EINVAL, "plm table: spherical blocks need the metric tables first");
  artemis::launch_plm_table_fill(artemis::make_pack_view(*p), table_dev, S(stream));
  return after_launch("plm_table_fill");
}

// ---- refined meshes on the one-kernel stages: fine-side faces, then the coarse zones next to them redone ----------
static int validate_diffusion(const artemis_pack_t *p, const artemis_diffusion_t *d, bool need_flux);
static int validate_stage_nbody(const artemis_pack_t *p, const artemis_stage_general_args_t *a) {
  if (a->nbody_n < 0 || (a->nbody_n > 0 && !a->nbody_dev)) return fail(ARTEMIS_HIP_EINVAL, "stage: nbody_dev / nbody_n");
  if (a->nbody_n == 0) return 0;
  if (a->gravity) return fail(ARTEMIS_HIP_EINVAL, "stage: N-body gravity and an external gravity type are exclusive (gravity.cpp:60-118)");
  if (a->cooling) return fail(ARTEMIS_HIP_EUNSUPPORTED, "stage: N-body gravity together with cooling runs on the per-task chain");
  if (p->gas.nspecies > 1 || p->dust.nspecies > 1)
    return fail(ARTEMIS_HIP_EUNSUPPORTED, "stage: N-body gravity inside the stage takes at most one species per fluid");
  return validate_nbody_coords(p);
}
static int validate_ml_fix(const artemis_pack_t *p, const artemis_stage_general_args_t *a) {
  if (int rc = validate(p)) return rc;
  if (!a) return fail(ARTEMIS_HIP_EINVAL, "null stage args");
  if (int rc = validate_fluid(p, ARTEMIS_GAS, a->pcm)) return rc;
  if (int rc = validate_fluid(p, ARTEMIS_DUST, a->pcm)) return rc;
  if (a->defer_finish < 0 || a->defer_finish > 2) return fail(ARTEMIS_HIP_EINVAL, "defer_finish must be 0, 1 or 2");
  if (a->drag && !a->defer_finish)
    return fail(ARTEMIS_HIP_EUNSUPPORTED, "refined-mesh fix-up: drag couples the fluids after the update: set defer_finish and run "
                                          "artemis_hip_stage_finish after the fix-up");
  if (int rc = validate_stage_nbody(p, a)) return rc;
  if (a->defer_finish && ((p->gas.nspecies && !p->gas.cons0) || (p->dust.nspecies && !p->dust.cons0)))
    return fail(ARTEMIS_HIP_EINVAL, "defer_finish: cons0 tables are required");
  if (p->gas.nspecies && (!a->gas_in || !a->gas_u1 || !a->gas_out))
    return fail(ARTEMIS_HIP_EINVAL, "refined-mesh fix-up: gas_in / gas_u1 / gas_out are required");
  if (p->dust.nspecies && (!a->dust_in || !a->dust_u1 || !a->dust_out))
    return fail(ARTEMIS_HIP_EINVAL, "refined-mesh fix-up: dust_in / dust_u1 / dust_out are required");
  const int ndim = (p->nx3 > 1) ? 3 : ((p->nx2 > 1) ? 2 : 1);
  for (int d = 0; d < ndim; ++d)
    if ((p->gas.nspecies && (!p->gas.flux[d] || !p->gas.pflux[d] || !p->gas.vface[d])) || (p->dust.nspecies && !p->dust.flux[d]))
      return fail(ARTEMIS_HIP_EINVAL, "refined-mesh fix-up: flux / pflux / vface tables are required");
  return 0;
}
int artemis_hip_ml_face_fluxes(const artemis_pack_t *p, const artemis_stage_general_args_t *a,
                               const artemis_ml_face_box_t *boxes_dev, int nboxes, void *stream) {
  if (int rc = validate_ml_fix(p, a)) return rc;
  if (nboxes < 0 || (nboxes > 0 && !boxes_dev)) return fail(ARTEMIS_HIP_EINVAL, "multilevel: bad box list");
  artemis::launch_ml_face_fluxes(artemis::make_pack_view(*p), *a, p->gas.recon, p->gas.riemann, p->dust.recon,
                                 p->dust.riemann, boxes_dev, nboxes, S(stream));
  return after_launch("ml_face_fluxes");
}
int artemis_hip_ml_stage_fixup(const artemis_pack_t *p, const artemis_stage_general_args_t *a,
                               const artemis_ml_fix_cell_t *cells_dev, int ncells, void *stream) {
  if (int rc = validate_ml_fix(p, a)) return rc;
  if (ncells < 0 || (ncells > 0 && !cells_dev)) return fail(ARTEMIS_HIP_EINVAL, "multilevel: bad zone list");
  if (a->gravity) {
    const artemis_gravity_t *g = a->gravity;
    if (g->type != ARTEMIS_GRAVITY_UNIFORM && g->type != ARTEMIS_GRAVITY_POINT && g->type != ARTEMIS_GRAVITY_BINARY)
      return fail(ARTEMIS_HIP_EUNSUPPORTED, "gravity type %d (nbody) is not built", g->type);
  }
  if (a->diffusion)
    if (int rc = validate_diffusion(p, a->diffusion, true)) return rc;
  if (a->cooling && p->gas.nspecies && (!a->cooling->tref || !a->cooling->beta))
    return fail(ARTEMIS_HIP_EINVAL, "cooling: tref / beta tables are required (artemis_hip_cooling_table_fill)");
  artemis::launch_ml_stage_fixup(artemis::make_pack_view(*p), *a, p->gas.recon, p->gas.riemann, p->dust.recon,
                                 p->dust.riemann, cells_dev, ncells, S(stream));
  return after_launch("ml_stage_fixup");
}

size_t artemis_hip_redo_scratch_bytes(const artemis_pack_t *p) {
  if (!p) return 0;
  const size_t zones = static_cast<size_t>(p->nblocks) * p->nx1 * p->nx2 * p->nx3;
  return 64 + 2 * zones * sizeof(unsigned long long);
}
int artemis_hip_stage_fused(const artemis_pack_t *p, const artemis_stage_args_t *a, void *stream) {
  if (int rc = validate(p)) return rc;
  if (!a) return fail(ARTEMIS_HIP_EINVAL, "null stage args");
  if (p->gas.nspecies != 1 || p->dust.nspecies != 0)
    return fail(ARTEMIS_HIP_EUNSUPPORTED, "fused stage: one gas species, no dust (DESIGN.md)");
  if (p->coords != ARTEMIS_CARTESIAN)
    return fail(ARTEMIS_HIP_EUNSUPPORTED,
                "fused stage: Cartesian only; curvilinear systems use the per-task kernels");
  if (int rc = validate_fluid(p, ARTEMIS_GAS, a->pcm)) return rc;
  if (!a->prim_in || !a->prim_u1 || !a->prim_out)
    return fail(ARTEMIS_HIP_EINVAL, "fused stage: prim_in / prim_u1 / prim_out are required");
  if (a->prim_in == a->prim_out)
    return fail(ARTEMIS_HIP_EINVAL, "fused stage: prim_out must not alias prim_in");
  if (a->region < 0 || a->region > 2) return fail(ARTEMIS_HIP_EINVAL, "fused stage: region must be 0, 1 or 2");
  if (a->shell_done && a->region != 0)
    return fail(ARTEMIS_HIP_EINVAL, "fused stage: shell_done requires region 0");
  if (p->nghost < 2) return fail(ARTEMIS_HIP_EUNSUPPORTED, "fused stage: needs nghost >= 2");
  if (a->outflow_faces_by_block && p->nblocks > 10)
    return fail(ARTEMIS_HIP_EUNSUPPORTED, "fused stage: outflow_faces_by_block serves packs of up to 10 blocks");
  const int recon = a->pcm ? ARTEMIS_PCM : p->gas.recon;
  const int rc = artemis::launch_stage_fused(artemis::make_pack_view(*p), *a, p->gas.riemann, recon,
                                             S(stream));
  if (rc == 5) return fail(ARTEMIS_HIP_EDEVICE, "fused stage: no memory for the redo lists");
  if (rc == 6) return fail(ARTEMIS_HIP_EUNSUPPORTED, "fused stage: mesh blocks of 2^29 zones or more (ghost zones included) are not addressed by the tile march");
  if (rc) return fail(ARTEMIS_HIP_EUNSUPPORTED, "fused stage: configuration not built (rc=%d)", rc);
  return after_launch("stage_fused");
}
int artemis_hip_stage_fused_redo_shell(const artemis_pack_t *p, const artemis_stage_args_t *a, void *stream) {
  if (int rc = validate(p)) return rc;
  if (!a || !a->prim_in || !a->prim_u1 || !a->prim_out) return fail(ARTEMIS_HIP_EINVAL, "fused stage (shell redo): the stage's own arguments are required");
  const int recon = a->pcm ? ARTEMIS_PCM : p->gas.recon;
  artemis::launch_stage_fused_redo_shell(artemis::make_pack_view(*p), *a, p->gas.riemann, recon, S(stream));
  return after_launch("stage_fused_redo_shell");
}

// Diffusion guards shared by the four tasks (see include/artemis_hip.h for what is built)
static int validate_diffusion(const artemis_pack_t *p, const artemis_diffusion_t *d, bool need_flux) {
  if (int rc = validate(p)) return rc;
  if (!d) return fail(ARTEMIS_HIP_EINVAL, "null diffusion parameters");
  if (p->gas.nspecies < 1) return fail(ARTEMIS_HIP_EINVAL, "diffusion only works with a gas fluid");
  if ((p->coords == ARTEMIS_CYLINDRICAL || p->coords == ARTEMIS_AXISYMMETRIC) && !p->metric)
    return fail(ARTEMIS_HIP_EINVAL, // Coords::Distance needs cos/sin of the azimuth
                "gas diffusion: cylindrical / axisymmetric blocks need the metric tables (artemis_hip_metric_fill)");
  if (p->nghost < 2) return fail(ARTEMIS_HIP_EINVAL, "gas diffusion needs nghost >= 2");
  if (!(d->cv > 0.0)) return fail(ARTEMIS_HIP_EINVAL, "diffusion: specific heat cv must be positive");
  for (const artemis_diffcoeff_t *c : {&d->visc, &d->cond}) {
    if (c->type == ARTEMIS_DIFF_OFF) continue;
    const bool is_visc = (c == &d->visc);
    if (is_visc ? (c->type != ARTEMIS_VISCOSITY_PLAW && c->type != ARTEMIS_VISCOSITY_ALPHA)
                : (c->type != ARTEMIS_CONDUCTIVITY_PLAW && c->type != ARTEMIS_THERMALDIFF_PLAW))
      return fail(ARTEMIS_HIP_EINVAL, is_visc ? "Invalid viscosity type" : "Invalid conductivity type");
    if (c->avg != 0 && c->avg != 1) return fail(ARTEMIS_HIP_EINVAL, "averaging is not supported");
    if (!is_visc && (c->temp_exp != 0.0 || c->rho_exp != 0.0) && !(c->T_ref > 0.0 && c->rho_ref > 0.0))
      return fail(ARTEMIS_HIP_EINVAL, "conductivity power laws: T_ref and rho_ref must be positive");
    if (is_visc && (c->type == ARTEMIS_VISCOSITY_ALPHA || c->r_exp != 0.0) && !c->radial)
      return fail(ARTEMIS_HIP_EINVAL,
                  "viscosity: alpha / radial power law need the radial table (artemis_hip_diffusion_radial_fill)");
    if (c->type == ARTEMIS_VISCOSITY_ALPHA && !(c->omega0 > 0.0 && c->r0 > 0.0))
      return fail(ARTEMIS_HIP_EINVAL, "alpha viscosity: r0 and omega0 = sqrt(gm / r0^3) must be positive");
  }
  if (need_flux)
    for (int dd = 0; dd < ((p->nx3 > 1) ? 3 : ((p->nx2 > 1) ? 2 : 1)); ++dd)
      if (!p->gas.diff_flux[dd]) return fail(ARTEMIS_HIP_EINVAL, "diffusion: gas.diff_flux tables are required");
  return 0;
}
int artemis_hip_diffusion_radial_fill(const artemis_pack_t *p, const double *geom_host,
                                      const double *metric_host, const artemis_diffcoeff_t *c,
                                      int block, double *out_host) {
  if (!p || !geom_host || !c || !out_host) return fail(ARTEMIS_HIP_EINVAL, "null argument");
  if (block < 0 || block >= p->nblocks) return fail(ARTEMIS_HIP_EINVAL, "bad block index %d", block);
  if (c->type != ARTEMIS_VISCOSITY_PLAW && c->type != ARTEMIS_VISCOSITY_ALPHA)
    return fail(ARTEMIS_HIP_EINVAL, "radial table: viscosity laws only");
  if (artemis_hip_metric_count(p) > 0 && !metric_host &&
      (p->coords == ARTEMIS_SPHERICAL2D || p->coords == ARTEMIS_SPHERICAL3D))
    return fail(ARTEMIS_HIP_EINVAL, "radial table: spherical 2-D/3-D blocks need the host metric table");
  const artemis::PackView P = artemis::make_pack_view(*p);
  const double *m = metric_host ? metric_host + block * artemis::metric_block_stride(P.nj, P.nk) : nullptr;
  // (std::pow is the cost of this table.  Its argument repeats along x2 in cylindrical coordinates -- both radii are
  //  functions of (x1, x3) there -- and along x3 / x2 elsewhere for one of the two laws: the row above is remembered,
  //  and a zone whose argument has the same bits takes its value: the same result, a twentieth of the calls)
  std::vector<double> prev_arg(P.ni), prev_val(P.ni);
  for (int k = 0; k < P.nk; ++k)
    for (int j = 0; j < P.nj; ++j) {
      if (p->coords == ARTEMIS_CYLINDRICAL && j > 0) { // (both radii are functions of (x1, x3): the row below, as it is)
        double *row = out_host + (static_cast<long>(k) * P.nj + j) * P.ni;
        std::memcpy(row, row - P.ni, sizeof(double) * P.ni);
        continue;
      }
      for (int i = 0; i < P.ni; ++i) {
        const artemis::DCoords co = artemis::coords_of(p->coords, geom_host + 6 * block, m, P.nj, P.nk, k, j, i);
        double xv[3];
        co.centre(xv);
        double arg;
        if (c->type == ARTEMIS_VISCOSITY_PLAW) {
          const artemis::Frame fr = artemis::cyl_frame(co.sys, xv, co.cv, co.sv);
          arg = fr.x[0] / c->r0;
        } else {
          arg = artemis::sph_radius(co.sys, xv) / c->r0;
        }
        double v;
        if ((j > 0 || k > 0) && std::memcmp(&arg, &prev_arg[i], sizeof arg) == 0) {
          v = prev_val[i];
        } else {
          v = (c->type == ARTEMIS_VISCOSITY_PLAW) ? std::pow(arg, c->r_exp) : c->omega0 * std::pow(arg, -1.5);
          prev_arg[i] = arg, prev_val[i] = v;
        }
        out_host[(static_cast<long>(k) * P.nj + j) * P.ni + i] = v;
      }
    }
  return 0;
}
size_t artemis_hip_viscous_distance_count(const artemis_pack_t *p) {
  if (validate(p)) return 0;
  return artemis::viscous_distance_count(artemis::make_pack_view(*p));
}
int artemis_hip_viscous_distance_fill(const artemis_pack_t *p, double *table_dev, void *stream) {
  if (int rc = validate(p)) return rc;
  if (!table_dev) return fail(ARTEMIS_HIP_EINVAL, "viscous distance fill: table_dev is NULL");
  if (artemis_hip_metric_count(p) > 0 && !p->metric)
    return fail(ARTEMIS_HIP_EINVAL, "viscous distance fill: this coordinate system needs p->metric (artemis_hip_metric_fill)");
  artemis::launch_viscous_distance_fill(artemis::make_pack_view(*p), table_dev, S(stream));
  return after_launch("viscous distance fill");
}
int artemis_hip_zero_diffusion_flux(const artemis_pack_t *p, void *stream) {
  if (int rc = validate(p)) return rc;
  for (int dd = 0; dd < ((p->nx3 > 1) ? 3 : ((p->nx2 > 1) ? 2 : 1)); ++dd)
    if (!p->gas.diff_flux[dd]) return fail(ARTEMIS_HIP_EINVAL, "diffusion: gas.diff_flux tables are required");
  artemis::launch_zero_diffusion_flux(artemis::make_pack_view(*p), S(stream));
  return after_launch("ZeroDiffusionFlux");
}
int artemis_hip_viscous_flux(const artemis_pack_t *p, const artemis_diffusion_t *d, void *stream) {
  if (int rc = validate_diffusion(p, d, true)) return rc;
  if (d->visc.type == ARTEMIS_DIFF_OFF) return 0; // gas.cpp:545-546
  if (artemis::launch_viscous_flux(artemis::make_pack_view(*p), *d, S(stream)))
    return fail(ARTEMIS_HIP_EDEVICE, "viscous flux: scratch allocation failed");
  return after_launch("ViscousFlux");
}
int artemis_hip_zero_viscous_flux(const artemis_pack_t *p, const artemis_diffusion_t *d, void *stream) {
  if (int rc = validate_diffusion(p, d, true)) return rc;
  if (d->visc.type == ARTEMIS_DIFF_OFF) return artemis_hip_zero_diffusion_flux(p, stream);
  if (artemis::launch_viscous_flux(artemis::make_pack_view(*p), *d, S(stream), true))
    return fail(ARTEMIS_HIP_EDEVICE, "viscous flux: scratch allocation failed");
  return after_launch("ZeroDiffusionFlux + ViscousFlux");
}
int artemis_hip_viscous_source_covers(const artemis_pack_t *p) {
  if (!p || p->nblocks <= 0 || p->nx1 <= 0) return 0;
  return artemis::viscous_source_covers(artemis::make_pack_view(*p)) ? 1 : 0;
}
int artemis_hip_viscous_source(const artemis_pack_t *p, const artemis_diffusion_t *d, double dt, const double *dt_dev,
                               double *const *sums, void *stream) {
  if (int rc = validate_diffusion(p, d, false)) return rc;
  if (!sums) return fail(ARTEMIS_HIP_EINVAL, "viscous source: null output table");
  if (!p->gas.prim) return fail(ARTEMIS_HIP_EINVAL, "viscous source: gas.prim table is required");
  if (d->visc.type == ARTEMIS_DIFF_OFF) return fail(ARTEMIS_HIP_EINVAL, "viscous source: viscosity is off");
  if (!d->dist) return fail(ARTEMIS_HIP_EINVAL, "viscous source: the distance table is required (artemis_hip_viscous_distance_fill)");
  const artemis::PackView P = artemis::make_pack_view(*p);
  if (!artemis::viscous_source_covers(P))
    return fail(ARTEMIS_HIP_EUNSUPPORTED, "viscous source: 3-D blocks of one gas species, at least 8 x 8 zones wide (use the flux tasks)");
  artemis::launch_viscous_source(P, *d, dt, dt_dev, sums, S(stream));
  return after_launch("viscous source");
}
int artemis_hip_ml_viscous_faces(const artemis_pack_t *p, const artemis_diffusion_t *d, const artemis_ml_face_box_t *boxes_dev,
                                 int nboxes, const artemis_ml_fix_cell_t *cells_dev, int ncells, void *stream) {
  if (int rc = validate_diffusion(p, d, true)) return rc;
  if (p->gas.nspecies != 1) return fail(ARTEMIS_HIP_EUNSUPPORTED, "listed viscous faces: one gas species");
  if (d->visc.type == ARTEMIS_DIFF_OFF) return fail(ARTEMIS_HIP_EINVAL, "listed viscous faces: viscosity is off");
  if (nboxes < 0 || ncells < 0 || (nboxes > 0 && !boxes_dev) || (ncells > 0 && !cells_dev))
    return fail(ARTEMIS_HIP_EINVAL, "listed viscous faces: bad list");
  if (!p->gas.prim) return fail(ARTEMIS_HIP_EINVAL, "listed viscous faces: gas.prim table is required");
  artemis::launch_viscous_listed_faces(artemis::make_pack_view(*p), *d, boxes_dev, nboxes, cells_dev, ncells, S(stream));
  return after_launch("listed viscous faces");
}
int artemis_hip_thermal_flux(const artemis_pack_t *p, const artemis_diffusion_t *d, void *stream) {
  if (int rc = validate_diffusion(p, d, true)) return rc;
  if (d->cond.type == ARTEMIS_DIFF_OFF) return 0; // gas.cpp:582-583
  artemis::launch_thermal_flux(artemis::make_pack_view(*p), *d, S(stream));
  return after_launch("ThermalFlux");
}
int artemis_hip_diffusion_update(const artemis_pack_t *p, const artemis_diffusion_t *d, double dt,
                                 void *stream) {
  if (int rc = validate_diffusion(p, d, true)) return rc;
  artemis::launch_diffusion_update(artemis::make_pack_view(*p), *d, dt, S(stream));
  return after_launch("DiffusionUpdate");
}
int artemis_hip_diffusion_dt(const artemis_pack_t *p, const artemis_diffusion_t *d, double cfl,
                             double *dt_dev, void *stream) {
  if (int rc = validate_diffusion(p, d, false)) return rc;
  if (!dt_dev) return fail(ARTEMIS_HIP_EINVAL, "null dt_dev");
  artemis::launch_diffusion_dt(artemis::make_pack_view(*p), *d, cfl, dt_dev, S(stream));
  return after_launch("Diffusion::EstimateTimestep");
}

int artemis_hip_timestep_all(const artemis_pack_t *p, double cfl_gas, double cfl_dust, const artemis_diffusion_t *d,
                             double *dt_dev, void *stream) {
  if (int rc = validate(p)) return rc;
  if (d)
    if (int rc = validate_diffusion(p, d, false)) return rc;
  if (!dt_dev) return fail(ARTEMIS_HIP_EINVAL, "null dt_dev");
  if (p->gas.nspecies == 0 && p->dust.nspecies == 0) return 0;
  artemis::launch_timestep_all(artemis::make_pack_view(*p), d, cfl_gas, cfl_dust, dt_dev, S(stream));
  return after_launch("EstimateTimestepMesh (all limits)");
}

int artemis_hip_stage_general(const artemis_pack_t *p, const artemis_stage_general_args_t *a,
                              void *stream) {
  if (int rc = validate(p)) return rc;
  if (!a) return fail(ARTEMIS_HIP_EINVAL, "null stage args");
  if (int rc = validate_fluid(p, ARTEMIS_GAS, a->pcm)) return rc;
  if (int rc = validate_fluid(p, ARTEMIS_DUST, a->pcm)) return rc;
  if (p->gas.nspecies && (!a->gas_in || !a->gas_u1 || !a->gas_out))
    return fail(ARTEMIS_HIP_EINVAL, "general stage: gas_in / gas_u1 / gas_out are required");
  if (p->dust.nspecies && (!a->dust_in || !a->dust_u1 || !a->dust_out))
    return fail(ARTEMIS_HIP_EINVAL, "general stage: dust_in / dust_u1 / dust_out are required");
  if ((p->gas.nspecies && a->gas_in == a->gas_out) || (p->dust.nspecies && a->dust_in == a->dust_out))
    return fail(ARTEMIS_HIP_EINVAL, "general stage: *_out must not alias *_in");
  if (a->gravity) { // same guards as artemis_hip_external_gravity
    const artemis_gravity_t *g = a->gravity;
    if (g->type != ARTEMIS_GRAVITY_UNIFORM && g->type != ARTEMIS_GRAVITY_POINT && g->type != ARTEMIS_GRAVITY_BINARY)
      return fail(ARTEMIS_HIP_EUNSUPPORTED, "gravity type %d (nbody) is not built", g->type);
    if (g->type == ARTEMIS_GRAVITY_BINARY &&
        (p->coords == ARTEMIS_AXISYMMETRIC || p->coords == ARTEMIS_SPHERICAL1D || p->coords == ARTEMIS_SPHERICAL2D))
      return fail(ARTEMIS_HIP_EINVAL, "Binary gravity is not compatable with axisymmetric coordinates!");
    if (g->type != ARTEMIS_GRAVITY_UNIFORM && p->coords == ARTEMIS_CYLINDRICAL && !p->metric)
      return fail(ARTEMIS_HIP_EINVAL, "point-mass gravity on cylindrical blocks needs the metric tables");
  }
  if (a->rf_omega != 0.0 && p->coords != ARTEMIS_CARTESIAN && a->rf_qshear != 0.0) // rotating_frame.cpp:34-38
    return fail(ARTEMIS_HIP_EINVAL, "rotating_frame/qshear must be zero for non-Cartesian coordinate systems!");
  if (int rc = validate_stage_nbody(p, a)) return rc;
  if (a->diffusion) {
    if (int rc = validate_diffusion(p, a->diffusion, a->diffusion_sums == nullptr)) return rc;
    if (a->diffusion_sums && p->gas.nspecies != 1)
      return fail(ARTEMIS_HIP_EINVAL, "general stage: diffusion_sums are for one gas species");
  }
  if (a->cooling) {
    if (a->drag) return fail(ARTEMIS_HIP_EUNSUPPORTED, "general stage: cooling together with drag runs on the per-task kernels");
    if (p->gas.nspecies && (!a->cooling->tref || !a->cooling->beta))
      return fail(ARTEMIS_HIP_EINVAL, "cooling: tref / beta tables are required (artemis_hip_cooling_table_fill)");
    if (!(a->cooling->cv > 0.0)) return fail(ARTEMIS_HIP_EINVAL, "cooling: specific heat cv must be positive");
  }
  if (a->drag) {
    if (int rc = validate_damp_visc(a->drag)) return rc;
    if (a->drag->type == ARTEMIS_DRAG_SIMPLE_DUST && (p->gas.nspecies < 1 || p->dust.nspecies < 1))
      return fail(ARTEMIS_HIP_EINVAL, "drag type simple_dust requires do_gas = do_dust = true");
    if (p->dust.nspecies > ARTEMIS_MAX_DUST_SPECIES)
      return fail(ARTEMIS_HIP_EUNSUPPORTED, "drag: more than %d dust species", ARTEMIS_MAX_DUST_SPECIES);
    if ((p->gas.nspecies && !p->gas.cons0) || (p->dust.nspecies && !p->dust.cons0))
      return fail(ARTEMIS_HIP_EINVAL, "general stage with drag: cons0 tables are required as scratch");
  }
  if (a->defer_finish && ((p->gas.nspecies && !p->gas.cons0) || (p->dust.nspecies && !p->dust.cons0)))
    return fail(ARTEMIS_HIP_EINVAL, "general stage, defer_finish: cons0 tables are required");
  if (a->defer_finish && a->cooling)
    return fail(ARTEMIS_HIP_EUNSUPPORTED, "general stage: defer_finish with cooling (it follows drag in the task list)");
  if (a->strat_faces && artemis::stage_general_variant(artemis::make_pack_view(*p), *a, p->gas.recon, p->gas.riemann, p->dust.recon,
                                                       p->dust.riemann) != 1)
    return fail(ARTEMIS_HIP_EUNSUPPORTED, "general stage: strat_faces (conditions inside the kernel) is served by the 2-D row "
                                          "march only, for all four faces of one block (strat_faces = 15)");
  artemis::launch_stage_cell(artemis::make_pack_view(*p), *a, p->gas.recon, p->gas.riemann,
                             p->dust.recon, p->dust.riemann, S(stream));
  return after_launch("stage_general");
}

int artemis_hip_stage_general_variant(const artemis_pack_t *p, const artemis_stage_general_args_t *a) {
  if (!p || !a) return 0;
  return artemis::stage_general_variant(artemis::make_pack_view(*p), *a, p->gas.recon, p->gas.riemann, p->dust.recon,
                                        p->dust.riemann);
}

static int stage_epilogue_common(const artemis_pack_t *p, const artemis_stage_general_args_t *a, void *stream, bool to_cons) {
  if (int rc = validate(p)) return rc;
  if (!a) return fail(ARTEMIS_HIP_EINVAL, "null stage args");
  if (a->drag) return fail(ARTEMIS_HIP_EUNSUPPORTED, "stage epilogue: drag couples the fluids; use the separate tasks");
  const int ndim = (p->nx3 > 1) ? 3 : ((p->nx2 > 1) ? 2 : 1);
  for (const artemis_fluid_pack_t *f : {&p->gas, &p->dust}) {
    if (!f->nspecies) continue;
    if (!f->prim || !f->cons0 || !f->cons1) return fail(ARTEMIS_HIP_EINVAL, "stage epilogue: prim / cons0 / cons1 tables are required");
    for (int d = 0; d < ndim; ++d)
      if (!f->flux[d] || (f == &p->gas && (!f->pflux[d] || !f->vface[d])))
        return fail(ARTEMIS_HIP_EINVAL, "stage epilogue: flux / pflux / vface tables are required");
  }
  if (a->gravity) {
    const artemis_gravity_t *g = a->gravity;
    if (g->type != ARTEMIS_GRAVITY_UNIFORM && g->type != ARTEMIS_GRAVITY_POINT && g->type != ARTEMIS_GRAVITY_BINARY)
      return fail(ARTEMIS_HIP_EUNSUPPORTED, "gravity type %d (nbody) is not built", g->type);
    if (g->type == ARTEMIS_GRAVITY_BINARY &&
        (p->coords == ARTEMIS_AXISYMMETRIC || p->coords == ARTEMIS_SPHERICAL1D || p->coords == ARTEMIS_SPHERICAL2D))
      return fail(ARTEMIS_HIP_EINVAL, "Binary gravity is not compatable with axisymmetric coordinates!");
    if (g->type != ARTEMIS_GRAVITY_UNIFORM && p->coords == ARTEMIS_CYLINDRICAL && !p->metric)
      return fail(ARTEMIS_HIP_EINVAL, "point-mass gravity on cylindrical blocks needs the metric tables");
  }
  if (a->rf_omega != 0.0 && p->coords != ARTEMIS_CARTESIAN && a->rf_qshear != 0.0)
    return fail(ARTEMIS_HIP_EINVAL, "rotating_frame/qshear must be zero for non-Cartesian coordinate systems!");
  if (a->diffusion)
    if (int rc = validate_diffusion(p, a->diffusion, true)) return rc;
  if (a->cooling) {
    if (p->gas.nspecies && (!a->cooling->tref || !a->cooling->beta))
      return fail(ARTEMIS_HIP_EINVAL, "cooling: tref / beta tables are required (artemis_hip_cooling_table_fill)");
    if (!(a->cooling->cv > 0.0)) return fail(ARTEMIS_HIP_EINVAL, "cooling: specific heat cv must be positive");
  }
  if (to_cons && a->cooling) return fail(ARTEMIS_HIP_EUNSUPPORTED, "stage epilogue (cons): cooling acts after drag; use the separate tasks");
  artemis::launch_stage_epilogue(artemis::make_pack_view(*p), *a, S(stream), to_cons);
  return after_launch(to_cons ? "stage_epilogue_cons" : "stage_epilogue");
}
int artemis_hip_stage_epilogue(const artemis_pack_t *p, const artemis_stage_general_args_t *a, void *stream) {
  return stage_epilogue_common(p, a, stream, false);
}
int artemis_hip_stage_epilogue_cons(const artemis_pack_t *p, const artemis_stage_general_args_t *a, void *stream) {
  return stage_epilogue_common(p, a, stream, true);
}
int artemis_hip_stage_finish_cells(const artemis_pack_t *p, const artemis_drag_t *drag, double time, double dt,
                                   const artemis_ml_fix_cell_t *cells_dev, int ncells, void *stream) {
  (void)time;
  if (int rc = validate(p)) return rc;
  // (cons0 alone: the three tasks read and write the current conserved state; the start-of-step copy plays no part)
  if (ncells < 0 || (ncells > 0 && !cells_dev)) return fail(ARTEMIS_HIP_EINVAL, "stage finish: bad zone list");
  for (const artemis_fluid_pack_t *f : {&p->gas, &p->dust})
    if (f->nspecies && (!f->prim || !f->cons0)) return fail(ARTEMIS_HIP_EINVAL, "stage finish: prim and cons0 tables are required");
  if (!drag || drag->type != ARTEMIS_DRAG_SIMPLE_DUST || p->gas.nspecies != 1 || p->dust.nspecies < 1)
    return fail(ARTEMIS_HIP_EUNSUPPORTED, "stage finish of listed zones: one gas species coupled by simple_dust drag");
  if (int rc = validate_damp_visc(drag)) return rc;
  if (p->dust.nspecies > ARTEMIS_MAX_DUST_SPECIES)
    return fail(ARTEMIS_HIP_EUNSUPPORTED, "drag: more than %d dust species", ARTEMIS_MAX_DUST_SPECIES);
  if (ncells == 0) return ARTEMIS_HIP_OK;
  artemis::launch_drag_finish_cells(artemis::make_pack_view(*p), *drag, dt, cells_dev, ncells, S(stream));
  return after_launch("stage_finish_cells");
}
int artemis_hip_stage_finish(const artemis_pack_t *p, const artemis_drag_t *drag, double time, double dt, void *stream) {
  (void)time;
  if (int rc = validate(p)) return rc;
  for (const artemis_fluid_pack_t *f : {&p->gas, &p->dust}) // (cons0 alone: the start-of-step copy plays no part)
    if (f->nspecies && (!f->prim || !f->cons0)) return fail(ARTEMIS_HIP_EINVAL, "stage finish: prim and cons0 tables are required");
  const artemis::PackView P = artemis::make_pack_view(*p);
  if (drag) {
    if (int rc = validate_damp_visc(drag)) return rc;
    if (drag->type == ARTEMIS_DRAG_SIMPLE_DUST && (p->gas.nspecies < 1 || p->dust.nspecies < 1))
      return fail(ARTEMIS_HIP_EINVAL, "drag type simple_dust requires do_gas = do_dust = true");
    if (p->dust.nspecies > ARTEMIS_MAX_DUST_SPECIES)
      return fail(ARTEMIS_HIP_EUNSUPPORTED, "drag: more than %d dust species", ARTEMIS_MAX_DUST_SPECIES);
    if (artemis::launch_drag_finish(P, *drag, dt, nullptr, S(stream))) return after_launch("stage_finish");
    artemis::launch_drag_source(P, *drag, dt, nullptr, S(stream));
  }
  if (P.gas.ns) artemis::launch_set_aux(P, S(stream));
  artemis::launch_cons_to_prim(P, S(stream));
  return after_launch("stage_finish");
}


static int validate_refine(const artemis_refine_t *r, bool prolongate) {
  if (!r) return fail(ARTEMIS_HIP_EINVAL, "null refinement descriptor");
  if (r->coords < ARTEMIS_CARTESIAN || r->coords > ARTEMIS_AXISYMMETRIC) return fail(ARTEMIS_HIP_EINVAL, "Coordinate type not recognized!");
  if (r->ndim < 1 || r->ndim > 3 || r->nvar < 1) return fail(ARTEMIS_HIP_EINVAL, "bad ndim / nvar");
  if (!r->fgeom || !r->cgeom || !r->fine || !r->coarse) return fail(ARTEMIS_HIP_EINVAL, "null table");
  if ((r->coords == ARTEMIS_SPHERICAL2D || r->coords == ARTEMIS_SPHERICAL3D) && (!r->fmetric || !r->cmetric))
    return fail(ARTEMIS_HIP_EINVAL, "spherical 2-D/3-D needs the metric tables of both index spaces");
  const int g = prolongate ? 1 : 0; // prolongation reads the coarse neighbours
  const int clo[3] = {r->cis, r->cjs, r->cks}, chi[3] = {r->cie, r->cje, r->cke}, cn[3] = {r->cni, r->cnj, r->cnk};
  const int cb[3] = {r->cib, r->cjb, r->ckb}, fb[3] = {r->fib, r->fjb, r->fkb}, fn[3] = {r->fni, r->fnj, r->fnk};
  for (int d = 0; d < 3; ++d) {
    const int act = d < r->ndim;
    if (clo[d] > chi[d] || clo[d] - g * act < 0 || chi[d] + g * act >= cn[d])
      return fail(ARTEMIS_HIP_EINVAL, "coarse range out of bounds in direction %d", d + 1);
    const int f0 = act ? (clo[d] - cb[d]) * 2 + fb[d] : fb[d], f1 = act ? (chi[d] - cb[d]) * 2 + fb[d] + 1 : fb[d];
    if (f0 < 0 || f1 >= fn[d]) return fail(ARTEMIS_HIP_EINVAL, "fine range out of bounds in direction %d", d + 1);
    if (!act && clo[d] != chi[d]) return fail(ARTEMIS_HIP_EINVAL, "inactive direction %d must be one zone", d + 1);
  }
  return device_ready();
}
int artemis_hip_restrict_average(const artemis_refine_t *r, void *stream) {
  if (int rc = validate_refine(r, false)) return rc;
  artemis::launch_refine(*r, 0, S(stream));
  return after_launch("RestrictAverage");
}
int artemis_hip_prolongate_minmod(const artemis_refine_t *r, void *stream) {
  if (int rc = validate_refine(r, true)) return rc;
  artemis::launch_refine(*r, 1, S(stream));
  return after_launch("ProlongateSharedMinMod");
}


static int amr_criterion(const artemis_amr_criterion_t *a, int magnitude, int *tag, double *maxval, void *stream) {
  if (!a || !tag) return fail(ARTEMIS_HIP_EINVAL, "null refinement criterion / tag");
  if (a->coords < ARTEMIS_CARTESIAN || a->coords > ARTEMIS_AXISYMMETRIC) return fail(ARTEMIS_HIP_EINVAL, "Coordinate type not recognized!");
  if (a->ndim < 1 || a->ndim > 3) return fail(ARTEMIS_HIP_EINVAL, "bad ndim");
  if (!a->geom || !a->field || !a->scratch) return fail(ARTEMIS_HIP_EINVAL, "null table");
  if ((a->coords == ARTEMIS_SPHERICAL2D || a->coords == ARTEMIS_SPHERICAL3D) && !a->metric && !magnitude)
    return fail(ARTEMIS_HIP_EINVAL, "spherical 2-D/3-D needs the metric table");
  *tag = 0;
  if (maxval) *maxval = 0.0;
  if (!magnitude && a->ndim == 1) return ARTEMIS_HIP_OK; // amr_criteria.hpp:122-124: AmrTag::same
  const int lo[3] = {a->is, a->js, a->ks}, hi[3] = {a->ie, a->je, a->ke}, n[3] = {a->ni, a->nj, a->nk};
  for (int d = 0; d < 3; ++d) {
    const int g = (!magnitude && d < a->ndim) ? 2 : 0; // the derivative of the grown range reads +-2
    if (lo[d] > hi[d] || lo[d] - g < 0 || hi[d] + g >= n[d])
      return fail(ARTEMIS_HIP_EINVAL, "refinement criterion: range out of bounds in direction %d", d + 1);
  }
  if (int rc = device_ready()) return rc;
  artemis::launch_amr_criterion(*a, magnitude, S(stream));
  double m = 0.0;
  if (hipMemcpyAsync(&m, a->scratch, sizeof(double), hipMemcpyDeviceToHost, S(stream)) != hipSuccess ||
      hipStreamSynchronize(S(stream)) != hipSuccess)
    return fail(ARTEMIS_HIP_EDEVICE, "refinement criterion: %s", hipGetErrorString(hipGetLastError()));
  if (maxval) *maxval = m;
  if (magnitude) *tag = (m > a->refine_thr) ? 1 : ((m < a->deref_thr) ? -1 : 0);       // :164-166
  else *tag = (m > a->refine_thr) ? 1 : ((m < 0.25 * a->refine_thr) ? -1 : 0);          // :126-130
  return after_launch(magnitude ? "ScalarMagnitude" : "ScalarFirstDerivative");
}
int artemis_hip_amr_first_derivative(const artemis_amr_criterion_t *a, int *tag, double *maxval, void *stream) {
  return amr_criterion(a, 0, tag, maxval, stream);
}
int artemis_hip_amr_magnitude(const artemis_amr_criterion_t *a, int *tag, double *maxval, void *stream) {
  return amr_criterion(a, 1, tag, maxval, stream);
}

int artemis_hip_amr_block_maxima(const artemis_pack_t *p, int field, int magnitude, double *maxima_dev, void *stream) {
  if (int rc = validate(p)) return rc;
  if (!maxima_dev || field < 0 || field > 2 || !p->gas.nspecies || !p->gas.prim)
    return fail(ARTEMIS_HIP_EINVAL, "amr_block_maxima: field must be 0 (density), 1 (pressure) or 2 (pressure from rho, sie) of a gas pack, "
                                    "maxima_dev non-null");
  if (!magnitude && p->nghost < 2) return fail(ARTEMIS_HIP_EINVAL, "ScalarFirstDerivative needs two ghost zones");
  const artemis::PackView P = artemis::make_pack_view(*p);
  if (field == 2) artemis::launch_pack_criterion(P, 0, 5 * p->gas.nspecies, magnitude, maxima_dev, S(stream));
  else artemis::launch_pack_criterion(P, field == 0 ? 0 : 4 * p->gas.nspecies, -1, magnitude, maxima_dev, S(stream));
  return after_launch("refinement criterion (pack)");
}

int artemis_hip_advance_dt(double *state, double tlim, int nstages, const double *beta, void *stream) {
  if (int rc = device_ready()) return rc;
  if (!state || !beta || nstages < 1 || nstages > 3) return fail(ARTEMIS_HIP_EINVAL, "bad advance_dt arguments");
  artemis::launch_advance_dt(state, tlim, nstages, beta, S(stream));
  return after_launch("advance_dt");
}

int artemis_hip_wait_counter(unsigned *counter, unsigned target, unsigned *timeout_flag, void *stream) {
  if (int rc = device_ready()) return rc;
  if (!counter) return fail(ARTEMIS_HIP_EINVAL, "null counter");
  artemis::launch_wait_counter(counter, target, timeout_flag, S(stream));
  return after_launch("wait_counter");
}

// ---- runtime shim (include/artemis_rt.h) ------------------------------------------------
int artemis_rt_set_device(int dev) {
  if (int rc = device_ready()) return rc;
  return check_hip(hipSetDevice(dev), "hipSetDevice");
}
// device bytes handed out through this shim: current and high-water mark (artemis_rt_device_bytes)
namespace {
std::mutex g_bytes_mu;
size_t g_bytes_now = 0, g_bytes_peak = 0;
} // namespace
void artemis_rt_device_bytes(size_t *current, size_t *peak, int reset_peak) {
  std::lock_guard<std::mutex> lk(g_bytes_mu);
  if (current) *current = g_bytes_now;
  if (peak) *peak = g_bytes_peak;
  if (reset_peak) g_bytes_peak = g_bytes_now;
}
// A remesh frees tens of GB and allocates about as much again in buffers whose sizes differ by a fraction of a percent
// (a handful of blocks more or fewer): through hipFree / hipMalloc that unmaps and re-maps the lot (1.3 - 1.5 s per
// remesh on the 29 M-zone configs[4] mesh, twenty cycle-times).  With the cache enabled (artemis_rt_pool_limit: the
// standalone driver turns it on for adaptive meshes; a library host such as the Parthenon adapter never pays for it
// unless it asks) freed buffers are kept instead -- per device, sizes rounded up to 1/16 of their power of two, so that
// consecutive states land in the same size class -- and handed out again; the cache is trimmed to the limit and
// emptied when the device runs out.
namespace {
struct PoolEntry {
  void *p;
  unsigned long seq; // when it was freed: the cache is trimmed oldest first
  int dev;           // the device the buffer lives on: only handed out while that device is current
};
struct LiveEntry {
  size_t cap;
  int dev;
};
std::multimap<size_t, PoolEntry> g_pool; // capacity -> free buffer
std::unordered_map<void *, LiveEntry> g_live;
size_t g_pool_bytes = 0, g_pool_limit = 0; // limit 0: the cache is off (plain hipMalloc / hipFree)
unsigned long g_pool_seq = 0;
size_t size_class(size_t bytes) {
  if (bytes < (size_t(1) << 16)) return (bytes + 255) & ~size_t(255);
  size_t p2 = 1;
  while ((p2 << 1) <= bytes) p2 <<= 1;
  const size_t g = p2 >> 4;
  return (bytes + g - 1) / g * g;
}
void pool_trim_locked(size_t keep) { // least recently freed first (what earlier, smaller meshes left behind goes first)
  while (g_pool_bytes > keep && !g_pool.empty()) {
    auto it = g_pool.begin();
    for (auto q = g_pool.begin(); q != g_pool.end(); ++q)
      if (q->second.seq < it->second.seq) it = q;
    g_pool_bytes -= it->first, g_bytes_now -= it->first;
    (void)hipFree(it->second.p);
    g_pool.erase(it);
  }
}
} // namespace
void artemis_rt_pool_limit(size_t limit_bytes) {
  std::lock_guard<std::mutex> lk(g_bytes_mu);
  g_pool_limit = limit_bytes;
  pool_trim_locked(limit_bytes);
}
void *artemis_rt_malloc(size_t bytes) {
  if (device_ready()) return nullptr;
  void *p = nullptr;
  int dev = 0;
  (void)hipGetDevice(&dev);
  bool pool;
  {
    std::lock_guard<std::mutex> lk(g_bytes_mu);
    pool = g_pool_limit > 0;
  }
  // a cached buffer serves if it fits (up to half as large again); a fresh one gets 3 % of headroom before its size class is
  // taken, so that a mesh that keeps growing by a few blocks per remesh does not cross a class boundary -- a fresh
  // hipMalloc of GBs -- right after the buffers were made
  const size_t need = pool ? size_class(bytes ? bytes : 8) : (bytes ? bytes : 8);
  // (headroom of a fresh buffer: 3 %, or 20 % from 64 MB up -- the per-field slabs of an adaptive mesh, which grow by a
  //  few per cent per remesh while a feature is being refined.  Every slab crosses its class in the same remesh, and
  //  mapping 70 GB afresh costs 0.4 s: with 6 % that was every second or third such remesh, with 12 % every fifth, with
  //  20 % every seventh or eighth; what a shrinking mesh leaves behind goes back at the trim after each remesh)
  const size_t cap = pool ? size_class((bytes ? bytes : 8) + (bytes >= (size_t(64) << 20) ? bytes / 5 : bytes / 32)) : need;
  if (pool) {
    std::lock_guard<std::mutex> lk(g_bytes_mu);
    for (auto it = g_pool.lower_bound(need); it != g_pool.end() && it->first <= need + need / 2; ++it) { // (covers the headroom below)
      if (it->second.dev != dev) continue; // (a buffer of another device is not this device's memory)
      p = it->second.p;
      g_pool_bytes -= it->first;
      g_live[p] = LiveEntry{it->first, dev};
      g_pool.erase(it);
      break;
    }
  }
  if (!p) {
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess) { // out of memory with buffers cached: give them back and try again
      (void)hipGetLastError();
      {
        std::lock_guard<std::mutex> lk(g_bytes_mu);
        pool_trim_locked(0);
      }
      e = hipMalloc(&p, cap);
    }
    if (check_hip(e, "hipMalloc")) return nullptr;
    std::lock_guard<std::mutex> lk(g_bytes_mu);
    g_live[p] = LiveEntry{cap, dev};
    g_bytes_now += cap; // (the device footprint: live buffers and cached ones, at their capacities)
    if (g_bytes_now > g_bytes_peak) g_bytes_peak = g_bytes_now;
  }
  // ARTEMIS_POISON=1 (debugging aid): fresh device memory holds NaN patterns, so that a read of something never
  // written shows up as NaN instead of depending on what the allocator handed back
  const bool poison = artemis::opt(artemis::OPT_POISON) != 0;
  if (poison && bytes) {
    (void)hipMemset(p, 0xFF, bytes);
    (void)hipDeviceSynchronize();
  }
  return p;
}
void artemis_rt_free(void *p) {
  if (!p) return;
  LiveEntry le{0, 0};
  bool pool;
  {
    std::lock_guard<std::mutex> lk(g_bytes_mu);
    auto it = g_live.find(p);
    if (it != g_live.end()) le = it->second, g_live.erase(it);
    pool = g_pool_limit > 0 && le.cap > 0 && le.cap <= g_pool_limit;
    if (!pool && le.cap) g_bytes_now -= le.cap;
  }
  if (!pool) {
    (void)hipFree(p); // (synchronises by itself)
    return;
  }
  (void)hipDeviceSynchronize(); // what hipFree guarantees: nothing in flight still uses the buffer
  std::lock_guard<std::mutex> lk(g_bytes_mu);
  g_pool.emplace(le.cap, PoolEntry{p, ++g_pool_seq, le.dev});
  g_pool_bytes += le.cap;
  pool_trim_locked(g_pool_limit);
}
size_t artemis_rt_pool_bytes(void) {
  std::lock_guard<std::mutex> lk(g_bytes_mu);
  return g_pool_bytes;
}
void artemis_rt_pool_trim(size_t keep_bytes) {
  std::lock_guard<std::mutex> lk(g_bytes_mu);
  pool_trim_locked(keep_bytes);
}
void *artemis_rt_malloc_host(size_t bytes) {
  if (device_ready()) return nullptr;
  void *p = nullptr;
  if (check_hip(hipHostMalloc(&p, bytes ? bytes : 8), "hipHostMalloc")) return nullptr;
  return p;
}
void artemis_rt_free_host(void *p) {
  if (p) (void)hipHostFree(p);
}
int artemis_rt_memcpy_h2d(void *dst, const void *src, size_t n, void *stream) {
  return check_hip(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, S(stream)), "memcpy h2d");
}
int artemis_rt_memcpy_d2h(void *dst, const void *src, size_t n, void *stream) {
  return check_hip(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, S(stream)), "memcpy d2h");
}
int artemis_rt_memcpy_d2d(void *dst, const void *src, size_t n, void *stream) {
  return check_hip(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, S(stream)), "memcpy d2d");
}
int artemis_rt_memset(void *dst, int value, size_t n, void *stream) {
  return check_hip(hipMemsetAsync(dst, value, n, S(stream)), "memset");
}
void *artemis_rt_stream_create(void) {
  if (device_ready()) return nullptr;
  hipStream_t s = nullptr;
  if (check_hip(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate")) return nullptr;
  return s;
}
void artemis_rt_stream_destroy(void *s) {
  if (s) (void)hipStreamDestroy(S(s));
}
int artemis_rt_stream_sync(void *s) { return check_hip(hipStreamSynchronize(S(s)), "stream sync"); }
int artemis_rt_device_sync(void) { return check_hip(hipDeviceSynchronize(), "device sync"); }
void *artemis_rt_event_create(void) {
  if (device_ready()) return nullptr;
  hipEvent_t e = nullptr;
  if (check_hip(hipEventCreate(&e), "hipEventCreate")) return nullptr;
  return e;
}
void artemis_rt_event_destroy(void *e) {
  if (e) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
}
int artemis_rt_event_record(void *e, void *stream) {
  return check_hip(hipEventRecord(static_cast<hipEvent_t>(e), S(stream)), "event record");
}
int artemis_rt_stream_wait_event(void *stream, void *e) {
  return check_hip(hipStreamWaitEvent(S(stream), static_cast<hipEvent_t>(e), 0), "stream wait event");
}
int artemis_rt_event_sync(void *e) {
  return check_hip(hipEventSynchronize(static_cast<hipEvent_t>(e)), "event sync");
}
double artemis_rt_event_elapsed_ms(void *e0, void *e1) {
  float ms = -1.0f;
  if (hipEventElapsedTime(&ms, static_cast<hipEvent_t>(e0), static_cast<hipEvent_t>(e1)) != hipSuccess)
    return -1.0;
  return ms;
}
void artemis_rt_tables_changed(void) { artemis::invalidate_table_cache(); }
int artemis_rt_capture_begin(void *stream) {
  return check_hip(hipStreamBeginCapture(S(stream), hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
}
void *artemis_rt_capture_end(void *stream) {
  hipGraph_t g = nullptr;
  if (check_hip(hipStreamEndCapture(S(stream), &g), "hipStreamEndCapture") || !g) return nullptr;
  hipGraphExec_t e = nullptr;
  const hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (check_hip(rc, "hipGraphInstantiate")) return nullptr;
  return e;
}
int artemis_rt_graph_launch(void *graph_exec, void *stream) {
  return check_hip(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), S(stream)), "hipGraphLaunch");
}
void artemis_rt_graph_destroy(void *graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec));
}

} // extern "C"
