/* =======================================================================================
 * artemis_hip.h -- C ABI of the MI355X-native finite-volume hydro update for Artemis.
 *
 * This is the drop-in boundary: every entry point replaces one Parthenon task / package
 * callback of lanl/artemis (reference @ 2025-01-17; file:line cited per function, paths
 * relative to the reference's src/).  Plain pointers and sizes only -- no Kokkos, no torch,
 * no C++ types.  A host adapter fills `artemis_pack_t` from a MeshData's SparsePacks (see
 * INTEGRATION.md) and forwards the call; the repo's own host driver (artemis_driver.h)
 * fills the same struct when Parthenon is absent.
 *
 * Conventions
 *  - All arrays are fp64 (`Real` = double in the reference).
 *  - A cell array of one variable of one block is contiguous [nk][nj][ni], i fastest, with
 *    ni = nx1 + 2*nghost, nj = nx2 + 2*nghost if nx2 > 1 else 1, nk likewise (Parthenon's
 *    layout; ndim = 3 if nx3 > 1, else 2 if nx2 > 1, else 1).
 *  - Face-centred quantities (fluxes, interface pressure, face velocity) of direction d are
 *    stored at the cell index of the cell whose LOWER d-face they live on
 *    (fluid_fluxes.hpp:119-121 writes faces [is, ie+1] at cell indices [is, ie+1]).
 *  - Pointer tables (`double *const *`) are DEVICE arrays of DEVICE pointers, indexed
 *    [block * nvar + var] -- exactly what a SparsePack holds.  Variable order inside a table
 *    follows the reference's pack order (hllc.hpp:66-73, hlle.hpp:78-86, gas.cpp:479-486):
 *      gas  prim : rho[n], vel[ns + 3n + d], P[4ns + n], sie[5ns + n]            (6*ns)
 *      gas  cons : D[n],   mom[ns + 3n + d], E[4ns + n], eint[5ns + n]           (6*ns)
 *      dust prim : rho[n], vel[ns + 3n + d]                                      (4*ns)
 *      dust cons : D[n],   mom[ns + 3n + d]                                      (4*ns)
 *    flux[d] tables use the cons order; pflux[d] / vface[d] have one entry per gas species
 *    (the flux slot of gas.prim.pressure, hllc.hpp:166, and gas.face.velocity, hllc.hpp:179).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls enqueue
 *    work and return; only the functions documented as synchronous wait for the device.
 *  - Return value 0 = TaskStatus::complete.  Non-zero = error code; the message is
 *    available from artemis_hip_last_error() (thread-local).  Nothing throws across the ABI.
 *    PARTHENON_FAIL conditions of the reference (unknown solver / reconstruction / coordinate
 *    system, fluid_fluxes.hpp:235,258,290; too few ghost cells, gas.cpp:61-76) map to
 *    ARTEMIS_HIP_EINVAL.
 *
 * Switches (debugging / measurement aids of the library and of the host driver -- results do not depend on them unless
 * stated, and none is needed in production):
 *  They live in ONE table (artemis_amd/csrc/options.hpp) that is filled from the environment once,
 *  when the library first needs it -- ARTEMIS_<NAME> present = its integer value, or 1 when it holds no number -- and is
 *  changed at run time only through artemis_hip_set_option("<name>", value) below (names case-insensitive, with or
 *  without the ARTEMIS_ prefix).  No launch path calls getenv.
 *  path selection (every path gives the same bits; tests use these to compare them)
 *    NO_TUNED, TUNED_2D, NO_STAGE2D, NO_FUSED_CURV, NO_CURV_MARCH, NO_CURV_DUST, NO_CURV_DUST_MARCH, NO_DRAG_IN_MARCH
 *    (the drag finish as its own launch instead of inside the dust march), NO_STRAT_IN_KERNEL (the `strat` conditions as
 *    boundary-fill launches instead of inside the 2-D row march), NO_CART_MARCH (Cartesian packs with gravity / viscosity on
 *    the cell-centred stage instead of the tile march of kernels_curv.hip), NO_IC_IN_SHELL (`ic` faces as their own
 *    boundary-fill launches instead of inside the one-launch fill of the copy-type conditions), NO_IC_SKIP (host driver, refined
 *    meshes: `ic` faces refilled at every ghost fill although their zones never change), NO_ML_FUSED,
 *    NO_EPILOGUE, NO_TILED_FLUX, NO_VISC_SOURCE (the diffusion-flux tasks instead of artemis_hip_viscous_source),
 *    NBODY_TASK (N-body gravity as its own task with the host-side reduction), NBODY_GENERAL, NO_PLM_TABLE,
 *    NO_DISTANCE_TABLE, NO_FLAT_RANGES, FULL_REMESH (a remesh rebuilds the whole state next to the old one instead of
 *    the lean hand-over)
 *  exactness machinery
 *    NO_REDO         artemis_hip_stage_fused / the 2-D row march: switches off the detect-and-redo of zones next to
 *                    vanishing velocities (the hand-scheduled divisions are then NOT exact there: DESIGN.md section 4
 *                    states the bound) -- for measuring what the mechanism costs
 *    NO_TINY_HINT    ... and the per-stage hint words that let a plane skip the IEEE path
 *    NO_ML_FLOOR     host driver, refined meshes on the one-kernel stages: leaves out artemis_hip_ml_floor_ghosts (ghost
 *                    zones that took restricted or prolongated values then keep what rounding left below a floor, which
 *                    the reference's PrimToCons would have floored: NOT bit-exact where a floor binds next to a level
 *                    boundary) -- for measuring what the pass costs
 *  host loop
 *    NO_GRAPH        the driver launches every stage kernel itself instead of replaying a captured hipGraph of one step
 *    SYNC_LOOP       the time step comes back to the host every cycle (no device-resident dt)
 *    FORCE_OVERLAP   the shell-first / bulk launch order of a multi-rank run on one rank (bench.py's emulation)
 *    LOOPBACK_COMM, WAIT_SPIN_LIMIT, TEST_SHELL_TARGET_BUMP   transport test hooks
 *    HOST_THREADS=n  host threads of the problem generator; SETUP_TIMING prints its phases; AMR_DEBUG prints the energy
 *                    integral before and after a remesh hand-over
 *  tuning knobs (measurement)
 *    FUSED_KCHUNK, CURV_KCHUNK, VISC_KCHUNK   planes per x3 chunk of the three march kernels
 *    STAGE2D_ROWS, STAGE2D_RGRID, FUSED_NO_SWIZZLE
 *  memory
 *    TRIM_POOL       ... and, set, returns what is left in it to the device after every remesh (30 % less footprint; stalls of
 *                    1 - 2 s when a whole size class of slabs goes back at once)
 *    NO_POOL, POOL_GB=n   the standalone driver enables artemis_rt's buffer cache for adaptive meshes with this limit
 *                    (default 64) / not at all; library hosts: artemis_rt_pool_limit (artemis_rt.h)
 *    POISON          fresh device memory from artemis_rt_malloc holds NaN patterns (a read of something never written
 *                    shows up as NaN instead of whatever the allocator returned)
 *    DENSE_FLUX      the standalone driver gives every block of a refined mesh its flux arrays on the one-kernel stages
 *                    too (default: rows only for the blocks that own a coarse-fine face, the only place they are used)
 * ===================================================================================== */
#ifndef ARTEMIS_HIP_H_
#define ARTEMIS_HIP_H_

#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ARTEMIS_HIP_OK 0
#define ARTEMIS_HIP_EINVAL 1   /* bad argument / unsupported option combination */
#define ARTEMIS_HIP_EDEVICE 2  /* HIP runtime error (no device, launch failure, OOM) */
#define ARTEMIS_HIP_EUNSUPPORTED 3 /* valid in the reference, not built yet */

/* Enumerations use the reference's order (artemis.hpp:78-90). */
enum artemis_coords { ARTEMIS_CARTESIAN = 0, ARTEMIS_CYLINDRICAL = 1, ARTEMIS_SPHERICAL1D = 2,
                      ARTEMIS_SPHERICAL2D = 3, ARTEMIS_SPHERICAL3D = 4, ARTEMIS_AXISYMMETRIC = 5 };
enum artemis_rsolver { ARTEMIS_HLLC = 0, ARTEMIS_HLLE = 1, ARTEMIS_LLF = 2 };
enum artemis_recon { ARTEMIS_PCM = 0, ARTEMIS_PLM = 1, ARTEMIS_PPM = 2 };
enum artemis_fluid { ARTEMIS_GAS = 0, ARTEMIS_DUST = 1 };
/* Physical boundary flags understood by artemis_hip_apply_bc (parthenon outflow /
 * reflecting / periodic, upstream; `none` = face is filled by a neighbour exchange). */
enum artemis_bc { ARTEMIS_BC_PERIODIC = 0, ARTEMIS_BC_OUTFLOW = 1, ARTEMIS_BC_REFLECT = 2,
                  ARTEMIS_BC_NONE = 3,
                  /* user conditions of the `strat` problem (pgen/strat.hpp:158-466, registered as
                   * `extrap` / `inflow` at problem_modifier.hpp:114-128): */
                  ARTEMIS_BC_STRAT_EXTRAP = 4, /* x1 faces; x3 faces (density continued with pow() of a state
                                                 * ratio on the device: to rounding, not bitwise) */
                  ARTEMIS_BC_STRAT_INFLOW = 5, /* x2 faces */
                  /* `conductive` of the `conduction` problem (pgen/conduction.hpp:105-232): fixed heat
                   * flux through inner faces, fixed temperature at outer ones */
                  ARTEMIS_BC_CONDUCTIVE = 6,
                  /* user conditions of the `disk` problem (pgen/disk.hpp, registered as `ic` /
                   * `extrap` at problem_modifier.hpp:67-96), any face: */
                  ARTEMIS_BC_IC = 7,          /* DiskBoundaryIC, disk.hpp:597-632 */
                  ARTEMIS_BC_DISK_EXTRAP = 8, /* DiskBoundaryExtrap, disk.hpp:634-825 */
                  ARTEMIS_BC_DISK_VISC = 9    /* DiskBoundaryVisc (`viscous`), disk.hpp:415-595, x1 faces */ };
enum artemis_gravity_type { ARTEMIS_GRAVITY_UNIFORM = 1, ARTEMIS_GRAVITY_POINT = 2, ARTEMIS_GRAVITY_BINARY = 3 };
enum artemis_drag_type { ARTEMIS_DRAG_SIMPLE_DUST = 1, ARTEMIS_DRAG_SELF = 2 }; /* drag.hpp:57 */
enum artemis_drag_model { ARTEMIS_DRAG_CONSTANT = 0, ARTEMIS_DRAG_STOKES = 1 }; /* drag.hpp:58 */
#define ARTEMIS_MAX_DUST_SPECIES 16 /* drag parameter arrays travel by value as kernel arguments */

typedef struct artemis_fluid_pack {
  int nspecies;              /* 0 = fluid absent (physics/gas|dust = false, artemis.cpp:63-64) */
  int recon;                 /* <gas|dust>/reconstruct  (gas.cpp:59-82, dust.cpp:50-74)  */
  int riemann;               /* <gas|dust>/riemann      (gas.cpp:84-96, dust.cpp:76-86); dust: hlle|llf only */
  double dfloor;             /* <gas|dust>/dfloor       (gas.cpp:170, dust.cpp:93) */
  double siefloor;           /* gas/siefloor            (gas.cpp:171) */
  double de_switch;          /* gas/de_switch           (gas.cpp:177) */
  double *const *prim;       /* [nblocks * nprim] */
  double *const *cons0;      /* u0: [nblocks * ncons] */
  double *const *cons1;      /* u1 (start-of-step copy, artemis_driver.cpp:157-163) */
  double *const *flux[3];    /* [nblocks * ncons] per direction */
  double *const *pflux[3];   /* gas only: [nblocks * ns] interface pressure */
  double *const *vface[3];   /* gas only: [nblocks * ns] face-normal velocity */
  double *const *diff_flux[3]; /* gas only, optional: [nblocks * 4ns] diffusion fluxes, gas::diff::momentum
                                  (3n + component) then gas::diff::energy (3ns + n), gas.cpp:281-284 */
} artemis_fluid_pack_t;

typedef struct artemis_pack {
  int nblocks;               /* MeshData::NumBlocks() */
  int nghost;                /* parthenon/mesh/nghost */
  int nx1, nx2, nx3;         /* interior cells per block (parthenon/meshblock) */
  int coords;                /* artemis/coordinates after geometry::CoordSelect (geometry.hpp:38-56) */
  double gm1;                /* gamma - 1 = IdealGas Gruneisen parameter (hllc.hpp:60) */
  const double *geom;        /* DEVICE [nblocks][6] = {x1f0, dx1, x2f0, dx2, x3f0, dx3}:
                                Coordinates_t::Xf<d>(idx) = xf0 + idx*dx, idx counted from the
                                first ghost cell (geometry.hpp:65-72) */
  const double *metric;      /* DEVICE trigonometry tables (see artemis_hip_metric_count / _fill);
                                required for spherical2D/3D, optional otherwise */
  artemis_fluid_pack_t gas, dust;
  double omega_frame;        /* rotating_frame/omega when physics/rotating_frame is on, else 0: the frame
                                velocity RotationVelocity<GEOM>(xv, omf) inside FluxSource's coordinate
                                source terms (fluid_fluxes.hpp:345, :395-415, :433-437) */
  const double *plm_table;   /* optional DEVICE table of PLM_G's geometric weights (artemis_hip_plm_table_count /
                                _fill), or NULL: CalculateFluxes then forms them per face */
} artemis_pack_t;

/* PLM_G (plm.hpp:54-73) weighs every limited slope with quotients of cell-centroid and face positions along the
 * sweep direction -- geometry of the block, a function of the index along that direction only: (x_i+1 - x_i) /
 * (x_f1 - x_i), (x_i - x_i-1) / (x_i - x_f0), the two face offsets and the refined reciprocals of the two centroid
 * distances, plus the x1 centroid of every column (the cell widths of PLM_G's x2 / x3 sweeps carry it as a factor and
 * stay per cell).  Forming them per face costs ~40 divisions per zone in CalculateFluxes on a curvilinear mesh (half of
 * that kernel's instructions -- though only 3 % of its time, which goes to memory latency); the table holds them once
 * per mesh: 3 directions x 9 rows of max(ni, nj, nk) doubles
 * per block, filled on the device with the same functions the kernels would call -- the same bits.  Host-owned like
 * the metric tables: fill after p->geom / p->metric are in place (and again after a remesh), point p->plm_table at it. */
long artemis_hip_plm_table_count(const artemis_pack_t *p);
int artemis_hip_plm_table_fill(const artemis_pack_t *p, double *table_dev, void *stream);

/* ---- Parthenon task functions ---------------------------------------------------------*/

/* Gas::CalculateFluxes / Dust::CalculateFluxes (gas.cpp:473-494, dust.cpp:281-298) ->
 * ArtemisUtils::CalculateFluxes<FLUID> (fluid_fluxes.hpp:78-292).  Reads prim (stencil +-2
 * PLM / +-3 PPM per direction), writes flux[d], and for gas pflux[d], vface[d], on faces
 * [s, e+1] of every active direction.  `pcm` forces first-order reconstruction (VL2 stage
 * 1, artemis_driver.cpp:182).
 * Gas, one species, PCM / PLM, Cartesian blocks of at least 32 x 8 zones in 2-D / 3-D run through the
 * LDS-staged tile march of the fused stage (same device functions, same bits).  That path reads density,
 * velocity and specific internal energy and derives the pressure as FillDerived stores it
 * (max(0, (gamma-1) rho sie), fill_derived.cpp:246-262) instead of loading prim's pressure block: the input
 * is a state FillDerived has visited, as everywhere in the reference's task list.  ARTEMIS_NO_TILED_FLUX=1
 * keeps the one-thread-per-zone kernel, which loads the stored pressure. */
int artemis_hip_calculate_fluxes(const artemis_pack_t *p, int fluid, int pcm, void *stream);

/* ArtemisUtils::ApplyUpdate<GEOM> (artemis_integrator.hpp:57-110) for every
 * Conserved+WithFluxes variable of both fluids: u0 = gam0*u0 + gam1*u1 + beta_dt/V*div(F). */
int artemis_hip_apply_update(const artemis_pack_t *p, double gam0, double gam1, double beta_dt,
                             void *stream);

/* Gas::FluxSource / Dust::FluxSource (gas.cpp:499-519, dust.cpp:303-326) ->
 * FluxSourceImpl (fluid_fluxes.hpp:300-420): pressure gradient on momentum and -P div(v)
 * on internal energy, then the coordinate sources rho dt sum_d dh_d/dx_a (v_d + vf_d)^2 with the
 * frame velocity vf = RotationVelocity<GEOM>(xv, p->omega_frame) (:345, :395-415).  Interior cells
 * only (the reference also scribbles on ghost cells [is-2, ie+1] that PrimToCons overwrites; see
 * DESIGN.md).  Cartesian dust: no-op. */
int artemis_hip_flux_source(const artemis_pack_t *p, int fluid, double dt, void *stream);

/* ArtemisDerived::SetAuxillaryFields<GEOM> (fill_derived.cpp:30-75). */
int artemis_hip_set_aux(const artemis_pack_t *p, void *stream);

/* ArtemisDerived::ConsToPrim<GEOM> (fill_derived.cpp:82-167), interior cells. */
int artemis_hip_cons_to_prim(const artemis_pack_t *p, void *stream);

/* ArtemisDerived::PrimToCons<MeshData, GEOM> (fill_derived.cpp:173-277), entire block. */
int artemis_hip_prim_to_cons(const artemis_pack_t *p, void *stream);
/* PrimToCons on the GHOST zones only.  For a host that keeps `cons` current: artemis_hip_stage_fused with cons_out set
 * stores the conserved state of every zone it updates (the bits PrimToCons would give), so after the boundary
 * exchange only the ghost zones are left to convert -- 5 % of a 256^3 block instead of a whole-block pass. */
int artemis_hip_prim_to_cons_ghosts(const artemis_pack_t *p, void *stream);

/* ArtemisUtils::DeepCopyConservedData (artemis_integrator.hpp:30-51): u1 <- u0, entire. */
int artemis_hip_deep_copy_conserved(const artemis_pack_t *p, void *stream);

/* Gas::EstimateTimestepMesh / Dust::EstimateTimestepMesh (gas.cpp:392-468,
 * dust.cpp:239-276): *dt_out = cfl * min over interior cells.  SYNCHRONOUS (returns a host
 * Real like the reference's par_reduce). */
int artemis_hip_estimate_dt(const artemis_pack_t *p, int fluid, double cfl, double *dt_out,
                            void *stream);
/* Asynchronous form: min-combines cfl*min(...) into the DEVICE scalar *dt_dev (the caller
 * initialises it, e.g. to DBL_MAX). */
int artemis_hip_estimate_dt_async(const artemis_pack_t *p, int fluid, double cfl, double *dt_dev,
                                  void *stream);

/* Metric tables.  geometry::Coords<GEOM> evaluates transcendentals per cell: for spherical 2-D/3-D
 * cos/sin of the x2 faces, the x2 centroid and the x2 midpoint (spherical.hpp:61-146); for
 * ConvertCoordsToCart (used by Coords::Distance in the diffusion tasks) cos/sin of the azimuth of
 * the cell centre -- x2 in cylindrical, x3 in spherical3D / axisymmetric coordinates.  All depend on
 * one index only, so the adapter tabulates them once per mesh with the host libm.
 * artemis_hip_metric_count = number of doubles (0 for Cartesian and spherical1D);
 * artemis_hip_metric_fill writes them to HOST memory from a HOST copy of p->geom (p->geom itself is
 * a device pointer and is not read).  Upload the result and set p->metric.  Required for spherical
 * 2-D/3-D always, for cylindrical / axisymmetric only by the diffusion entry points and the
 * CONDUCTIVE boundary condition. */
long artemis_hip_metric_count(const artemis_pack_t *p);
int artemis_hip_metric_fill(const artemis_pack_t *p, const double *geom_host, double *out_host);

/* Physical boundary conditions on the FillGhost primitives (gas rho, v, sie; dust rho, v;
 * gas.cpp:244-270, dust.cpp:201-213) of every block: bc[b*6 + {ix1,ox1,ix2,ox2,ix3,ox3}]
 * is a HOST array of artemis_bc values.  Parthenon order: periodic images first, then
 * x1, x2, x3, each over the entire extent of the other dimensions.  `params` (may be NULL
 * unless a STRAT flag is present) carries what the strat conditions read from StratParams
 * (strat.hpp:44-52, :60-61): the shear rate q and the frame frequency Om0. */
typedef struct artemis_bc_params {
  double qshear, omega;       /* strat conditions */
  /* conductive conditions (conduction.hpp:30-37 CondParams, :181-196): boundary temperature, heat
   * flux, uniform gravity along x1/x2/x3 (0 when gravity is off or not uniform), the constant heat
   * conductivity K or diffusivity (K = kappa*rho*cv) and the IdealGas specific heat */
  double cond_temp, cond_flux, cond_g[3], cond_coeff, cond_cv;
  int cond_type;              /* ARTEMIS_CONDUCTIVITY_PLAW | ARTEMIS_THERMALDIFF_PLAW */
  double cond_temp_exp, cond_rho_exp, cond_T_ref, cond_rho_ref; /* its power laws (0 exponents: bit-exact;
                                 otherwise pow() of the state on the device, to rounding) */
  /* disk conditions (pgen/disk.hpp).  IC: DEVICE pointer tables laid out like gas.prim /
   * dust.prim holding the initial primitives over the entire block (the time-independent disk
   * profile DiskBoundaryIC re-evaluates per call, :597-632); ghost zones are copied from them.
   * DISK_EXTRAP (:634-825): the frame frequency DiskParams::omf.  That condition takes log / exp
   * of the state on the device: it matches a host libm to rounding, not bitwise. */
  double *const *ic_gas, *const *ic_dust;
  double disk_omf;
  /* DISK_VISC: ViscosityProfile nu0 (R/r0)^nu_indx (disk.hpp:131-135) and the accretion rate */
  double disk_nu0, disk_nu_indx, disk_r0, disk_mdot;
  /* non-zero: finish by applying PrimToCons's primitive floors (fill_derived.cpp:227, :245, :262) to
   * the ghost zones of every block.  The reference runs PrimToCons over the entire block right after
   * the conditions (artemis_driver.cpp:258-261), which floors what user conditions wrote; a caller of
   * the fused stages (which rebuild the conserved state in registers and never run PrimToCons) sets
   * this instead. */
  int floor_ghosts;
  /* 3: leave the x1 ghost columns of the ACTIVE rows alone (both x1 faces; the x2 / x3 ghost rows and planes are
   * filled over the entire x1 extent as always).  For callers of artemis_hip_stage_fused with outflow_faces bits 0 and 1, which
   * does not read those columns.  Ignored when a block of the pack carries a user condition.  0 = fill everything. */
  int x1_interior_done;
} artemis_bc_params_t;
int artemis_hip_apply_bc(const artemis_pack_t *p, const int *bc, const artemis_bc_params_t *params,
                         void *stream);

/* ---- source-term tasks between FluxSource and SetAuxillaryFields ------------------------*/

/* Gravity::ExternalGravity<GEOM> (gravity/gravity.cpp:126-155), active for
 * tstart <= time < tstop: UniformGravity (uniform.cpp:28-84, every coordinate system) or
 * PointMassGravity (point_mass.cpp:27-198: Cartesian with offset mass, softening and sink;
 * spherical1D/2D and axisymmetric with the mass at the origin).  gm = G*mass in code units
 * (gravity.cpp:57).  BinaryMassGravity (binary_mass.cpp:27-203): two softened point masses
 * with sinks at positions the adapter computes from the orbit; Cartesian, cylindrical, spherical3D
 * (not the axisymmetric systems, gravity.cpp:82-83).  Reads prim, updates cons0 momenta / total
 * energy (/ density for the sinks) of both fluids on interior cells.  The nbody type has its own
 * entry point, artemis_hip_nbody_gravity. */
typedef struct artemis_gravity {
  int type;                   /* artemis_gravity_type */
  double g[3];                /* <gravity/uniform> gx1, gx2, gx3 */
  double gm, soft, sink, sink_rate, pos[3]; /* <gravity/point>; BINARY: gm of the pair and body 1's
                                               soft1, sink1, sink_rate1, position */
  double tstart, tstop;       /* <gravity> tstart, tstop (gravity.cpp:36-38) */
  /* BINARY (gravity/binary_mass.cpp:27-203): mass ratio q = m2/m1, body 2's softening, sink and
   * position.  The adapter evaluates Orbit::solve(time, omf) (gravity.hpp:66-94, host libm) per
   * stage and passes pos = com - mu2 rb, pos2 = com + mu1 rb (binary_mass.cpp:56-70). */
  double q, soft2, sink2, sink_rate2, pos2[3];
} artemis_gravity_t;
int artemis_hip_external_gravity(const artemis_pack_t *p, const artemis_gravity_t *g, double time,
                                 double dt, void *stream);

/* Gravity::NBodyGravity<GEOM> (gravity/nbody_gravity.hpp:28-221; gravity type `nbody`, gravity.cpp:110-117,
 * :150-155): acceleration and accretion from the nbody package's particles (nbody/particle_base.hpp:96-258:
 * Plummer or spline softening, sink radius with mass / momentum removal) on both fluids, and the back-reaction
 * on every particle -- {mass accreted, gravity force x3, accretion force x3} per unit time, the seven numbers
 * NBody::Advance reduces over ranks and hands to REBOUND (nbody/nbody_advance.cpp:123-131).  Cartesian,
 * cylindrical and spherical3D (nbody.cpp:61: not the axisymmetric systems).
 *   particles : HOST array, the state the host's N-body integrator currently holds (positions / velocities in
 *               the simulation frame; xf / vf = frame origin and velocity, particle_base.hpp:86-93)
 *   omf       : rotating_frame/omega when the frame correction is on (nbody_gravity.hpp:179-186), else 0
 *   force     : HOST [npart][7], ADDED to (the reference accumulates particle_force over the stage's calls)
 * Reads prim, updates cons0 of the interior zones particle by particle in order; the fluid update is bitwise
 * reproducible, the seven sums are reduced in a fixed tree (workgroup partials, then the host in order) and
 * agree with a serial sum to round-off.  SYNCHRONOUS (the reference's par_reduce returns host values). */
typedef struct artemis_nbody_particle {
  double gm, pos[3], vel[3], xf[3], vf[3];
  double rs, racc, gamma, beta; /* softening radius; sink radius and its mass / momentum removal rates */
  int spline, couple;           /* soft type spline (1) or plummer / none (0); couple = 0: ignored */
} artemis_nbody_particle_t;
int artemis_hip_nbody_gravity(const artemis_pack_t *p, const artemis_nbody_particle_t *particles, int npart, double omf,
                              double time, double dt, double *force, void *stream);

/* RotatingFrame::RotatingFrameForce (rotating_frame/rotating_frame.cpp:56-86).  Cartesian:
 * ShearingBoxImpl (rotating_frame_impl.hpp:28-93), tidal potential differenced across the cell
 * plus the Coriolis force, gas and dust.  Every other system: RotatingFrameImpl<GEOM> (:95-199), the
 * angular-momentum-conserving source from the mass fluxes CalculateFluxes left in flux[d] (qshear must
 * be 0, rotating_frame.cpp:34-38); the centrifugal / Coriolis terms of the other two momenta enter
 * through p->omega_frame in artemis_hip_flux_source. */
int artemis_hip_rotating_frame_force(const artemis_pack_t *p, double omega, double qshear,
                                     double time, double dt, void *stream);

/* Drag::DragSource<GEOM> (drag/drag.cpp:89-175): `self` =
 * SelfDragSourceImpl (drag.hpp:171-294, damping ramps towards the mesh edges), `simple_dust` =
 * SimpleDragSourceImpl (drag.hpp:296-482, implicit gas-dust momentum exchange; one gas species,
 * <= ARTEMIS_MAX_DUST_SPECIES dust species).  tau[n] already includes `scale` for the constant
 * model (drag.hpp:129-137); `stokes` uses scale, grain_density, sizes (drag.hpp:407-409). */
typedef struct artemis_damping {
  double ix[3], ox[3], irate[3], orate[3]; /* SelfDragParams, drag.hpp:68-117 */
} artemis_damping_t;
typedef struct artemis_drag {
  int type, model;            /* artemis_drag_type, artemis_drag_model */
  double scale, grain_density;
  double tau[ARTEMIS_MAX_DUST_SPECIES], sizes[ARTEMIS_MAX_DUST_SPECIES];
  artemis_damping_t gas, dust;
  double xmin[3], xmax[3];    /* parthenon/mesh x?min, x?max (drag.cpp:37-42) */
  /* <gas/damping> damp_to_visc (drag.hpp:101, drag.cpp:109-121,135-157): NULL = off; otherwise the gas
   * package's viscosity (viscosity_plaw or viscosity_alpha, with its radial table) -- the gas damping
   * relaxes towards the viscous inflow velocity v_R = -1.5 mu / (R rho) instead of towards rest. */
  const struct artemis_diffcoeff *damp_visc;
} artemis_drag_t;
int artemis_hip_drag_source(const artemis_pack_t *p, const artemis_drag_t *d, double time, double dt,
                            void *stream);

/* Gas::Cooling::CoolingSource<GEOM> (gas/cooling/cooling.cpp:94-106 -> beta_cooling.cpp:40-126), task at
 * artemis_driver.cpp:243-248 (after DragSource, before SetAuxillaryFields): backward-Euler
 * relaxation of the gas temperature towards Tref(R, r) = tfloor + tcyl R^a + tsph r^b on the
 * time scale beta / Omega_K, beta = beta_min + beta0 exp(-exp_scale z^2 / Tref); reads cons0,
 * updates total and internal energy.  Tref and beta are functions of the cell position with
 * std::pow / std::exp in them: the adapter tabulates both once per mesh on the host
 * (artemis_hip_cooling_table_fill; arrays over the entire block) and the task stays bit-exact.
 * tref = nbody: ARTEMIS_HIP_EUNSUPPORTED. */
typedef struct artemis_cooling {
  double beta0, beta_min, exp_scale;           /* <cooling> beta0, beta_min, exp_scale */
  double tfloor, tcyl, cyl_plaw, tsph, sph_plaw; /* TempParams, cooling.hpp:37-43 */
  double gm;                                   /* gravity package's gm (NaN = Null<Real>() when gravity is off) */
  double cv;                                   /* IdealGas specific heat */
  const double *const *tref, *const *beta;     /* DEVICE tables [nblocks] of per-cell arrays */
} artemis_cooling_t;
int artemis_hip_cooling_table_fill(const artemis_pack_t *p, const double *geom_host, const double *metric_host,
                                   const artemis_cooling_t *c, int block, double *tref_host, double *beta_host);
int artemis_hip_cooling_source(const artemis_pack_t *p, const artemis_cooling_t *c, double time, double dt,
                               void *stream);

/* ---- Fused stage (the fast path) --------------------------------------------------------
 * One RK stage of artemis_driver.cpp:182-261 with every optional package disabled, i.e.
 *   CalculateFluxes -> ApplyUpdate -> FluxSource -> SetAuxillaryFields -> ConsToPrim ->
 *   [interior part of] PrimToCons
 * in ONE pass over the block, fluxes kept in registers/LDS.  It exploits the invariant that
 * at the start of a stage u0 == PrimToCons(prim) cell by cell (fill_derived.cpp:212-276 is
 * applied to the entire block at the end of every stage and by PostInitialization), so the
 * conserved state is rebuilt in registers instead of being read.
 *
 *   prim_in   : gas prim table at stage start (ghosts filled); read with the stencil.
 *   prim_u1   : gas prim table of the start-of-STEP state (read cell-wise, to rebuild u1);
 *               pass prim_in itself for stage 1.
 *   prim_out  : where the new interior primitives go.  May alias prim_u1 (cell-wise access
 *               only) but NOT prim_in.
 *   cons_out  : optional cons table for u0 (NULL = do not materialise; call
 *               artemis_hip_prim_to_cons when the conserved state is needed).
 *   dt_dev    : optional DEVICE scalar; when non-NULL the new-state timestep estimate
 *               cfl*min(...) of gas.cpp:411-433 is min-combined into it (fused
 *               EstimateTimestepMesh for the last stage).
 * Results are bit-identical to the unfused sequence above. Gas only (dust: DESIGN.md).
 * The P entries of prim_out are written; ghosts of prim_out are NOT touched (apply_bc /
 * halo exchange follow, as in the reference).
 *
 * Exactness next to vanishing velocities (detect-and-redo).  The kernel divides through hand-scheduled refined
 * reciprocals, which give the bits of an IEEE division unless a numerator is non-zero and below 2^-969 -- reachable only
 * from velocities below 2^-200 (1e-61: ahead of a shock the limited slopes square the perturbation from zone to zone).
 * A zone whose stencil holds such a velocity is therefore NOT stored by the kernel; its id goes to a list (library-owned,
 * per calling thread) and a second, list-driven kernel enqueued behind it on `stream` by the same call computes it with
 * IEEE arithmetic from prim_in / prim_u1.  The list is normally empty; ARTEMIS_NO_REDO=1 switches the mechanism off
 * (every zone is then stored by the fast kernel: the pre-round-3 behaviour with its 1e-120 parity limit).  The one case
 * the call cannot finish by itself is the shell of a shell-first launch (shell_done != NULL): those zones must be
 * final before the stream that waits for the shell packs them -- call artemis_hip_stage_fused_redo_shell with the same
 * arguments on THAT stream, after artemis_hip_wait_counter. */
typedef struct artemis_stage_args {
  double gam0, gam1, beta_dt; /* LowStorageIntegrator weights, artemis_integrator.hpp:64-66 */
  double bdt;                 /* beta*dt handed to FluxSource, artemis_driver.cpp:168,211 */
  int pcm;                    /* artemis_driver.cpp:182 */
  double *const *prim_in, *const *prim_u1, *const *prim_out, *const *cons_out;
  double cfl;
  double *dt_dev;
  int region;                 /* 0 = all interior cells; 1 = only the boundary shell (the cells
                                 neighbours' ghost slabs are cut from, rounded out to whole
                                 tiles); 2 = only the rest.  1 then 2 == 0: lets a driver send
                                 halos while the bulk is still being computed. */
  unsigned *shell_done;       /* optional DEVICE counter (zeroed by the caller before the launch).
                                 When non-NULL (region must be 0) the whole block is computed in
                                 ONE launch with the boundary-shell workgroups first; each of them
                                 publishes its stores at agent scope and increments the counter
                                 when done.  *shell_target (HOST, optional) receives the count to
                                 wait for with artemis_hip_wait_counter on another stream. */
  unsigned *shell_target;
  const double *beta_dt_dev;  /* optional DEVICE scalar holding beta*dt for this stage: when
                                 non-NULL it replaces the host values beta_dt and bdt above
                                 (artemis_integrator.hpp:66, artemis_driver.cpp:168), so a driver
                                 can keep dt on the device and never synchronise */
  int shell_faces;            /* bit f set: face f (0..5 = ix1,ox1,ix2,ox2,ix3,ox3) has a neighbour
                                 whose ghosts are cut from this block; 0 = all six */
  /* Optional hint words (DEVICE, one unsigned each; all NULL = detect in every stage, the safe default).  The per-zone
   * detection of vanishing velocities (below) costs ~2.5 % of the kernel; a caller who controls EVERY producer of the
   * primitives it passes as prim_in can let the kernel skip it in stages where no such velocity exists anywhere:
   *   tiny_out   the launch ORs 1 into it when it stores a velocity below 2^-200 (the redo kernel likewise);
   *   tiny_in    the word a previous launch produced as tiny_out for the state now in prim_in: detection runs only if
   *              it is non-zero.  Valid only if every zone of prim_in (ghost zones included) was written by launches
   *              that reported into that word, or copied from such zones (periodic / outflow / reflecting conditions,
   *              same-rank block-to-block slabs);
   *   tiny_clear zeroed AFTER the launch's kernels have finished (by the list-driven kernel the call enqueues behind the
   *              stage kernel): normally the stage's own tiny_in word, so that a ring of one word per stage of a time
   *              step -- in = word[s], out = word[(s + 1) % nstages] -- needs no memset and uses the same pointers every
   *              step.  With separate region 1 / 2 launches pass it only with the last launch of the stage. */
  const unsigned *tiny_in;
  unsigned *tiny_out, *tiny_clear;
  /* bit f (f = 0..5: lower / upper x1, x2, x3 face): that face of EVERY block of the pack is a parthenon `outflow`
   * boundary and the kernel shall not read the ghost zones behind it: it stages the first / last active zone of the row,
   * column or march in their place -- the value an outflow condition puts there.  A caller that sets a bit may leave
   * those ghost zones unfilled between stages (artemis_bc_params_t.x1_interior_done for the x1 columns -- the strided
   * third of the boundary shell and most of its cost --, or no artemis_hip_apply_bc call at all when every physical
   * face of the pack is covered) as long as it fills them before anything else reads the state.  0 = read the ghosts. */
  int outflow_faces;
  /* Optional HOST array [nblocks]: the same mask block by block (outflow_faces is then ignored) -- a pack whose blocks
   * differ in which of their faces are physical (two blocks of a rank stacked along x3: the face between them is a
   * neighbour's, the outer ones are outflow).  Packs of up to 10 blocks; more: ARTEMIS_HIP_EUNSUPPORTED. */
  const unsigned char *outflow_faces_by_block;
  /* Optional DEVICE scratch for the detect-and-redo lists (zones next to vanishing velocities, deferred to the exact
   * kernel), artemis_hip_redo_scratch_bytes(p) bytes, zeroed once by the caller.  NULL = the library's own buffers, one
   * set per calling THREAD (allocated on first use, grown with a device synchronisation): fine for one state and one
   * compute stream per thread.  A caller with several states or streams in flight on one thread passes one scratch
   * per state -- two launches must not share lists -- and the same scratch to artemis_hip_stage_fused_redo_shell.  The
   * lists have room for every active zone of the pack, so they cannot overflow as long as each launch's list is
   * drained (the call enqueues the draining kernel itself; the shell list of a shell_done launch is drained by
   * artemis_hip_stage_fused_redo_shell). */
  void *redo_scratch;
} artemis_stage_args_t;
size_t artemis_hip_redo_scratch_bytes(const artemis_pack_t *p);
int artemis_hip_stage_fused(const artemis_pack_t *p, const artemis_stage_args_t *a, void *stream);
int artemis_hip_stage_fused_redo_shell(const artemis_pack_t *p, const artemis_stage_args_t *a, void *stream);
/* ---- gas diffusion (viscosity, heat conduction) ------------------------------------------
 * Gas::ZeroDiffusionFlux / ViscousFlux<GEOM> / ThermalFlux<GEOM> / DiffusionUpdate<GEOM>
 * (gas.cpp:522-641 -> utils/diffusion/{diffusion,momentum_diffusion,thermal_diffusion}.hpp),
 * tasks at artemis_driver.cpp:189-193 and :218-221, and the diffusive timestep limit folded into
 * Gas::EstimateTimestepMesh (gas.cpp:435-467, diffusion.hpp:66-108).  Fluxes go to
 * p->gas.diff_flux[d] on faces [s, e+1] like the hydro fluxes.
 * Built: every coordinate system (cylindrical / axisymmetric blocks need p->metric for
 * Coords::Distance); viscosity `constant`, `powerlaw` (nu (R/r0)^r_exp) and `alpha`
 * (alpha B / (Omega0 (r/r0)^-1.5)); conductivity / diffusivity with temp_exp = rho_exp = 0;
 * arithmetic or harmonic face averaging.  The reference evaluates std::pow per cell: the radial
 * factors depend on the cell centre only, so the adapter tabulates them once per mesh with the
 * HOST libm (artemis_hip_diffusion_radial_fill -> artemis_diffcoeff_t.radial) and results stay
 * bit-identical; the temperature / density power laws of the conductivity depend on the state:
 * exactly 1 for zero exponents (bit-exact), pow() on the device otherwise (to rounding). */
enum artemis_diff_type { ARTEMIS_DIFF_OFF = 0, ARTEMIS_VISCOSITY_PLAW = 1, ARTEMIS_VISCOSITY_ALPHA = 2,
                         ARTEMIS_CONDUCTIVITY_PLAW = 3, ARTEMIS_THERMALDIFF_PLAW = 4 };
typedef struct artemis_diffcoeff { /* DiffCoeffParams, diffusion_coeff.hpp:58-136 */
  int type;                   /* artemis_diff_type */
  int avg;                    /* 0 arithmetic, 1 harmonic (diffusion_coeff.hpp:50-56) */
  double coeff;               /* nu | alpha | cond | kappa */
  double eta, r_exp, r0, omega0;
  double temp_exp, rho_exp, rho_ref, T_ref;
  const double *const *radial; /* viscosity only: DEVICE table [nblocks] of per-cell arrays (entire
                                  block incl. ghosts) filled by artemis_hip_diffusion_radial_fill;
                                  required for ALPHA and for PLAW with r_exp != 0, else NULL */
} artemis_diffcoeff_t;
typedef struct artemis_diffusion {
  artemis_diffcoeff_t visc, cond;
  double cv;                  /* IdealGas specific heat: T = sie / cv (gas.cpp:106-116) */
  const double *dist;         /* optional DEVICE table of artemis_hip_viscous_distance_count(p) doubles filled by
                                 artemis_hip_viscous_distance_fill: the Coords::Distance values between neighbouring
                                 cell centres (geometry.hpp:407-412) the flux tasks divide by.  They are static
                                 geometry; with the table the viscous / thermal flux tasks look them up instead of
                                 evaluating 15 square roots per cell and stage (same bits).  NULL = on the fly. */
} artemis_diffusion_t;
/* The table behind artemis_diffusion_t.dist: [6][nblocks][(nx3+2g)(nx2+2g)(nx1+2g)] doubles (directions that are
 * not active stay untouched).  Fill once per mesh (again after a remesh or a change of p->geom / p->metric). */
size_t artemis_hip_viscous_distance_count(const artemis_pack_t *p);
int artemis_hip_viscous_distance_fill(const artemis_pack_t *p, double *table_dev, void *stream);
/* Radial factor of one block's cells for a viscosity law, on the HOST with the host libm:
 * PLAW  -> std::pow(R / r0, r_exp), R = ConvertToCyl(cell centre)[0]   (diffusion_coeff.hpp:222-224)
 * ALPHA -> omega0 * std::pow(r / r0, -1.5), r = ConvertToSph(centre)[0]  (:262-264)
 * geom_host / metric_host = host copies of p->geom / p->metric (metric may be NULL where
 * artemis_hip_metric_count is 0); out_host receives (nx3+2g)(nx2+2g)(nx1+2g) doubles. */
int artemis_hip_diffusion_radial_fill(const artemis_pack_t *p, const double *geom_host,
                                      const double *metric_host, const artemis_diffcoeff_t *c,
                                      int block, double *out_host);
int artemis_hip_zero_diffusion_flux(const artemis_pack_t *p, void *stream);
int artemis_hip_viscous_flux(const artemis_pack_t *p, const artemis_diffusion_t *d, void *stream);
/* Gas::ZeroDiffusionFlux followed by Gas::ViscousFlux as ONE task (artemis_driver.cpp:189-191 are adjacent in the
 * task list): the viscous fluxes are STORED (0.0 + flux, the bits of zero-then-add) on the face ranges the update
 * and the flux correction read, instead of zeroing the arrays and adding to them in three more passes.  Entries
 * outside those ranges keep their old contents.  With viscosity off it only zeroes. */
int artemis_hip_zero_viscous_flux(const artemis_pack_t *p, const artemis_diffusion_t *d, void *stream);
int artemis_hip_thermal_flux(const artemis_pack_t *p, const artemis_diffusion_t *d, void *stream);
int artemis_hip_diffusion_update(const artemis_pack_t *p, const artemis_diffusion_t *d, double dt,
                                 void *stream);
/* Gas::ZeroDiffusionFlux, Gas::ViscousFlux and the viscous part of Gas::DiffusionUpdate (artemis_driver.cpp:189-191,
 * :218-221; momentum_diffusion.hpp:591-759, diffusion.hpp:110-241) as ONE source: for every active zone the five numbers
 * DiffusionUpdate subtracts from the zone's conserved state -- dt * div(viscous momentum flux) with its metric terms
 * for M1, M2, M3, dt * div(viscous energy flux) for E, and that minus the work term dt * (div F_M) . v / h for e_int --
 * formed from p->gas.prim (the stage's input primitives, ghost zones filled, edge and corner zones included) in one
 * LDS-staged march, with the expression trees of the three tasks: the bits DiffusionUpdate would subtract after
 * ZeroDiffusionFlux -> ViscousFlux, but no diffusion-flux array is read or written (twelve stores and twenty-four
 * loads per zone and stage less).  Hand the result to artemis_hip_stage_general as `diffusion_sums`.
 *   d->dist     : REQUIRED here (artemis_hip_viscous_distance_fill): the march reads the table a plane ahead of its use
 *   dt / dt_dev : beta * dt of the stage (dt_dev: optional DEVICE scalar that replaces dt)
 *   sums        : DEVICE table [nblocks * 5] of cell arrays (entire-block extents; only active zones are written)
 * artemis_hip_viscous_source_covers: non-zero when the march covers the pack (3-D blocks at least 8 x 8 zones wide,
 * one gas species); otherwise, and whenever heat conduction is on (it adds to the same energy flux) or a flux
 * correction needs the face fluxes themselves, use the flux tasks. */
int artemis_hip_viscous_source_covers(const artemis_pack_t *p);
int artemis_hip_viscous_source(const artemis_pack_t *p, const artemis_diffusion_t *d, double dt, const double *dt_dev,
                               double *const *sums, void *stream);
/* min-combines cfl * min(viscous, conductive limit) into the DEVICE scalar *dt_dev */
int artemis_hip_diffusion_dt(const artemis_pack_t *p, const artemis_diffusion_t *d, double cfl,
                             double *dt_dev, void *stream);

/* Gas::EstimateTimestepMesh with its diffusive limits (gas.cpp:411-467) and Dust::EstimateTimestepMesh (dust.cpp:256-272)
 * of p's primitives in ONE pass: what artemis_hip_estimate_dt_async for both fluids and artemis_hip_diffusion_dt
 * min-combine into *dt_dev, with one read of the state (every limit is a minimum over the zones: same bits).
 * d = NULL: no diffusive limits. */
int artemis_hip_timestep_all(const artemis_pack_t *p, double cfl_gas, double cfl_dust, const artemis_diffusion_t *d,
                             double *dt_dev, void *stream);

/* General fused stage: the same contract as artemis_hip_stage_fused for EVERY configuration the
 * per-task entry points accept -- gas and/or dust, any number of species, PCM/PLM/PPM, all six
 * coordinate systems, with ExternalGravity, RotatingFrameForce and DragSource between FluxSource
 * and SetAuxillaryFields (artemis_driver.cpp:182-255).  One cell-centred kernel per fluid
 * (csrc/kernels_stage_cell.hip): no flux / pressure-flux / face-velocity arrays are touched.
 *   *_in  : prim tables of the state at the start of the stage (ghosts filled; only the FillGhost
 *           variables rho, v, sie are read -- gas pressure is recomputed, fill_derived.cpp:247)
 *   *_u1  : prim tables of the start-of-step state (cell-wise), *_in itself for stage 1
 *   *_out : where the new interior primitives go (rho, v, sie); may alias *_u1, not *_in
 *   gravity / drag : NULL = package off; rf_omega == 0 = rotating frame off
 *   drag  : the coupled update needs scratch for the conserved state: p->gas.cons0 and
 *           p->dust.cons0 must be valid tables (their contents are overwritten)
 *   dt_dev: optional DEVICE scalar min-combined with cfl*min(...) of the new state
 * Bit-identical to the per-task sequence; p->{gas,dust}.prim are not used.
 * Curvilinear packs with one gas species run the gas on the tile march (kernels_curv.hip) also with dust, drag and
 * N-body gravity around it: the march leaves the gas conserved state in cons0 when drag follows, the dust runs on its
 * cell-centred kernel, and the drag finish couples them. */
typedef struct artemis_stage_general_args {
  double gam0, gam1, beta_dt, bdt;
  int pcm;
  double time;                /* tm.time at the start of the step (artemis_driver.cpp:167) */
  double *const *gas_in, *const *gas_u1, *const *gas_out;
  double *const *dust_in, *const *dust_u1, *const *dust_out;
  const artemis_gravity_t *gravity;
  double rf_omega, rf_qshear;
  const artemis_drag_t *drag;
  double cfl_gas, cfl_dust;
  double *dt_dev;
  const double *beta_dt_dev;  /* optional DEVICE scalar holding beta*dt of this stage; replaces the
                                 host values beta_dt and bdt (synchronisation-free time loop) */
  /* optional tasks folded into the same kernel (NULL = off).  diffusion: Gas::DiffusionUpdate from
   * p->gas.diff_flux, which the caller has filled for THIS stage's input primitives
   * (artemis_hip_zero_diffusion_flux / viscous_flux / thermal_flux on a pack whose gas.prim is
   * gas_in).  cooling: Gas::Cooling::CoolingSource (not together with drag).  A non-zero rf_omega on
   * a non-Cartesian pack selects RotatingFrameImpl, evaluated from the cell's own mass fluxes. */
  const artemis_diffusion_t *diffusion;
  const artemis_cooling_t *cooling;
  /* optional: the result of artemis_hip_viscous_source for THIS stage's input primitives (DEVICE table
   * [nblocks * 5]).  With it (and `diffusion` non-NULL) DiffusionUpdate subtracts these sums instead of forming them
   * from p->gas.diff_flux, which is then not read.  Same bits. */
  double *const *diffusion_sums;
  /* optional: Gravity::NBodyGravity (gravity/nbody_gravity.hpp:28-221) folded into the stage in the task's slot of
   * the list (after DiffusionUpdate, before RotatingFrameForce: artemis_driver.cpp:218-236) -- nbody_dev is a DEVICE
   * array of nbody_n particles, nbody_omf the frame rate its frame correction uses (0 = none); `gravity` must be
   * NULL with it (the reference has ONE gravity type).  The kernels add each coupled particle's acceleration and
   * accretion terms to the conserved state they hold, particle by particle, from the stage's input primitives: the
   * bits artemis_hip_nbody_gravity leaves.  The seven back-reaction sums per particle are NOT formed here: they depend
   * on the input primitives alone -- artemis_hip_nbody_force_sums.  Cartesian, cylindrical and spherical 3-D packs. */
  const artemis_nbody_particle_t *nbody_dev;
  int nbody_n;
  double nbody_omf;
  /* 1 = stop after the sources with the conserved state of the active zones in p->{gas,dust}.cons0 (no DragSource,
   * SetAuxillaryFields, ConsToPrim or dt): a refined mesh with drag redoes its listed coarse zones the same way
   * (artemis_hip_ml_stage_fixup with the same flag) and then runs artemis_hip_stage_finish on a pack whose prim tables
   * are the *_out tables.
   * 2 = the stage finishes EVERY zone itself, as with 0 (where one dust species is coupled by simple_dust drag the dust
   * march does it on its registers: no conserved round trip, no finish launch), and the caller -- whose
   * artemis_hip_ml_stage_fixup, given the same flag, leaves the conserved state of its LISTED zones in cons0 -- finishes
   * those zones again with artemis_hip_stage_finish_cells.  Same bits as 1: DragSource, SetAuxillaryFields and
   * ConsToPrim are pointwise (drag.hpp:296-482, fill_derived.cpp:58-164).  p->{gas,dust}.cons0 are required;
   * what they hold for zones outside the list afterwards is unspecified. */
  int defer_finish;
  /* Boundary conditions applied by the stage kernel itself, on the rows it loads (0 = none: the caller fills every ghost
   * zone before the call, the plain contract).  strat_faces = 15 (bits 0..3 = the inner / outer x1 and x2 faces): the
   * block's x1 faces carry the `strat` problem's `extrap` condition and its x2 faces `inflow` (pgen/strat.hpp:158-466,
   * ARTEMIS_BC_STRAT_EXTRAP / ARTEMIS_BC_STRAT_INFLOW of artemis_hip_apply_bc, with its qshear / omega) -- the kernel
   * then reads NO ghost zone of gas_in / dust_in and forms the conditions' values in registers, in parthenon's order (x1
   * over the entire x2 extent, then x2): the same bits as artemis_hip_apply_bc followed by the plain call.  The ghost zones
   * of the *_out arrays are left as they were; artemis_hip_apply_bc completes them whenever somebody else reads them.
   * Served by the 2-D row march only (artemis_hip_stage_general_variant == 1, one block): otherwise
   * ARTEMIS_HIP_EUNSUPPORTED. */
  int strat_faces;
  double strat_qshear, strat_omega;
} artemis_stage_general_args_t;
int artemis_hip_stage_general(const artemis_pack_t *p, const artemis_stage_general_args_t *a,
                              void *stream);
/* Which kernel artemis_hip_stage_general runs for this pack and these arguments (no launch, no device
 * access): 0 = the cell-centred kernels (one per fluid, + the drag finish), 1 = the 2-D row-march kernel
 * (kernels_stage2d.hip: both fluids, drag, aux, ConsToPrim and dt in one launch), 3 = the curvilinear tile march
 * (kernels_curv.hip: gas on any non-Cartesian system, geometry in LDS tables, two waves per SIMD; diffusion only as
 * diffusion_sums), 2 = its predecessor with the geometry in registers (kernels_fused.hip: taken when the diffusion
 * fluxes come from stored arrays).  Same results either way; benchmarks name the kernel they timed with this. */
int artemis_hip_stage_general_variant(const artemis_pack_t *p, const artemis_stage_general_args_t *a);
/* The cell-local remainder of a stage in ONE pass over stored fluxes: after Gas/Dust::CalculateFluxes
 * (and the diffusion-flux tasks) have filled flux / pflux / vface (/ diff_flux) for p's primitives,
 * this does ApplyUpdate on cons0 / cons1, FluxSource, DiffusionUpdate, ExternalGravity,
 * RotatingFrameForce, CoolingSource, SetAuxillaryFields and ConsToPrim (artemis_driver.cpp:205-255) and
 * writes the new primitives of the active zones in place.  Same argument struct (gam*, beta_dt, bdt,
 * time, gravity, rf_*, diffusion, cooling, beta_dt_dev; the prim tables and dt fields are ignored).
 * cons0 is NOT updated: the PrimToCons that follows the boundary conditions rebuilds it.  Drag couples
 * the fluids and is not part of it: ARTEMIS_HIP_EUNSUPPORTED, use the separate tasks. */
int artemis_hip_stage_epilogue(const artemis_pack_t *p, const artemis_stage_general_args_t *a, void *stream);
/* The same pass when something still has to act on the conserved state before SetAuxillaryFields: DragSource (it
 * couples the fluids) or NBodyGravity (a task of its own with a reduction the host reads).
 *   artemis_hip_stage_epilogue_cons: ApplyUpdate, FluxSource, DiffusionUpdate and -- when given -- ExternalGravity and
 *     RotatingFrameForce, in this order (artemis_driver.cpp:205-236), result left in cons0 (cons1 and the primitives
 *     untouched).  With N-body gravity pass gravity = NULL and rf_omega = 0 and run artemis_hip_nbody_gravity and
 *     artemis_hip_rotating_frame_force after it, as the reference orders them.  cooling / drag must be NULL.
 *   artemis_hip_stage_finish: DragSource (drag != NULL), SetAuxillaryFields and ConsToPrim of cons0 into the
 *     primitives of the active zones, one pass when one gas species is coupled by simple_dust drag.
 * Together they replace eight launches of the per-task chain by two or three; same bits. */
int artemis_hip_stage_epilogue_cons(const artemis_pack_t *p, const artemis_stage_general_args_t *a, void *stream);
int artemis_hip_stage_finish(const artemis_pack_t *p, const artemis_drag_t *drag, double time, double dt, void *stream);
/* The seven sums per particle of Gravity::NBodyGravity (mass-accretion rate, gravitational force, accreted-momentum
 * rate: nbody_gravity.hpp:190-215) of p's primitives, without touching the fluid: one pass over the pack (the task
 * kernel's force-only instantiation), then force_dev[7 n + q] += the workgroups' partial rows in index order -- the
 * additions artemis_hip_nbody_gravity makes on the host, on the device, so the stage loop needs no synchronisation.
 *   particles_dev : DEVICE array [npart]          dt / dt_dev : beta * dt of the stage (dt_dev replaces dt)
 *   scratch_dev   : DEVICE doubles, 7 * npart * artemis_hip_nbody_force_scratch(p) of them
 *   force_dev     : DEVICE accumulators [7 * npart], zeroed by the caller once
 * At most one species per fluid, npart <= 128. */
int artemis_hip_nbody_force_scratch(const artemis_pack_t *p);
int artemis_hip_nbody_force_sums(const artemis_pack_t *p, const artemis_nbody_particle_t *particles_dev, int npart, double omf,
                                 double dt, const double *dt_dev, double *scratch_dev, double *force_dev, void *stream);

/* ---- mesh-refinement data-path operators (the operators only) ---------------
 * ArtemisUtils::RestrictAverage<GEOM> (utils/refinement/restriction.hpp:42-114) and
 * ArtemisUtils::ProlongateSharedMinMod<GEOM> (utils/refinement/prolongation.hpp:83-184), the custom
 * refinement ops Artemis registers for its cell-centred fields, between a fine and a coarse array of one
 * block.  fgeom / cgeom = DEVICE edge tables {x1f0, dx1, x2f0, dx2, x3f0, dx3} of the fine / coarse index
 * space, fmetric / cmetric their metric tables (artemis_hip_metric_fill, NULL where not needed); fine /
 * coarse = DEVICE tables of nvar arrays.  Coarse zones [cis..cie] x [cjs..cje] x [cks..cke] are processed;
 * coarse index c?b coincides with fine index f?b (the reference's cib.s <-> ib.s).  Prolongation reads the
 * coarse neighbours at +-1 in every active direction.  The refinement framework around the operators
 * (block tree, flux correction, load balancing) is not built. */
typedef struct artemis_refine {
  int coords, ndim, nvar;
  int fni, fnj, fnk, cni, cnj, cnk;      /* array extents incl. ghosts */
  const double *fgeom, *fmetric, *cgeom, *cmetric;
  double *const *fine, *const *coarse;
  int cis, cie, cjs, cje, cks, cke;
  int cib, cjb, ckb, fib, fjb, fkb;
} artemis_refine_t;
int artemis_hip_restrict_average(const artemis_refine_t *r, void *stream);
int artemis_hip_prolongate_minmod(const artemis_refine_t *r, void *stream);

/* ---- multilevel (SMR / AMR) block-graph data path -------------------------------------------------
 * What Parthenon does around Artemis' tasks on a refined mesh, batched for the GPU: every mesh block of
 * the pack has the same shape but its own level (its own row of p->geom).  The host builds a list of
 * box-to-box operations once per mesh (artemis_amd/csrc/driver/block_tree.hpp) and uploads it; each call
 * below processes a whole list in ONE launch (one workgroup per operation).
 *   ghost exchange of the FillGhost primitives (AddBoundaryExchangeTasks with pmesh->multilevel,
 *   artemis_driver.cpp:258; upstream SendBoundBufs / SetBounds / ProlongateBounds, recalled):
 *     SAME          dst fine ghost box  <-  src fine interior                     (same-level neighbour)
 *     FROM_FINER    dst fine ghost box  <-  RestrictAverage<GEOM> of 2^ndim src fine zones per dst zone
 *                                           (restriction.hpp:42-114; the finer neighbour "sends restricted")
 *     FROM_COARSER  dst COARSE-BUFFER box <- src fine interior of the coarser neighbour
 *   flux correction (artemis_driver.cpp:196-202):
 *     FLUX          dst coarse-block face fluxes <- RestrictAverage<el = F_dir> (area-weighted, :57-94) of
 *                   the 2^(ndim-1) fine faces of the finer neighbour, for every WithFluxes variable (cons
 *                   D, M, E, e_int; the pressure flux, gas.cpp:212-252) and the Metadata::Flux diffusion
 *                   fluxes (gas.cpp:276-284); gas::face::velocity is not WithFluxes and is left alone
 * Source index of destination index i along dimension q: a_q * i + off[q] with a_q = 1 (SAME,
 * FROM_COARSER), 2 (FROM_FINER; FLUX tangential), 0 (FLUX normal: off = the fine face index).
 * A block index of -1 means "on another rank": the op then packs into sendbuf (dst remote) or unpacks
 * from recvbuf (src remote) at `buf`, slot layout [variable][n3][n2][n1] over the op's variables
 * (ghost ops: the 5 ns_gas + 4 ns_dust FillGhost primitives; FLUX: 6 ns_gas + ns_gas (+ 4 ns_gas with
 * diffusion fluxes) + 4 ns_dust). */
enum artemis_ml_kind { ARTEMIS_ML_SAME = 0, ARTEMIS_ML_FROM_FINER = 1, ARTEMIS_ML_FROM_COARSER = 2,
                       ARTEMIS_ML_FLUX = 3 };
typedef struct artemis_ml_op {
  int kind;                 /* artemis_ml_kind */
  int dst_block, src_block; /* indices into the pack, -1 = remote */
  int dir;                  /* FLUX: direction 0..2 of the faces */
  int lo[3], n[3];          /* destination box: first index and extent along x1, x2, x3 */
  int off[3];
  long buf;                 /* slot offset in doubles (remote ops) */
} artemis_ml_op_t;
typedef struct artemis_ml_box { /* a box of coarse-buffer zones of one block */
  int block, lo[3], n[3];
} artemis_ml_box_t;
typedef struct artemis_ml_pack {
  /* coarse buffers: one array of (nx/2 + 2 nghost) zones per active dimension for every variable of the
   * prim tables, laid out like them ([nblocks * 6 ns_gas], [nblocks * 4 ns_dust]) */
  double *const *gas_coarse, *const *dust_coarse;
  const double *cgeom;   /* DEVICE [nblocks][6] edge table of the coarse buffers' index space */
  const double *cmetric; /* their metric tables (artemis_hip_metric_fill on the coarse shape), or NULL */
} artemis_ml_pack_t;
/* ops_dev: DEVICE array of nops operations.  sendbuf / recvbuf: DEVICE message buffers (NULL when no op is
 * remote).  The FillGhost primitives are read from / written to p->gas.prim, p->dust.prim. */
int artemis_hip_ml_exchange(const artemis_pack_t *p, const artemis_ml_pack_t *ml, const artemis_ml_op_t *ops_dev,
                            int nops, double *sendbuf, const double *recvbuf, void *stream);
/* FLUX ops on p->{gas,dust}.flux / pflux / diff_flux (ml may be NULL) */
int artemis_hip_ml_flux_correction(const artemis_pack_t *p, const artemis_ml_op_t *ops_dev, int nops, double *sendbuf,
                                   const double *recvbuf, void *stream);
/* RestrictAverage of blocks_dev[0..nblocks)'s own fine arrays (interior + nghost/2 coarse zones of ghost halo
 * all round) into their coarse buffers: gives ProlongateSharedMinMod its +-1 stencil next to the zones that
 * came from a coarser neighbour (upstream: restriction of the ghost halos inside ProlongateBounds). */
int artemis_hip_ml_restrict_halos(const artemis_pack_t *p, const artemis_ml_pack_t *ml, const int *blocks_dev, int nblocks,
                                  void *stream);
/* ProlongateSharedMinMod<GEOM> (prolongation.hpp:83-184) from the coarse buffer into the fine ghost zones
 * covered by each box (boxes_dev: DEVICE array; coarse index cs + q <-> fine index s + 2 q). */
int artemis_hip_ml_prolongate(const artemis_pack_t *p, const artemis_ml_pack_t *ml, const artemis_ml_box_t *boxes_dev,
                              int nboxes, void *stream);
/* The primitive floors of the PrimToCons that follows the boundary fill in the reference (fill_derived.cpp:227, :245,
 * :262: gas density, gas sie, dust density) on every ghost zone of blocks_dev[0..nblocks).  A caller that runs
 * artemis_hip_prim_to_cons over the whole block after the fill does not need it; the one-kernel stages keep no conserved
 * ghost zones and call this instead, last of the fill, for the blocks that took restricted or prolongated values: a
 * restricted average can round an ulp below a floor all its zones sit on, the three limited slopes of a prolongation can
 * add up to less than the lowest neighbour.  Zones above the floors are read, not written. */
int artemis_hip_ml_floor_ghosts(const artemis_pack_t *p, const int *blocks_dev, int nblocks, void *stream);

/* ---- one-kernel stages on a refined mesh: flux correction as a thin fix-up ------------------------
 * The reference corrects the coarse side of every coarse-fine face between CalculateFluxes and ApplyUpdate
 * (artemis_driver.cpp:196-202), which is why its stage needs the flux arrays.  Only the coarse zones that touch such
 * a face see the corrected flux, so a refined mesh can run the one-kernel stages (artemis_hip_stage_fused,
 * artemis_hip_stage_general: no flux arrays) on every block and then redo exactly those zones:
 *   1. artemis_hip_stage_fused / artemis_hip_stage_general on the whole pack (prim_in -> prim_out);
 *   2. artemis_hip_ml_face_fluxes: the fine side's faces on the coarse-fine boundaries, from prim_in, into the
 *      flux / pflux arrays of the pack (only those entries) -- what CalculateFluxes leaves there;
 *   3. artemis_hip_ml_flux_correction (and the messages between ranks) exactly as on the per-task chain: the
 *      restricted fine fluxes land in the coarse blocks' flux / pflux / diff_flux entries of those faces;
 *   4. artemis_hip_ml_stage_fixup: the listed coarse zones again, the whole stage from prim_in with the same device
 *      functions as artemis_hip_stage_general's cell-centred form, except that the faces flagged in `faces` take
 *      their mass / momentum / energy / pressure fluxes from the arrays step 3 wrote (the face velocity is not a
 *      flux field and stays the zone's own, as in the reference) -- the same sum, in the same order, as
 *      ApplyUpdate after SetFluxCorrections; results overwrite prim_out at those zones.
 * Diffusion fluxes: the caller computed them for the whole pack before step 1 (artemis_hip_zero_viscous_flux);
 * step 3 corrects them in place and step 4 reads them like the stage kernels do.  args: the arguments of step 1
 * (dt_dev is ignored: estimate the timestep after the fix-up).  Drag couples the fluids after the update: see
 * artemis_hip_stage_finish_cells below. */
typedef struct artemis_ml_face_box { /* faces of direction dir stored at the zones of a box of one block */
  int block, dir, lo[3], n[3];
} artemis_ml_face_box_t;
typedef struct artemis_ml_fix_cell {
  int block, k, j, i;
  unsigned faces; /* bit 2 d + side: the lower (side 0) / upper (side 1) face of direction d is a corrected one */
} artemis_ml_fix_cell_t;
int artemis_hip_ml_face_fluxes(const artemis_pack_t *p, const artemis_stage_general_args_t *a,
                               const artemis_ml_face_box_t *boxes_dev, int nboxes, void *stream);
/* The same for the viscous fluxes when step 1 ran on artemis_hip_viscous_source's sums (no diffusion-flux array was
 * filled for the pack): Gas::ZeroDiffusionFlux + ViscousFlux evaluated on exactly the faces the flux correction
 * touches -- the faces of the boxes (the fine side of the coarse-fine boundaries, as above) and the 2 ndim faces of
 * every listed zone (what artemis_hip_ml_stage_fixup reads) -- from p->gas.prim (the stage's input primitives) into
 * those entries of p->gas.diff_flux: the same bits the whole-pack tasks would leave there.  Then steps 3 and 4 as
 * before, step 4 with diffusion_sums = NULL so that it forms the listed zones' sums from the corrected arrays.
 * One gas species; viscosity only. */
int artemis_hip_ml_viscous_faces(const artemis_pack_t *p, const artemis_diffusion_t *d, const artemis_ml_face_box_t *boxes_dev,
                                 int nboxes, const artemis_ml_fix_cell_t *cells_dev, int ncells, void *stream);
int artemis_hip_ml_stage_fixup(const artemis_pack_t *p, const artemis_stage_general_args_t *a,
                               const artemis_ml_fix_cell_t *cells_dev, int ncells, void *stream);
/* Drag on a refined mesh (DragSource follows the flux correction in the task list, artemis_driver.cpp:196-255): run steps
 * 1 and 4 with args.defer_finish = 2 -- step 1 then finishes every zone itself, step 4 leaves the conserved state of the
 * listed zones in p->{gas,dust}.cons0 -- and finish the listed zones with
 *   artemis_hip_stage_finish_cells: DragSource (drag != NULL), SetAuxillaryFields and ConsToPrim of cons0 into the
 *     primitives of p (point its prim tables at the *_out tables), for the listed zones only; `faces` is ignored.
 * (defer_finish = 1 is the older form: steps 1 and 4 stop at cons0 and artemis_hip_stage_finish runs over every zone.) */
int artemis_hip_stage_finish_cells(const artemis_pack_t *p, const artemis_drag_t *drag, double time, double dt,
                                   const artemis_ml_fix_cell_t *cells_dev, int ncells, void *stream);

/* ---- refinement criteria (utils/refinement/amr_criteria.hpp) ------------------------------------
 * ArtemisUtils::ScalarFirstDerivative<FIELD, GEOM> (:28-132) and ScalarMagnitude<FIELD> (:137-168): the
 * block-wide maximum that decides a mesh block's AmrTag, for one cell-centred scalar (the first
 * component of FIELD: gas density or pressure in the reference's gas package, gas.cpp refine_field).
 * first_derivative: max over the interior grown by one zone in every active direction of
 *   |grad q| / (q / stencil diagonal), centred differences over the cell-centre spacing and the
 *   volume-averaged scale factors; ndim == 1 returns "same" without computing (as the reference does).
 *   tag = +1 (refine) when the maximum exceeds refine_thr, -1 (derefine) below 0.25 * refine_thr.
 * magnitude: max of q over the interior; +1 above refine_thr, -1 below deref_thr.
 * scratch = one DEVICE double (overwritten).  Both calls synchronise `stream` (the reference's
 * par_reduce does) and write *tag (AmrTag: -1 derefine, 0 same, +1 refine) and, if non-NULL, *maxval. */
typedef struct artemis_amr_criterion {
  int coords, ndim;
  int ni, nj, nk;                     /* array extents incl. ghosts (>= 2 ghost zones for the derivative) */
  const double *geom, *metric;        /* as artemis_refine_t's fgeom / fmetric */
  const double *field;                /* DEVICE array [nk][nj][ni] */
  int is, ie, js, je, ks, ke;         /* interior bounds */
  double refine_thr, deref_thr;
  double *scratch;
} artemis_amr_criterion_t;
int artemis_hip_amr_first_derivative(const artemis_amr_criterion_t *a, int *tag, double *maxval, void *stream);
int artemis_hip_amr_magnitude(const artemis_amr_criterion_t *a, int *tag, double *maxval, void *stream);
/* The same block maxima for EVERY block of a pack in one launch: field = 0 (gas density of species 0), 1 (gas
 * pressure of species 0, read from the primitives' pressure slot) or 2 (the same pressure recomputed as
 * max(0, (gamma - 1) rho sie), fill_derived.cpp:247 -- for callers whose pressure slot is not maintained in the
 * ghost zones: the one-kernel stages), magnitude = 0 (ScalarFirstDerivative) or 1 (ScalarMagnitude); maxima_dev = DEVICE array of
 * p->nblocks doubles.  Asynchronous on `stream` (copy the maxima back and compare with the thresholds as above):
 * an adaptive mesh of thousands of small blocks is tagged with one launch and one copy instead of a launch and a
 * synchronisation per block. */
int artemis_hip_amr_block_maxima(const artemis_pack_t *p, int field, int magnitude, double *maxima_dev, void *stream);

/* Device-side SetGlobalTimeStep (parthenon EvolutionDriver, upstream): state = DEVICE
 * {time, dt, dt_est, beta_dt[0..2]}.  time += dt; dt = min(2*dt, dt_est), clipped so that
 * time + dt <= tlim (tlim <= 0: no limit); dt_est = DBL_MAX for the next cycle's reduction;
 * beta_dt[s] = beta[s]*dt for the nstages (<= 3) stage weights in `beta` (HOST array). */
int artemis_hip_advance_dt(double *state, double tlim, int nstages, const double *beta, void *stream);

/* Enqueue on `stream` a one-wave kernel that returns once *counter >= target (agent-scope
 * poll + acquire).  Work queued behind it on that stream sees everything the counting
 * workgroups published.  `timeout_flag` (DEVICE, optional) is set to 1 if the poll gives up
 * after ~seconds instead of hanging the GPU. */
int artemis_hip_wait_counter(unsigned *counter, unsigned target, unsigned *timeout_flag, void *stream);

/* ---- Halo slabs (inter-block / inter-GPU ghost exchange of FillGhost primitives) -------
 * Pack the `nghost`-deep interior slab adjacent to face `face` (0..5 = ix1,ox1,ix2,ox2,
 * ix3,ox3) of block `block` into the contiguous DEVICE buffer `buf`
 * ([nfill][slab cells], nfill = 5*ns_gas + 4*ns_dust), or unpack such a buffer into the
 * ghost slab behind that face.  Slabs span the interior extent of the other dimensions
 * (the hydro stencil never reads edge/corner ghosts: fluid_fluxes.hpp:105-106,130-131,
 * 172-173).  artemis_hip_halo_count returns the number of doubles in one slab buffer. */
long artemis_hip_halo_count(const artemis_pack_t *p, int face);
/* `_ext` variants with extended != 0: the slab spans the ENTIRE extent (ghost zones included) of the
 * dimensions below the face's own, so that exchanging x1 slabs, then x2, then x3 also fills the
 * edge and corner ghost zones (Parthenon fills them from the diagonal neighbours; only the viscous
 * cross-derivatives read them, momentum_diffusion.hpp:95-141).  extended == 0: as above. */
long artemis_hip_halo_count_ext(const artemis_pack_t *p, int face, int extended);
int artemis_hip_halo_pack_ext(const artemis_pack_t *p, int block, int face, int extended, double *buf,
                              void *stream);
int artemis_hip_halo_unpack_ext(const artemis_pack_t *p, int block, int face, int extended,
                                const double *buf, void *stream);
int artemis_hip_halo_pack(const artemis_pack_t *p, int block, int face, double *buf, void *stream);
int artemis_hip_halo_unpack(const artemis_pack_t *p, int block, int face, const double *buf,
                            void *stream);

/* Self test (device): q_fast/s_fast from the hand-scheduled division / square root of the fused
 * kernel (device_math.hpp), q_ieee/s_ieee from the compiler's IEEE-correct a/b and sqrt(|b|).
 * All pointers are DEVICE arrays of n doubles. */
int artemis_hip_selftest_divsqrt(long n, const double *a, const double *b, double *q_fast,
                                 double *q_ieee, double *s_fast, double *s_ieee, void *stream);

/* ---- Library state ---------------------------------------------------------------------*/
const char *artemis_hip_last_error(void);
/* The library's switches (list above): set one at run time (ARTEMIS_HIP_EINVAL: unknown name) / read it back (-1: unknown).  Not thread-safe against
 * launches in flight on other threads: set options before the work they steer. */
int artemis_hip_set_option(const char *name, long value);
long artemis_hip_get_option(const char *name);
/* Number of visible HIP devices (0 = none; every compute entry point then fails loudly with
 * ARTEMIS_HIP_EDEVICE -- there is no CPU fallback in this library). */
int artemis_hip_device_count(void);
const char *artemis_hip_version(void);
/* Identity of the sources this library was built from (artemis_amd/build.py generates the table at build time):
 * artemis_hip_source_sha() = sha1 over every file of csrc/ and include/ plus the compiler flags;
 * artemis_hip_object_sha("kernels_fused") = sha256 over that translation unit, the headers it reaches, its command line
 * and the compiler version (NULL: no such unit).  A measurement record (profiles/) names the identity it was taken on
 * and is quoted only by a library that reports the same one. */
const char *artemis_hip_source_sha(void);
const char *artemis_hip_object_sha(const char *unit);

#ifdef __cplusplus
}
#endif
#endif /* ARTEMIS_HIP_H_ */
