// Fused RK stage for gas (one species) on the CURVILINEAR coordinate systems (cylindrical, spherical 1-D / 2-D / 3-D,
// axisymmetric), PCM / PLM_G, HLLC / HLLE / LLF:
//   CalculateFluxes (PLM_G, ScaleMomentumFlux) -> ApplyUpdate -> FluxSource (pressure + coordinate sources with the
//   frame velocity) -> DiffusionUpdate (from artemis_hip_viscous_source's sums) -> ExternalGravity -> RotatingFrameImpl
//   -> SetAuxillaryFields -> ConsToPrim (-> EstimateTimestepMesh)
// (artemis_driver.cpp:182-255; plm.hpp:54-73, fluid_fluxes.hpp:33-70, :323-417) in ONE pass: the 2.5-D tile march of
// kernels_fused.hip -- a 256-thread workgroup owns an FTX x FTY column of zones and marches along x3, rolling window,
// carried x3 face state and flux in registers, x2 exchange through LDS, two barriers per plane, perimeter duties
// rotating over the waves -- rebuilt around the one thing that kept its curvilinear instantiation at one wave per SIMD:
//
//   * GEOMETRY LIVES IN LDS, NOT IN REGISTERS.  In every system the metric depends on (x1, x2) only, so a thread's
//     Coords-derived constants are constants of the march -- and the compiler keeps ~100 doubles of them per thread
//     (256 VGPR + 190 AGPR in round 3).  Here a workgroup tabulates, once, per COLUMN index the x1 edges and every
//     x1-only expression with a division in it, per ROW index the x2 edges, the trigonometry and the x2-only quotients
//     (geometry.hpp GeoTabs), and PLM_G's geometric weights per column / row / plane; a thread rebuilds the Coords of
//     any zone of its tile -- own, halo, perimeter -- from one column and one row entry when it needs them and lets
//     DCoordsT<true> do the remaining multiplications.  LDS loads sit between barriers, so nothing is hoisted out of
//     the plane loop.
//   * THE COORDINATE SYSTEM IS A TEMPLATE CONSTANT: every switch inside Coords folds (the per-task kernels keep the
//     run-time switch -- they wait for memory; this kernel waits for its own instructions).
//   * x1 NEIGHBOURS THROUGH DPP: the upper face value and the lower-face flux travel to the lane next door with
//     wave shifts (two VALU moves per double, no LDS round trip); only the tile's two edge columns go through small
//     LDS arrays filled by the perimeter duties.  That frees 29 KB of LDS for the tables.
//   * FLUXES ARE FOLDED INTO THE UPDATE'S SUMS DIRECTION BY DIRECTION (the sums ApplyUpdate, FluxSource and
//     RotatingFrameImpl form, in their order of additions), so at most one direction's face records are alive.
// Result: <= 256 registers, two workgroups per CU, no scratch.  Same device functions and expression trees as the
// cell-centred general stage and the per-task chain: bit-identical (tests/test_parity_stage_general.py).
#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "device_math.hpp"
#include "diffusion_device.hpp"
#include "fused_device.hpp"
#include "geometry.hpp"
#include "kernels.hpp"
#include "options.hpp"
#include "pack_view.hpp"
#include "sources_device.hpp"
#include "nbody_device.hpp"
#include "task_device.hpp"

// -DCURV_PROF (development builds only: ARTEMIS_HIPFLAGS_KERNELS_CURV=-DCURV_PROF): every wave sums the shader-clock
// cycles it spends in each phase of a plane (work and waiting alike) into g_curv_prof; artemis_hip_debug_curv_prof reads them
#ifdef CURV_PROF
__device__ unsigned long long g_curv_prof[16];
#define PROF_DECL unsigned long long prof_t = __builtin_readcyclecounter(), prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PROF(slot)                                              \
  do {                                                          \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    prof_acc[slot] += now_ - prof_t;                            \
    prof_t = now_;                                              \
  } while (0)
#else
#define PROF_DECL
#define PROF(slot)
#endif

namespace artemis {
namespace {
using namespace fused;

constexpr int CKMAX = 64; // planes per chunk at most (the x3 weights of a chunk's planes sit in LDS)

struct CurvK {
  double gam0, gam1, beta_dt, bdt, cfl;
  const double *bdt_ptr;
  double *const *prim_in, *const *prim_u1, *const *prim_out;
  unsigned long long *dt_bits;
  int has_u1;
  int nti, ntj, nchunk, kchunk;
  int grav_on, rfc_on, diff_on;
  double rf_omega;
  artemis_gravity_t grav;
  double *const *dsum;
  int to_cons;                          // 1: stop after the sources, conserved state to P.gas.cons0 (drag follows)
  int finish;                           // DUST march: DragSource (simple_dust) + SetAuxillaryFields + ConsToPrim of BOTH fluids here
  DragLaw1 drag;                        //   (the gas march has left its conserved state in P.gas.cons0)
  double *const *gas_out;               //   the gas primitives' output tables
  double cfl_gas;                       //   (the gas fluid's timestep limit is formed here too)
  const artemis_nbody_particle_t *nb_pl; // N-body gravity in the gravity task's slot (device array), nb_n particles
  int nb_n;
  double nb_omf;
};

enum { PW_CR = 0, PW_CL, PW_UP, PW_LO, PW_RAB, PW_RAY, PW_RBB, PW_RBY, PW_NF, // PLM_G weights that depend on one index only
       PW_DX = PW_NF, PW_RDB, PW_RDY, PW1_NF };                               // + the cell width along x1 (x1 records)

// Workgroup constants the update reads on every plane.  As kernel arguments they sit in scalar registers for the whole
// march -- with ~20 array pointers, the pack's bounds and the solver constants that is more than the 100-odd a wave has,
// so the compiler parks them in VGPR lanes (v_writelane / v_readlane around every use) and, at 256 VGPRs, in scratch.
// One thread copies them to LDS once; a use is a broadcast ds_read next to the instruction that consumes it.
struct CurvConst {
  double gam0, gam1, beta_dt, bdt, cfl, rf_omega, omf, nb_omf;
  double dfloor, siefloor, de_switch;
  artemis_gravity_t grav;
  // the block's output / viscous-sum / conserved arrays.  Read from the pointer tables inside the march they would be
  // vector loads (the kernel has stores in flight: no scalar load) followed by vmcnt(0) -- a wait for every prefetch
  double *out[6], *cons0[6];
  const double *dsum[5], *u1[5]; // (u1: rho, v1, v2, v3, sie of the start-of-step state)
  // DUST march with the drag finish (CurvK.finish): the gas conserved state it reads, the gas primitives it writes
  const double *gcons[6];
  double *gout[5]; // rho, v1, v2, v3, sie
  double g_dfloor, g_siefloor, g_de_switch, cfl_gas;
  DragLaw1 drag;
};
// artemis_gravity_t as gravity_accel reads it: the law's type from the kernel argument (a scalar: the branches on it stay
// scalar branches), every number from the LDS copy
struct GravLds {
  const int type;
  const artemis_gravity_t &L;
  const double (&g)[3] = L.g, (&pos)[3] = L.pos, (&pos2)[3] = L.pos2;
  const double &gm = L.gm, &soft = L.soft, &sink = L.sink, &sink_rate = L.sink_rate, &q = L.q, &soft2 = L.soft2,
               &sink2 = L.sink2, &sink_rate2 = L.sink_rate2;
};

template <int FTX_>
struct CurvTile {
  static constexpr int FTX = FTX_, FTY = 256 / FTX_, QX = FTX + 4, QY = FTY + 4;
  double Q[6][QY][QX];          // staged primitives of plane k (rho, v1, v2, v3, P, sie), halo 2 (no corners)
  double UPY[6][FTY + 1][FTX];  // upper x2-face value of rows j0-1 .. j0+FTY-1
  double LOY[6][FTX];           // lower x2-face value of row j0+FTY
  double UPX0[6][FTY];          // upper x1-face value of column i0-1      (perimeter duty -> lanes tx == 0)
  double UPXE[6][FTY];          // upper x1-face value of column i0+FTX-1  (lanes tx == FTX-1 -> perimeter duty)
  double LOXE[6][FTY];          // lower x1-face value of column i0+FTX    (perimeter duty)
  double FY[8][FTY][FTX];       // x2 faces j0+1 .. j0+FTY (upper faces of the tile's rows)
  double FXE[8][FTY];           // x1 face i0+FTX (perimeter duty -> lanes tx == FTX-1)
  GeoTabs<QX, QY> G;            // columns i0-2 .. i0+FTX+1, rows j0-2 .. j0+FTY+1
  double PX1[PW1_NF][QX];       // PLM_G records along x1, per column
  double PX2[PW_NF][QY];        // PLM_G weights along x2, per row (the width is (i, j): formed where it is used)
  double PX3[PW_NF][CKMAX + 4]; // ... along x3, planes k0-1 .. k1+1
  double ZL[6][256];            // the zone's upper x3 face value, parked between two x3 sweeps (12 VGPRs less through a plane)
  double C3[CKMAX], S3[CKMAX];  // cos / sin of the x3 centres of the chunk's planes (spherical3D, axisymmetric; else 1 / 0)
  int tiny[2];                  // plane (k & 1) holds a tiny-but-nonzero velocity: its slopes take IEEE division
  double wmin[4];
  CurvConst C;
};
static_assert(sizeof(CurvTile<32>) <= 80 * 1024 && sizeof(CurvTile<16>) <= 80 * 1024, "two workgroups per CU");

struct PlmG { // what plm_g_shared reads
  double dx, cr, cl, up, lo;
  Recip ra, rb, rdx;
};
// Riemann problem of sweep direction DIR for the kernel's fluid: the gas solvers of fused_device.hpp, or Dust's HLLE / LLF
// (dust/riemann: task_device.hpp riemann_dust) with the energy, pressure-flux and face-velocity slots left at zero
template <bool DUST, int RIEMANN, int DIR>
ADEV Flux8 solve_fluid(const GasK &gk, const Cell6 &L, const Cell6 &R, const bool fast) {
  if constexpr (!DUST) {
    return solve_face<RIEMANN, DIR>(gk, L, R, fast);
  } else {
    Prim4 l, r;
    l.d = L.d, r.d = R.d;
    if constexpr (DIR == 1) l.vx = L.v1, l.vy = L.v2, l.vz = L.v3, r.vx = R.v1, r.vy = R.v2, r.vz = R.v3;
    else if constexpr (DIR == 2) l.vx = L.v2, l.vy = L.v3, l.vz = L.v1, r.vx = R.v2, r.vy = R.v3, r.vz = R.v1;
    else l.vx = L.v3, l.vy = L.v1, l.vz = L.v2, r.vx = R.v3, r.vy = R.v1, r.vz = R.v2;
    FaceFlux F;
    if constexpr (RIEMANN == 2) llf_dust(l, r, F); // (no division, no root)
    else if (fast) hlle_dust_fast(l, r, F);
    else hlle_dust(l, r, F);
    Flux8 o;
    o.d = F.fd, o.e = o.eg = o.pf = o.vf = 0.0;
    if constexpr (DIR == 1) o.m1 = F.fmx, o.m2 = F.fmy, o.m3 = F.fmz;
    else if constexpr (DIR == 2) o.m2 = F.fmx, o.m3 = F.fmy, o.m1 = F.fmz;
    else o.m3 = F.fmx, o.m1 = F.fmy, o.m2 = F.fmz;
    return o;
  }
}
// The dust instantiation (DUST) carries rho, v1, v2, v3 only: the pressure / energy slots of the staged cell and the
// energy / pressure-flux / face-velocity slots of a face flux are skipped (every X body below is guarded by its index).
#define CFOR6(X) X(d, 0) X(v1, 1) X(v2, 2) X(v3, 3) X(p, 4) X(e, 5)
// ... with a scheduling fence after the third variable: three slope chains interleave (six would need the registers of
// twelve more doubles)
#define CFOR6_33(X) X(d, 0) X(v1, 1) X(v2, 2) __builtin_amdgcn_sched_barrier(0); X(v3, 3) X(p, 4) X(e, 5)
#define CGET6(dst, A, ...)                                                                 \
  dst.d = A[0] __VA_ARGS__, dst.v1 = A[1] __VA_ARGS__, dst.v2 = A[2] __VA_ARGS__,          \
  dst.v3 = A[3] __VA_ARGS__;                                                               \
  if constexpr (!DUST) dst.p = A[4] __VA_ARGS__, dst.e = A[5] __VA_ARGS__
#define CPUT8(A, fl, ...)                                                                  \
  A[0] __VA_ARGS__ = fl.d, A[1] __VA_ARGS__ = fl.m1, A[2] __VA_ARGS__ = fl.m2,             \
  A[3] __VA_ARGS__ = fl.m3;                                                                \
  if constexpr (!DUST)                                                                     \
  A[4] __VA_ARGS__ = fl.e, A[5] __VA_ARGS__ = fl.eg, A[6] __VA_ARGS__ = fl.pf, A[7] __VA_ARGS__ = fl.vf
#define CGET8(fl, A, ...)                                                                  \
  fl.d = A[0] __VA_ARGS__, fl.m1 = A[1] __VA_ARGS__, fl.m2 = A[2] __VA_ARGS__,             \
  fl.m3 = A[3] __VA_ARGS__;                                                                \
  if constexpr (!DUST)                                                                     \
  fl.e = A[4] __VA_ARGS__, fl.eg = A[5] __VA_ARGS__, fl.pf = A[6] __VA_ARGS__, fl.vf = A[7] __VA_ARGS__

// the value held by the lane below / above (kernels_stage2d.hip: wave_shr:1 / wave_shl:1, one move per dword)
ADEV double lane_below(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false));
}
ADEV double lane_above(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false));
}

// Workgroup `id` of the launch: tile (ti, tj), chunk and block (ids dealt so that each XCD's L2 sees one run of tiles)
// EXT: the instantiations that stop at the conserved state (drag follows) and / or carry N-body gravity
// DUST: the march of the dust species (rho, v: Dust::CalculateFluxes' HLLE / LLF, Dust::FluxSource's coordinate source,
// gravity / N-body / rotating frame on four conserved variables, Dust's ConsToPrim and timestep)
template <int SYS, int RIEMANN, int RECON, bool D3, int FTX, bool EXT = false, bool DUST = false>
__global__ __launch_bounds__(256, 2) void stage_curv_kernel(const PackView P, const CurvK a) {
  using T = CurvTile<FTX>;
  constexpr int FTY = T::FTY, QX = T::QX, QY = T::QY, FH = 2;
  constexpr int NV = DUST ? 4 : 6; // staged variables; also the stride of the fluid's pointer tables (one species)
  constexpr bool PG = (RECON == 1);                       // a limited slope (PLM): the tiny-velocity guard applies
  constexpr bool PGG = PG && SYS != ARTEMIS_CARTESIAN;    // PLM_G: the slope takes geometric weights (plm.hpp:54-73)
  __shared__ T S;
  const int t = threadIdx.x, tx = t % FTX, ty = t / FTX;
  int id = blockIdx.x;
  {
    const int n = static_cast<int>(gridDim.x), q = n >> 3, rem = n & 7, xcd = id & 7;
    id = xcd * q + min(xcd, rem) + (id >> 3);
  }
  const int ti = id % a.nti;
  id /= a.nti;
  const int tj = id % a.ntj;
  id /= a.ntj;
  const int chunk = id % a.nchunk, b = id / a.nchunk;
  const bool multi_d = D3 || P.ndim > 1;
  const int i0 = P.is + ti * FTX, j0 = P.js + tj * FTY;
  const int i = i0 + tx, j = j0 + ty;
  const bool active = (i <= P.ie) && (j <= P.je);
  const int il = min(i, P.ni - 1), jl = min(j, P.nj - 1);
  const int k0 = D3 ? P.ks + chunk * a.kchunk : P.ks;
  const int k1 = D3 ? min(P.ke, k0 + a.kchunk - 1) : P.ks;
  const double gm1 = P.gm1;
  const GasK gk = gas_constants(gm1);
  const FluidView &f = DUST ? P.dust : P.gas;
  if (t == 0) { // (read back after the barrier that follows the geometry tables)
    CurvConst c;
    c.gam0 = a.gam0, c.gam1 = a.gam1, c.beta_dt = a.beta_dt, c.bdt = a.bdt, c.cfl = a.cfl, c.rf_omega = a.rf_omega;
    if (a.bdt_ptr) c.beta_dt = c.bdt = *a.bdt_ptr;
    c.omf = P.omf, c.nb_omf = a.nb_omf;
    c.dfloor = f.dfloor, c.siefloor = f.siefloor, c.de_switch = f.de_switch;
    c.grav = a.grav;
    for (int q = 0; q < NV; ++q) c.out[q] = a.prim_out[b * NV + q], c.cons0[q] = a.to_cons ? f.cons0[b * NV + q] : nullptr;
    for (int q = 0; q < 5; ++q) c.dsum[q] = (!DUST && a.diff_on) ? a.dsum[b * 5 + q] : nullptr;
    for (int q = 0; q < 4; ++q) c.u1[q] = a.prim_u1[b * NV + q];
    c.u1[4] = a.prim_u1[b * NV + (DUST ? 0 : NV - 1)]; // (dust: never loaded)
    S.C = c;
    if constexpr (DUST) {
      if (a.finish) { // (field by field, straight to LDS: an aggregate copy under this condition would live in scratch)
#pragma unroll
        for (int q = 0; q < 6; ++q) S.C.gcons[q] = P.gas.cons0[b * 6 + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) S.C.gout[q] = a.gas_out[b * 6 + q];
        S.C.gout[4] = a.gas_out[b * 6 + 5];
        S.C.g_dfloor = P.gas.dfloor, S.C.g_siefloor = P.gas.siefloor, S.C.g_de_switch = P.gas.de_switch, S.C.cfl_gas = a.cfl_gas;
        S.C.drag.stokes = a.drag.stokes, S.C.drag.tau = a.drag.tau, S.C.drag.scale = a.drag.scale;
        S.C.drag.grain_density = a.drag.grain_density, S.C.drag.size = a.drag.size;
      }
    }
  }
  const double *g = P.geom + 6 * b;
  const double *in_r = a.prim_in[b * NV + 0], *in_1 = a.prim_in[b * NV + 1], *in_2 = a.prim_in[b * NV + 2];
  const double *in_3 = a.prim_in[b * NV + 3], *in_e = DUST ? in_r : a.prim_in[b * NV + (NV - 1)]; // (dust: never loaded)
  const unsigned sj = static_cast<unsigned>(P.sj), sk = static_cast<unsigned>(P.sk);
  const unsigned col = static_cast<unsigned>(jl) * sj + static_cast<unsigned>(il);
  auto ldraw = [&](const double *r_, const double *v1_, const double *v2_, const double *v3_, const double *e_, unsigned c_) {
    Raw5 q;
    q.d = gld(r_, c_), q.v1 = gld(v1_, c_), q.v2 = gld(v2_, c_), q.v3 = gld(v3_, c_);
    if constexpr (DUST) q.e = 0.0;
    else q.e = gld(e_, c_);
    return q;
  };
  auto ldcell = [&](const double *r_, const double *v1_, const double *v2_, const double *v3_, const double *e_, unsigned c_) {
    return finish_cell(ldraw(r_, v1_, v2_, v3_, e_, c_), gm1);
  };
  // cos / sin of the x3 cell centres (spherical3D, axisymmetric): ConvertCoordsToCart of the gravity task
  const double *m3 = nullptr;
  if ((SYS == ARTEMIS_SPHERICAL3D || SYS == ARTEMIS_AXISYMMETRIC) && P.metric)
    m3 = P.metric + b * metric_block_stride(P.nj, P.nk) + static_cast<long>(MT_ROWS) * (P.nj + 1);
  const double *mrow = P.metric ? P.metric + b * metric_block_stride(P.nj, P.nk) : nullptr;
  // halo duty: threads 0 .. 4 FTX - 1 stage the x2 halo rows (Q rows 0, 1, FTY+2, FTY+3), the next 4 FTY threads the
  // x1 halo columns (Q columns 0, 1, FTX+2, FTX+3); each owns one halo column for the whole march
  int hr = -1, hc = -1;
  if (t < 4 * FTX) {
    const int rr = t / FTX;
    hr = (rr < 2) ? rr : FTY + rr, hc = (t % FTX) + FH;
  } else if (t < 4 * FTX + 4 * FTY) {
    const int u = t - 4 * FTX, cc = u & 3;
    hr = (u >> 2) + FH, hc = (cc < 2) ? cc : FTX + cc;
  }
  unsigned hcol = col; // (threads without a halo duty: their own column)
  if (hr >= 0) {
    const int gi = min(max(i0 - FH + hc, 0), P.ni - 1), gj = min(max(j0 - FH + hr, 0), P.nj - 1);
    hcol = static_cast<unsigned>(gj) * sj + static_cast<unsigned>(gi);
  }
  // ---- the workgroup's geometry tables -----------------------------------------------------------------------------
  geotabs_fill(S.G, P, b, i0 - FH, j0 - FH, t);
  if constexpr (PGG) {
    auto coords = [&](int kk, int jj, int ii) { return coords_of(SYS, g, mrow, P.nj, P.nk, kk, jj, ii); };
    if (t >= 128 && t < 128 + QX) { // x1 records, columns i0-2 .. i0+FTX+1 (those next to the array's ends are never read)
      const int x = t - 128, ii = min(max(i0 - FH + x, 1), P.ni - 2);
      PlmGeo r;
      const DCoords c = coords(0, 0, ii);
      r.xvm = coords(0, 0, ii - 1).x1v(), r.xvc = c.x1v(), r.xvp = coords(0, 0, ii + 1).x1v();
      r.xf0 = c.x1[0], r.xf1 = c.x1[1], r.dx = c.width1();
      plm_geo_finish(r);
      S.PX1[PW_CR][x] = r.cr, S.PX1[PW_CL][x] = r.cl, S.PX1[PW_UP][x] = r.up, S.PX1[PW_LO][x] = r.lo;
      S.PX1[PW_RAB][x] = r.ra.b, S.PX1[PW_RAY][x] = r.ra.y, S.PX1[PW_RBB][x] = r.rb.b, S.PX1[PW_RBY][x] = r.rb.y;
      S.PX1[PW_DX][x] = r.dx, S.PX1[PW_RDB][x] = r.rdx.b, S.PX1[PW_RDY][x] = r.rdx.y;
    } else if (multi_d && t >= 192 && t < 192 + QY) { // x2 weights, rows j0-2 .. j0+FTY+1
      const int y = t - 192, jj = min(max(j0 - FH + y, 1), P.nj - 2);
      PlmGeo r;
      const DCoords c = coords(0, jj, 0);
      r.xvm = coords(0, jj - 1, 0).x2v(), r.xvc = c.x2v(), r.xvp = coords(0, jj + 1, 0).x2v();
      r.xf0 = c.x2[0], r.xf1 = c.x2[1], r.dx = 1.0; // (the width is not an x2-only quantity)
      plm_geo_finish(r);
      S.PX2[PW_CR][y] = r.cr, S.PX2[PW_CL][y] = r.cl, S.PX2[PW_UP][y] = r.up, S.PX2[PW_LO][y] = r.lo;
      S.PX2[PW_RAB][y] = r.ra.b, S.PX2[PW_RAY][y] = r.ra.y, S.PX2[PW_RBB][y] = r.rb.b, S.PX2[PW_RBY][y] = r.rb.y;
    }
    if (D3 && t < (k1 - k0 + 3)) { // x3 weights of planes k0-1 .. k1+1 (kernels_fused.hip plm_geo_x3 without its width)
      const int kc = k0 - 1 + t;
      PlmGeo r;
      const double f0 = g[4] + (kc - 1) * g[5], f1 = g[4] + kc * g[5];
      const double f2 = g[4] + (kc + 1) * g[5], f3 = g[4] + (kc + 2) * g[5];
      r.xvm = 0.5 * (f0 + f1), r.xvc = 0.5 * (f1 + f2), r.xvp = 0.5 * (f2 + f3);
      r.xf0 = f1, r.xf1 = f2, r.dx = 1.0;
      plm_geo_finish(r);
      S.PX3[PW_CR][t] = r.cr, S.PX3[PW_CL][t] = r.cl, S.PX3[PW_UP][t] = r.up, S.PX3[PW_LO][t] = r.lo;
      S.PX3[PW_RAB][t] = r.ra.b, S.PX3[PW_RAY][t] = r.ra.y, S.PX3[PW_RBB][t] = r.rb.b, S.PX3[PW_RBY][t] = r.rb.y;
    }
  }
  if (t == 0) S.tiny[0] = S.tiny[1] = 0;
  if (t >= 64 && t < 64 + (k1 - k0 + 1)) { // (a second wave: the first fills the x3 weights)
    const int kk = k0 + (t - 64);
    S.C3[t - 64] = m3 ? m3[MT3_COS * (P.nk + 1) + kk] : 1.0, S.S3[t - 64] = m3 ? m3[MT3_SIN * (P.nk + 1) + kk] : 0.0;
  }
  __syncthreads();
  // Coords of the zone at (column x, row y) of the staged rectangle on plane kk; c3 / s3 only where a caller reads them
  auto CO = [&](int x, int y, int kk, double c3 = 1.0, double s3 = 0.0) {
    return geotabs_coords(S.G, SYS, x, y, kk, c3, s3);
  };
  auto rec_x1 = [&](int x) {
    PlmG r;
    r.cr = S.PX1[PW_CR][x], r.cl = S.PX1[PW_CL][x], r.up = S.PX1[PW_UP][x], r.lo = S.PX1[PW_LO][x];
    r.ra.b = S.PX1[PW_RAB][x], r.ra.y = S.PX1[PW_RAY][x], r.rb.b = S.PX1[PW_RBB][x], r.rb.y = S.PX1[PW_RBY][x];
    r.dx = S.PX1[PW_DX][x], r.rdx.b = S.PX1[PW_RDB][x], r.rdx.y = S.PX1[PW_RDY][x];
    return r;
  };
  auto rec_x2 = [&](int x, int y) { // the width h (x2f1 - x2f0), h = 1 or x1v (Coords::width2)
    PlmG r;
    r.cr = S.PX2[PW_CR][y], r.cl = S.PX2[PW_CL][y], r.up = S.PX2[PW_UP][y], r.lo = S.PX2[PW_LO][y];
    r.ra.b = S.PX2[PW_RAB][y], r.ra.y = S.PX2[PW_RAY][y], r.rb.b = S.PX2[PW_RBB][y], r.rb.y = S.PX2[PW_RBY][y];
    r.dx = CO(x, y, k0).width2();
    r.rdx = recip(r.dx);
    return r;
  };
  auto rec_x3 = [&](int x, int y, int kc) { // cell kc of the own column: h (x3f1 - x3f0), h = 1, x1v or x1v sin(x2v)
    const int s = kc - (k0 - 1);
    PlmG r;
    r.cr = S.PX3[PW_CR][s], r.cl = S.PX3[PW_CL][s], r.up = S.PX3[PW_UP][s], r.lo = S.PX3[PW_LO][s];
    r.ra.b = S.PX3[PW_RAB][s], r.ra.y = S.PX3[PW_RAY][s], r.rb.b = S.PX3[PW_RBB][s], r.rb.y = S.PX3[PW_RBY][s];
    r.dx = CO(x, y, kc).width3();
    r.rdx = recip(r.dx);
    return r;
  };
  // face values of one cell along a direction: PCM / PLM_G with the guard of kernels_fused.hip.  FT = std::true_type: no
  // velocity of the stencil is tiny-but-nonzero, so every division may be the hand-scheduled one; the choice is made ONCE
  // per plane and phase (one branch around all variables of both sweeps: the six slope chains of a sweep share a basic
  // block and interleave) instead of once per variable
  auto faces_of = [&](auto FT, double qm, double q, double qp, const PlmG &r, double &up_, double &lo_) {
    if constexpr (PGG) {
      plm_g_shared<decltype(FT)::value ? 2 : 0>(qm, q, qp, up_, lo_, r);
    } else if constexpr (PG) { // Cartesian: plm.hpp:32-47, uniform spacing (fused_device.hpp slope_sel / up_val / lo_val)
      const double s_ = decltype(FT)::value ? plm_dqm_fast(qm, q, qp) : plm_dqm(qm, q, qp);
      up_ = q + s_, lo_ = q - s_;
    } else {
      up_ = q, lo_ = q; // pcm.hpp:34-88
    }
  };
  auto stage_plane = [&](const Cell6 &q, const Raw5 &hal, int par) {
#define PUTQ(m, n) if constexpr (n < NV) S.Q[n][ty + FH][tx + FH] = q.m;
    CFOR6(PUTQ)
#undef PUTQ
    bool tny = tiny_vel3(q.v1, q.v2, q.v3);
    if (hr >= 0) {
      const Cell6 h = finish_cell(hal, gm1);
#define PUTH(m, n) if constexpr (n < NV) S.Q[n][hr][hc] = h.m;
      CFOR6(PUTH)
#undef PUTH
      tny = tny || tiny_vel3(hal.v1, hal.v2, hal.v3);
    }
    if (PG && __any(tny) && (t & 63) == 0) S.tiny[par] = 1;
  };
  // Instantiations in which the five sums do not fit next to the x3 sweep (HLLC in spherical coordinates, or with the
  // N-body frame): the compiler spilled two of them as they arrived -- load, wait, store, twice per trip.  There the two
  // energy sums are fetched by the update where it subtracts them (their latency is the other wave's to cover).
  constexpr bool DS_SPLIT = !DUST && D3 && ((SYS == ARTEMIS_SPHERICAL3D && (RIEMANN == 0 || EXT)) || (RIEMANN == 0 && EXT && FTX == 32));
  // (how many of the five are prefetched: spherical HLLC with the N-body frame has room for two)
  constexpr int DS_EARLY = !DS_SPLIT ? 5 : ((SYS == ARTEMIS_SPHERICAL3D && RIEMANN == 0 && EXT) ? 2 : 3);
  double ldt = DBL_MAX, ldt_gas = DBL_MAX;
  PROF_DECL;

  // ---- the update of zone (k, j, i) from the folded sums -----------------------------------------------------------
  struct Sums { // what ApplyUpdate, FluxSource and RotatingFrameImpl sum over the faces, in their order of additions
    double dv[6];          // sum_d (A_d- F_d- - A_d+ F_d+) of D, M1, M2, M3, E, e_int
    double tm[3], te[3];   // FluxSource: pressure-gradient term of M_d, P div v term of e_int, per direction
    double rfd, rfx[3];    // RotatingFrameImpl: sum of the weighted mass fluxes, the face-mean mass flux per direction
  };
  // direction D of zone (k, j, i): lo / hi = the fluxes through its lower / upper face (momenta scaled);
  // A0 / A1 the face areas, W0 / W1 RFWeights, rdx the coordinate width's reciprocal
  auto fold = [&](auto DTAG, Sums &s, const Flux8 &lo, const Flux8 &hi, double A0, double A1, double W0, double W1,
                  double dtdx, double dt_vol, bool on) {
    constexpr int D = decltype(DTAG)::value;
    const double t0 = (A0 * lo.d - A1 * hi.d), t1 = (A0 * lo.m1 - A1 * hi.m1), t2 = (A0 * lo.m2 - A1 * hi.m2);
    const double t3 = (A0 * lo.m3 - A1 * hi.m3);
    if constexpr (D == 1) {
      s.dv[0] = t0, s.dv[1] = t1, s.dv[2] = t2, s.dv[3] = t3;
    } else if (on) {
      s.dv[0] += t0, s.dv[1] += t1, s.dv[2] += t2, s.dv[3] += t3;
    }
    if constexpr (!DUST) {
      const double t4 = (A0 * lo.e - A1 * hi.e), t5 = (A0 * lo.eg - A1 * hi.eg);
      if constexpr (D == 1) s.dv[4] = t4, s.dv[5] = t5;
      else if (on) s.dv[4] += t4, s.dv[5] += t5;
      s.tm[D - 1] = dtdx * (lo.pf - hi.pf);
      s.te[D - 1] = dt_vol * 0.5 * (lo.pf + hi.pf) * (A1 * hi.vf - A0 * lo.vf);
    }
    // sources_device.hpp rotating_frame_divf / rotating_frame_gas (inactive directions enter as 0 * (0 + 0))
    const double flo = on ? lo.d : 0.0, fup = on ? hi.d : 0.0, a0 = on ? A0 : 0.0, a1 = on ? A1 : 0.0;
    const double term = (flo * a0 * W0 + fup * a1 * W1);
    if constexpr (D == 1) s.rfd = term, s.rfx[0] = 0.5 * (flo + fup);
    else s.rfd = s.rfd + (on ? 1 : 0) * term, s.rfx[D - 1] = (on ? 1 : 0) * 0.5 * (flo + fup);
  };

  auto update = [&](const int k, const DCoordsT<true> &co, const CellMetric &cm, const double hx[3], const Cell6 &qc, Sums &s,
                    const Raw5 &u1raw, const double ds[5], const GasCons &gcz) {
    if (!active) return;
    const CurvConst &KC = S.C; // (LDS: every read below is a broadcast ds_read at its use)
    const unsigned c = col + static_cast<unsigned>(k) * sk;
    FluidPrim w;
    w.rho = qc.d, w.v1 = qc.v1, w.v2 = qc.v2, w.v3 = qc.v3, w.sie = qc.e;
    if constexpr (DUST) { // the same tasks on Dust's four conserved variables (kernels_stage_cell.hip's dust branch)
      DustCons u0 = prim_to_cons_dust(KC, w.rho, w.v1, w.v2, w.v3, hx);
      DustCons u1 = u0;
      if (a.has_u1) u1 = prim_to_cons_dust(KC, u1raw.d, u1raw.v1, u1raw.v2, u1raw.v3, hx);
      const Recip rvol = recip(cm.vol);
      const double nd = s.dv[0] * KC.beta_dt, n1m = s.dv[1] * KC.beta_dt, n2m = s.dv[2] * KC.beta_dt, n3m = s.dv[3] * KC.beta_dt;
      double qd, q1m, q2m, q3m; // (mass and momentum fluxes of a dust at rest can be tiny-but-nonzero: IEEE then)
      if (__any(tiny_nonzero(nd) || tiny_nonzero(n1m) || tiny_nonzero(n2m) || tiny_nonzero(n3m))) {
        qd = nd / cm.vol, q1m = n1m / cm.vol, q2m = n2m / cm.vol, q3m = n3m / cm.vol;
      } else {
        qd = div(nd, rvol), q1m = div(n1m, rvol), q2m = div(n2m, rvol), q3m = div(n3m, rvol);
      }
      u0.d = KC.gam0 * u0.d + KC.gam1 * u1.d + qd;
      u0.m1 = KC.gam0 * u0.m1 + KC.gam1 * u1.m1 + q1m;
      u0.m2 = KC.gam0 * u0.m2 + KC.gam1 * u1.m2 + q2m;
      u0.m3 = KC.gam0 * u0.m3 + KC.gam1 * u1.m3 + q3m;
      const double dt = KC.bdt;
      { // Dust::FluxSource (dust.cpp:303-326): the coordinate source only
        const double rdt = w.rho * dt;
        double vf[3];
        rotation_velocity(co, KC.omf, vf);
        if (co.x1dep())
          u0.m1 += rdt * (0.0 * sqr(w.v1 + vf[0]) + co.dh2dx1() * sqr(w.v2 + vf[1]) + co.dh3dx1() * sqr(w.v3 + vf[2]));
        if (co.x2dep() && multi_d)
          u0.m2 += rdt * (0.0 * sqr(w.v1 + vf[0]) + 0.0 * sqr(w.v2 + vf[1]) + co.dh3dx2() * sqr(w.v3 + vf[2]));
      }
      if (a.grav_on) {
        GravLds G{a.grav.type, KC.grav};
        gravity_dust(gravity_accel<true>(G, co, P.ndim, dt), dt, hx, w, u0);
      }
      if (a.nb_n) {
        const double wv[4] = {w.rho, w.v1, w.v2, w.v3};
        double u[4] = {u0.d, u0.m1, u0.m2, u0.m3};
        nb_apply<false>(a.nb_pl, a.nb_n, co, KC.nb_omf, dt, wv, u);
        u0.d = u[0], u0.m1 = u[1], u0.m2 = u[2], u0.m3 = u[3];
      }
      if (a.rfc_on) { // sources_device.hpp rotating_frame_dust on the folded sums
        const RotFrame rfc = rotating_frame_terms(co, KC.rf_omega, dt);
        const double qv = s.rfd / cm.vol;
        u0.m1 -= rfc.omdt * qv * rfc.ep[0];
        u0.m2 -= rfc.omdt * qv * rfc.ep[1];
        u0.m3 -= rfc.omdt * qv * rfc.ep[2];
      }
      if (a.to_cons) {
        gst(KC.cons0[0], c, u0.d), gst(KC.cons0[1], c, u0.m1), gst(KC.cons0[2], c, u0.m2);
        gst(KC.cons0[3], c, u0.m3);
        return;
      }
      if (a.finish) { // DragSource couples the fluids pointwise, on this zone's two conserved states (drag.hpp:296-482)
        const GasCons ug = gcz;
        struct { double dfloor, siefloor, de_switch; } GF{KC.g_dfloor, KC.g_siefloor, KC.g_de_switch};
        const double xv[3] = {co.x1v(), co.x2v(), co.x3v()};
        const CylVec cv = to_cyl_with_vec(co, xv);
        const double zero3[3] = {0.0, 0.0, 0.0};
        // the hand-scheduled divisions round like `/` unless a momentum is tiny-but-nonzero or a density is not a
        // positive normal number: wave-uniform choice
        const bool odd = tiny_nonzero(ug.m1) || tiny_nonzero(ug.m2) || tiny_nonzero(ug.m3) || tiny_nonzero(u0.m1) ||
                         tiny_nonzero(u0.m2) || tiny_nonzero(u0.m3) || !(ug.d > 1.0e-280 && ug.d < 1.0e280) ||
                         !(u0.d > 1.0e-280 && u0.d < 1.0e280) || !(ug.e > 1.0e-280 && ug.e < 1.0e280) ||
                         !(ug.eg > 1.0e-280 && ug.eg < 1.0e280);
        DragFinish1 o;
        if (__any(odd)) o = simple_drag1_finish<false>(KC.drag, GF, KC, gm1, dt, hx, cv, zero3, zero3, ug, u0);
        else o = simple_drag1_finish<true>(KC.drag, GF, KC, gm1, dt, hx, cv, zero3, zero3, ug, u0);
        gst(KC.out[0], c, o.dd), gst(KC.out[1], c, o.d1), gst(KC.out[2], c, o.d2), gst(KC.out[3], c, o.d3);
        gst(KC.gout[0], c, o.gd), gst(KC.gout[1], c, o.g1), gst(KC.gout[2], c, o.g2), gst(KC.gout[3], c, o.g3);
        gst(KC.gout[4], c, o.gs);
        if (a.dt_bits) { // EstimateTimestepMesh of both fluids on the new state (dust.cpp:256-272, gas.cpp:411-433)
          double denom = 0.0;
          denom += fabs(o.d1) / co.width1();
          if (multi_d) denom += fabs(o.d2) / co.width2();
          if (D3) denom += fabs(o.d3) / co.width3();
          ldt = amin(ldt, 1.0 / denom);
          const double bulk = (gm1 + 1.0) * gm1 * o.gd * o.gs; // IdealGas bulk modulus
          const double cs = sqrt(bulk / o.gd);
          double dg_ = 0.0;
          dg_ += (fabs(o.g1) + cs) / co.width1();
          if (multi_d) dg_ += (fabs(o.g2) + cs) / co.width2();
          if (D3) dg_ += (fabs(o.g3) + cs) / co.width3();
          ldt_gas = amin(ldt_gas, 1.0 / dg_);
        }
        return;
      }
      const double w_d = (u0.d > KC.dfloor) ? u0.d : KC.dfloor; // ConsToPrim (fill_derived.cpp:155-164)
      const double n1 = u0.m1 / (w_d * hx[0]), n2 = u0.m2 / (w_d * hx[1]), n3 = u0.m3 / (w_d * hx[2]);
      gst(KC.out[0], c, w_d);
      gst(KC.out[1], c, n1);
      gst(KC.out[2], c, n2);
      gst(KC.out[3], c, n3);
      if (a.dt_bits) { // Dust::EstimateTimestepMesh (dust.cpp:256-272; a dust at rest gives 1 / 0 = inf like the reference)
        double denom = 0.0;
        denom += fabs(n1) / co.width1();
        if (multi_d) denom += fabs(n2) / co.width2();
        if (D3) denom += fabs(n3) / co.width3();
        ldt = amin(ldt, 1.0 / denom);
      }
      return;
    }
    GasCons u0 = prim_to_cons_gas(KC, w.rho, w.v1, w.v2, w.v3, w.sie, hx);
    GasCons u1 = u0;
    if (a.has_u1) u1 = prim_to_cons_gas(KC, u1raw.d, u1raw.v1, u1raw.v2, u1raw.v3, u1raw.e, hx);
    // ---- ApplyUpdate (artemis_integrator.hpp:88-106)
    const Recip rvol = recip(cm.vol);
    const double nd = s.dv[0] * KC.beta_dt, n1m = s.dv[1] * KC.beta_dt, n2m = s.dv[2] * KC.beta_dt, n3m = s.dv[3] * KC.beta_dt;
    const double ne = s.dv[4] * KC.beta_dt, neg = s.dv[5] * KC.beta_dt;
    // momenta can be tiny-but-nonzero ahead of a shock, where only IEEE division is right: wave-uniform choice
    double q1m, q2m, q3m;
    if (__any(tiny_nonzero(n1m) || tiny_nonzero(n2m) || tiny_nonzero(n3m))) {
      q1m = n1m / cm.vol, q2m = n2m / cm.vol, q3m = n3m / cm.vol;
    } else {
      q1m = div(n1m, rvol), q2m = div(n2m, rvol), q3m = div(n3m, rvol);
    }
    u0.d = KC.gam0 * u0.d + KC.gam1 * u1.d + div(nd, rvol);
    u0.m1 = KC.gam0 * u0.m1 + KC.gam1 * u1.m1 + q1m;
    u0.m2 = KC.gam0 * u0.m2 + KC.gam1 * u1.m2 + q2m;
    u0.m3 = KC.gam0 * u0.m3 + KC.gam1 * u1.m3 + q3m;
    u0.e = KC.gam0 * u0.e + KC.gam1 * u1.e + div(ne, rvol);
    u0.eg = KC.gam0 * u0.eg + KC.gam1 * u1.eg + div(neg, rvol);
    // ---- FluxSource (fluid_fluxes.hpp:361-415)
    const double dt = KC.bdt;
    u0.m1 += s.tm[0];
    u0.eg -= s.te[0];
    if (multi_d) {
      u0.m2 += s.tm[1];
      u0.eg -= s.te[1];
    }
    if (D3) {
      u0.m3 += s.tm[2];
      u0.eg -= s.te[2];
    }
    {
      const double rdt = w.rho * dt;
      double vf[3];
      rotation_velocity(co, KC.omf, vf);
      if (co.x1dep())
        u0.m1 += rdt * (0.0 * sqr(w.v1 + vf[0]) + co.dh2dx1() * sqr(w.v2 + vf[1]) + co.dh3dx1() * sqr(w.v3 + vf[2]));
      if (co.x2dep() && multi_d)
        u0.m2 += rdt * (0.0 * sqr(w.v1 + vf[0]) + 0.0 * sqr(w.v2 + vf[1]) + co.dh3dx2() * sqr(w.v3 + vf[2]));
    }
    if (a.diff_on) { // Gas::DiffusionUpdate (artemis_driver.cpp:218-221): artemis_hip_viscous_source's sums
      auto dsv = [&](auto q) { // (a compile-time index: prefetched, or fetched here)
        if constexpr (decltype(q)::value < DS_EARLY) return ds[decltype(q)::value];
        else return gld(KC.dsum[decltype(q)::value], c);
      };
      u0.m1 -= dsv(std::integral_constant<int, 0>{}), u0.m2 -= dsv(std::integral_constant<int, 1>{});
      u0.m3 -= dsv(std::integral_constant<int, 2>{});
      u0.e -= dsv(std::integral_constant<int, 3>{});
      u0.eg -= dsv(std::integral_constant<int, 4>{});
    }
    if (a.grav_on) {
      GravLds G{a.grav.type, KC.grav};
      gravity_gas(gravity_accel<true>(G, co, P.ndim, dt), dt, hx, w, u0);
    }
    if constexpr (EXT) if (a.nb_n) { // Gravity::NBodyGravity (nbody_device.hpp), particle by particle on the registers
      const double wv[4] = {w.rho, w.v1, w.v2, w.v3};
      double u[6] = {u0.d, u0.m1, u0.m2, u0.m3, u0.e, u0.eg};
      nb_apply<true>(a.nb_pl, a.nb_n, co, KC.nb_omf, dt, wv, u);
      u0.d = u[0], u0.m1 = u[1], u0.m2 = u[2], u0.m3 = u[3], u0.e = u[4], u0.eg = u[5];
    }
    if (a.rfc_on) { // sources_device.hpp rotating_frame_gas on the folded sums
      const RotFrame rfc = rotating_frame_terms(co, KC.rf_omega, dt);
      const double qv = __any(tiny_nonzero(s.rfd)) ? s.rfd / cm.vol : div(s.rfd, rvol); // (divf / vol)
      u0.m1 -= rfc.omdt * qv * rfc.ep[0];
      u0.m2 -= rfc.omdt * qv * rfc.ep[1];
      u0.m3 -= rfc.omdt * qv * rfc.ep[2];
      u0.e += rfc.om2dt * rfc.R * (s.rfx[0] * rfc.eR[0] + s.rfx[1] * rfc.eR[1] + s.rfx[2] * rfc.eR[2]);
    }
    if constexpr (EXT) if (a.to_cons) { // DragSource couples the fluids next: the conserved state as the tasks would hold it
      gst(KC.cons0[0], c, u0.d), gst(KC.cons0[1], c, u0.m1), gst(KC.cons0[2], c, u0.m2);
      gst(KC.cons0[3], c, u0.m3), gst(KC.cons0[4], c, u0.e), gst(KC.cons0[5], c, u0.eg);
      return;
    }
    // ---- SetAuxillaryFields (fill_derived.cpp:58-71) + ConsToPrim (:132-146)
    const double w_d = (u0.d > KC.dfloor) ? u0.d : KC.dfloor;
    const double u_d2 = amax(u0.d, KC.dfloor);
    const Recip rd2 = recip(u_d2);
    const bool tiny_m = __any(tiny_nonzero(u0.m1) || tiny_nonzero(u0.m2) || tiny_nonzero(u0.m3));
    double rv2, rv3;
    if (tiny_m) rv2 = u0.m2 / hx[1], rv3 = u0.m3 / hx[2];
    else rv2 = div(u0.m2, hx[1]), rv3 = div(u0.m3, hx[2]);
    const double rv1 = u0.m1 / 1.0; // hx[0] == 1
    const double ke = div(0.5 * (sqr(rv1) + sqr(rv2) + sqr(rv3)), rd2);
    const double ue_cons = u0.e - ke;
    double sie = (ue_cons > KC.de_switch * u0.e) ? div(ue_cons, rd2) : div(u0.eg, rd2);
    sie = amax(sie, KC.siefloor);
    double u_u = sie * w_d;
    const double uflr = KC.siefloor * w_d;
    u_u = (u_u > uflr) ? u_u : uflr;
    const Recip rwd = recip(w_d);
    double n1, n2, n3;
    if (tiny_m) n1 = u0.m1 / (w_d * hx[0]), n2 = u0.m2 / (w_d * hx[1]), n3 = u0.m3 / (w_d * hx[2]);
    else n1 = div(u0.m1, rwd), n2 = div(u0.m2, w_d * hx[1]), n3 = div(u0.m3, w_d * hx[2]); // w_d * 1.0 == w_d
    double w_s = div(u_u, rwd);
    w_s = (w_s > KC.siefloor) ? w_s : KC.siefloor;
    gst(KC.out[0], c, w_d);
    gst(KC.out[1], c, n1);
    gst(KC.out[2], c, n2);
    gst(KC.out[3], c, n3);
    gst(KC.out[4], c, amax(0.0, gm1 * w_d * w_s)); // fill_derived.cpp:247 (consumers recompute it anyway)
    gst(KC.out[5], c, w_s);
    if (a.dt_bits) { // Gas::EstimateTimestepMesh on the new state (gas.cpp:411-433)
      const double bulk = (gm1 + 1.0) * gm1 * w_d * w_s;
      const double cs = sqrt_pos(div(bulk, rwd));
      double denom = div(fabs(n1) + cs, co.width1());
      if (multi_d) denom += div(fabs(n2) + cs, co.width2());
      if (D3) denom += div(fabs(n3) + cs, co.width3());
      ldt = amin(ldt, div(1.0, denom));
    }
  };

  // ---- one plane, phases 1 and 2: the x1 / x2 sweeps (two barriers) ----------------------------------------------------
  // plane k's primitives are staged; leaves the fluxes through the zone's lower x1 / x2 faces in fx_lo / fy_lo and the
  // tile's face fluxes in S.FY / S.FXE.  FT: the plane holds no tiny-but-nonzero velocity (workgroup-uniform).
  auto phase12 = [&](auto FT, const int k, const Cell6 &qc, const bool stage_next, const Cell6 &qn, const Raw5 &hal_next,
                     Flux8 &fx_lo, Flux8 &fy_lo) {
    constexpr bool fastp = decltype(FT)::value;
    PROF(0); // (since the end of the previous plane: this trip's loads issued)
    const int duty = (t + 64 * (k & 3)) & 255; // wave roles rotate with k
    if (duty >= 64) __builtin_amdgcn_s_setprio(2); // the duty waves first (kernels_fused.hip: -5 %)
    // ---- P1: slopes of the own zone; the tile's edge columns and rows on the duty waves -----------------------------
    Cell6 lox, loy, L;
    {
      PlmG r{};
      if constexpr (PGG) r = rec_x1(tx + FH);
#define SLX(m, n)                                                                                 \
  if constexpr (n < NV) {                                                                         \
    double up_;                                                                                   \
    faces_of(FT, S.Q[n][ty + FH][tx + FH - 1], qc.m, S.Q[n][ty + FH][tx + FH + 1], r, up_, lox.m); \
    L.m = lane_below(up_);                                                                        \
    if (tx == FTX - 1) S.UPXE[n][ty] = up_;                                                       \
  }
      CFOR6_33(SLX)
#undef SLX
    }
    __builtin_amdgcn_sched_barrier(0); // (one sweep's six chains interleave; two sweeps' would not fit the registers)
    if (multi_d) {
      PlmG r{};
      if constexpr (PGG) r = rec_x2(tx + FH, ty + FH);
#define SLY(m, n)                                                                                 \
  if constexpr (n < NV) {                                                                         \
    double up_;                                                                                   \
    faces_of(FT, S.Q[n][ty + FH - 1][tx + FH], qc.m, S.Q[n][ty + FH + 1][tx + FH], r, up_, loy.m); \
    S.UPY[n][ty + 1][tx] = up_;                                                                   \
  }
      CFOR6_33(SLY)
#undef SLY
    }
    if (duty >= 128 && duty < 128 + 2 * FTY) { // columns i0-1 (upper value) and i0+FTX (lower value)
      const int u = duty - 128, row = u >> 1, side = u & 1;
      const int cx = side ? FTX + FH : FH - 1;
      PlmG r{};
      if constexpr (PGG) r = rec_x1(cx);
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        double up_, lo_;
        faces_of(FT, S.Q[n][row + FH][cx - 1], S.Q[n][row + FH][cx], S.Q[n][row + FH][cx + 1], r, up_, lo_);
        if (side) S.LOXE[n][row] = lo_;
        else S.UPX0[n][row] = up_;
      }
    }
    if (multi_d && duty >= 192 && duty < 192 + 2 * FTX) { // rows j0-1 (upper value) and j0+FTY (lower value)
      const int u = duty - 192, cx = u % FTX, side = u / FTX;
      const int ry = side ? FTY + FH : FH - 1;
      PlmG r{};
      if constexpr (PGG) r = rec_x2(cx + FH, ry);
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        double up_, lo_;
        faces_of(FT, S.Q[n][ry - 1][cx + FH], S.Q[n][ry][cx + FH], S.Q[n][ry + 1][cx + FH], r, up_, lo_);
        if (side) S.LOY[n][cx] = lo_;
        else S.UPY[n][0][cx] = up_;
      }
    }
    PROF(1);
    __syncthreads();
    PROF(2);
    // ---- P2: Riemann problems at the own lower faces; the tile's upper perimeter on one duty wave --------------------
    if (tx == 0) { CGET6(L, S.UPX0, [ty]); }
    fx_lo = solve_fluid<DUST, RIEMANN, 1>(gk, L, lox, fastp);
    {
      double h[3];
      CO(tx + FH, ty + FH, k).face_scale(1, h); // ScaleMomentumFlux (fluid_fluxes.hpp:33-70; h1 == 1)
      fx_lo.m2 *= h[1], fx_lo.m3 *= h[2];
    }
    fy_lo = fx_lo;
    if (multi_d) {
      CGET6(L, S.UPY, [ty][tx]);
      fy_lo = solve_fluid<DUST, RIEMANN, 2>(gk, L, loy, fastp);
      double h[3];
      CO(tx + FH, ty + FH, k).face_scale(2, h);
      fy_lo.m2 *= h[1], fy_lo.m3 *= h[2];
      if (ty > 0) { CPUT8(S.FY, fy_lo, [ty - 1][tx]); }
    }
    if (duty >= 64 && duty < 128) { // lanes 0 .. FTY-1: x1 face i0+FTX per row; lanes 32 .. 32+FTX-1: x2 face j0+FTY
      // ONE Riemann pass for both kinds of face (kernels_fused.hip): the x2 lanes rotate their velocity components
      const int u = duty - 64;
      const bool isx = (u < FTY), isy = multi_d && (u >= 32) && (u < 32 + FTX);
      if (isx || isy) {
        const int cx = u - 32;
        Cell6 l, r;
        if (isx) {
          CGET6(l, S.UPXE, [u]);
          CGET6(r, S.LOXE, [u]);
        } else {
          CGET6(l, S.UPY, [FTY][cx]);
          CGET6(r, S.LOY, [cx]);
          double a_ = l.v1;
          l.v1 = l.v2, l.v2 = l.v3, l.v3 = a_;
          a_ = r.v1;
          r.v1 = r.v2, r.v2 = r.v3, r.v3 = a_;
        }
        Flux8 fe_ = solve_fluid<DUST, RIEMANN, 1>(gk, l, r, fastp);
        double h[3];
        if (isx) {
          CO(FTX + FH, u + FH, k).face_scale(1, h); // the face below zone (j0+u, i0+FTX)
          fe_.m2 *= h[1], fe_.m3 *= h[2];
          CPUT8(S.FXE, fe_, [u]);
        } else {
          const double n_ = fe_.m1; // (normal, t1, t2) = (m2, m3, m1) of the block's frame
          fe_.m1 = fe_.m3, fe_.m3 = fe_.m2, fe_.m2 = n_;
          CO(cx + FH, FTY + FH, k).face_scale(2, h); // the face below zone (j0+FTY, i0+cx)
          fe_.m2 *= h[1], fe_.m3 *= h[2];
          CPUT8(S.FY, fe_, [FTY - 1][cx]);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (stage_next) stage_plane(qn, hal_next, (k + 1) & 1);
    PROF(3);
    __syncthreads();
    PROF(4);
  };
  // (the wave-uniform, in fact workgroup-uniform, choice: S.tiny is read between two barriers by every thread)
  auto plane12 = [&](const int k, const Cell6 &qc, const bool stage_next, const Cell6 &qn, const Raw5 &hal_next, Flux8 &fx_lo,
                     Flux8 &fy_lo) {
    bool fastp = true;
    if constexpr (PG) {
      fastp = (S.tiny[k & 1] == 0);
      if (t == 0) S.tiny[(k + 1) & 1] = 0; // set again when the next plane is staged (after the first barrier)
    }
    if (fastp) phase12(std::true_type{}, k, qc, stage_next, qn, hal_next, fx_lo, fy_lo);
    else phase12(std::false_type{}, k, qc, stage_next, qn, hal_next, fx_lo, fy_lo);
  };
  // ---- after the second barrier: the upper x1 / x2 faces from the neighbours; everything folded with ONE rebuild of the
  // zone's Coords / CellMetric from the tables (nothing of them crosses a barrier) ----------------------------------
  auto upper12 = [&](const Flux8 &fx_lo, Flux8 &fx_hi, Flux8 &fy_hi) {
    fx_hi.d = lane_above(fx_lo.d), fx_hi.m1 = lane_above(fx_lo.m1), fx_hi.m2 = lane_above(fx_lo.m2);
    fx_hi.m3 = lane_above(fx_lo.m3);
    if constexpr (!DUST) {
      fx_hi.e = lane_above(fx_lo.e), fx_hi.eg = lane_above(fx_lo.eg);
      fx_hi.pf = lane_above(fx_lo.pf), fx_hi.vf = lane_above(fx_lo.vf);
    }
    if (tx == FTX - 1) { CGET8(fx_hi, S.FXE, [ty]); }
    fy_hi = fx_hi;
    if (multi_d) { CGET8(fy_hi, S.FY, [ty][tx]); }
  };

  // ---- the march ------------------------------------------------------------------------------------------------------
  Raw5 u1raw;
  u1raw.d = u1raw.v1 = u1raw.v2 = u1raw.v3 = u1raw.e = 0.0;
  double ds[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  GasCons gcz; // DUST march with the drag finish: the zone's gas conserved state
  gcz.d = gcz.m1 = gcz.m2 = gcz.m3 = gcz.e = gcz.eg = 0.0;
  auto load_gc = [&](unsigned c) {
    GasCons z;
    z.d = gld(S.C.gcons[0], c), z.m1 = gld(S.C.gcons[1], c), z.m2 = gld(S.C.gcons[2], c);
    z.m3 = gld(S.C.gcons[3], c), z.e = gld(S.C.gcons[4], c), z.eg = gld(S.C.gcons[5], c);
    return z;
  };
  auto load_ds = [&](unsigned c, int q0, int q1) {
#pragma unroll
    for (int q = 0; q < 5; ++q)
      if (q >= q0 && q < q1) ds[q] = gld(S.C.dsum[q], c);
  };
  if constexpr (!D3) {
    const unsigned c0 = col + static_cast<unsigned>(k0) * sk;
    const Cell6 qc = ldcell(in_r, in_1, in_2, in_3, in_e, c0);
    if (a.has_u1) u1raw = ldraw(S.C.u1[0], S.C.u1[1], S.C.u1[2], S.C.u1[3], S.C.u1[4], c0);
    if (a.diff_on) load_ds(c0, 0, 5);
    if constexpr (DUST) if (a.finish) gcz = load_gc(c0);
    Raw5 hal = u1raw;
    if (hr >= 0) hal = ldraw(in_r, in_1, in_2, in_3, in_e, hcol + static_cast<unsigned>(k0) * sk);
    stage_plane(qc, hal, k0 & 1);
    __syncthreads();
    Sums s;
    Flux8 fx_lo, fy_lo, fx_hi, fy_hi;
    plane12(k0, qc, false, qc, hal, fx_lo, fy_lo);
    upper12(fx_lo, fx_hi, fy_hi);
    double c3 = 1.0, s3 = 0.0;
    c3 = S.C3[0], s3 = S.S3[0];
    const auto co = CO(tx + FH, ty + FH, k0, c3, s3);
    const CellMetric cm = cell_metric_of(co);
    const double bdt = S.C.bdt;
        const double dt_vol = div(bdt, recip(cm.vol));
    double b1[2], b2[2], b3[2];
    co.rf_weights(b1, b2, b3);
    fold(std::integral_constant<int, 1>{}, s, fx_lo, fx_hi, cm.ax1[0], cm.ax1[1], b1[0], b1[1], div(bdt, cm.dx[0]), dt_vol, true);
    fold(std::integral_constant<int, 2>{}, s, fy_lo, fy_hi, cm.ax2[0], cm.ax2[1], b2[0], b2[1], multi_d ? div(bdt, cm.dx[1]) : 0.0,
         dt_vol, multi_d);
    double hx[3];
    scale_factors_of(co, hx);
    s.tm[2] = s.te[2] = 0.0, s.rfx[2] = 0 * 0.5 * (0.0 + 0.0);
    s.rfd = s.rfd + 0 * (0.0 * 0.0 * 0.0 + 0.0 * 0.0 * 0.0);
    update(k0, co, cm, hx, qc, s, u1raw, ds, gcz);
  } else {
    Cell6 qc = ldcell(in_r, in_1, in_2, in_3, in_e, col + static_cast<unsigned>(k0 - 1) * sk);
    Cell6 qn = ldcell(in_r, in_1, in_2, in_3, in_e, col + static_cast<unsigned>(k0) * sk);
    Cell6 zl;
    {
      const Cell6 qmm = ldcell(in_r, in_1, in_2, in_3, in_e, col + static_cast<unsigned>(k0 - 2) * sk);
      PlmG r{};
      if constexpr (PGG) r = rec_x3(tx + FH, ty + FH, k0 - 1);
      double unused_;
#define ZL0(m, n) if constexpr (n < NV) faces_of(std::false_type{}, qmm.m, qc.m, qn.m, r, zl.m, unused_);
      CFOR6(ZL0)
#undef ZL0
    }
#define ZPUT(m, n) if constexpr (n < NV) S.ZL[n][t] = zl.m;
    CFOR6(ZPUT)
    Flux8 fz_lo;
    fz_lo.d = fz_lo.m1 = fz_lo.m2 = fz_lo.m3 = fz_lo.e = fz_lo.eg = fz_lo.pf = fz_lo.vf = 0.0;
    Raw5 hal = u1raw; // halo zone of plane k+1 (staged by trip k)
    for (int k = k0 - 1; k <= k1; ++k) { // the first trip only primes fz_lo (face k0)
      // this trip's HBM loads first; consumed after the plane's LDS phases
      const Raw5 rnn = ldraw(in_r, in_1, in_2, in_3, in_e, col + static_cast<unsigned>(k + 2) * sk);
      const bool live = k >= k0;
      const unsigned ck = col + static_cast<unsigned>(max(k, 0)) * sk;
      // (UNCONDITIONAL, like every load of the trip: behind a branch the compiler's s_waitcnt for an OLDER load must
      // assume the younger ones were not issued and waits for them as well -- the prefetch would be waited for in full.
      // Threads without a halo zone fetch their own zone again: hcol == col for them, the line is in L1.)
      hal = ldraw(in_r, in_1, in_2, in_3, in_e, hcol + static_cast<unsigned>(min(k + 1, P.nk - 1)) * sk);
      Sums s;
      if (live) {
        Flux8 fx_lo, fy_lo, fx_hi, fy_hi;
        plane12(k, qc, k < k1, qn, hal, fx_lo, fy_lo);
        upper12(fx_lo, fx_hi, fy_hi);
        // the x1 / x2 faces are folded at once (only the sums live through the x3 sweep); the zone's Coords are rebuilt
        // from the tables wherever they are needed: nothing of them crosses a barrier or the sweep
        const auto co = CO(tx + FH, ty + FH, k);
        const CellMetric cm = cell_metric_of(co);
        const double bdt = S.C.bdt;
        const double dt_vol = div(bdt, recip(cm.vol));
        double b1[2], b2[2], b3[2];
        co.rf_weights(b1, b2, b3);
        fold(std::integral_constant<int, 1>{}, s, fx_lo, fx_hi, cm.ax1[0], cm.ax1[1], b1[0], b1[1], div(bdt, cm.dx[0]), dt_vol, true);
        fold(std::integral_constant<int, 2>{}, s, fy_lo, fy_hi, cm.ax2[0], cm.ax2[1], b2[0], b2[1], div(bdt, cm.dx[1]), dt_vol, true);
        PROF(5);
      } else { // priming trip: stage the first plane
        stage_plane(qn, hal, k0 & 1);
        __syncthreads();
      }
      // x3 sweep, registers only: slope of zone k+1, face k+1
      const Cell6 qnn = finish_cell(rnn, gm1);
      // what only the update reads (start-of-step state, viscous sums) is fetched here, AFTER the trip's prefetch has been
      // consumed (these loads sit behind run-time conditions: see above): the x3 sweep covers their latency, and ten
      // doubles less are alive through the plane's two LDS phases
      // (the trip's prefetch has arrived -- the halo zone was staged before the last barrier --, so this wait is free; it
      // tells the compiler's counter model so, wherever its scheduler moves the conditional loads below)
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
      if (a.has_u1 && live) u1raw = ldraw(S.C.u1[0], S.C.u1[1], S.C.u1[2], S.C.u1[3], S.C.u1[4], ck);
      if (a.diff_on && live) load_ds(ck, 0, DS_EARLY);
      if constexpr (DUST) if (a.finish && live) gcz = load_gc(ck);
      PROF(9);
      Cell6 zr, zl_next;
      Flux8 fz_hi;
      auto sweep3 = [&](auto FT) {
        PlmG r{};
        if constexpr (PGG) r = rec_x3(tx + FH, ty + FH, k + 1);
#define ZSL(m, n) if constexpr (n < NV) faces_of(FT, qc.m, qn.m, qnn.m, r, zl_next.m, zr.m);
        CFOR6_33(ZSL)
#undef ZSL
        Cell6 zl; // (written by this thread in the previous trip: no barrier needed)
#define ZGET(m, n) if constexpr (n < NV) zl.m = S.ZL[n][t];
        CFOR6(ZGET)
#undef ZGET
        fz_hi = solve_fluid<DUST, RIEMANN, 3>(gk, zl, zr, true);
        zl = zl_next;
        CFOR6(ZPUT)
      };
      bool fast3 = true;
      if constexpr (PG) {
        // the own column's three zones decide for the wave whether the hand-scheduled divisions are safe
        const bool tiny3 = tiny_nonzero(qc.v1) || tiny_nonzero(qc.v2) || tiny_nonzero(qc.v3) || tiny_nonzero(qn.v1) ||
                           tiny_nonzero(qn.v2) || tiny_nonzero(qn.v3) || tiny_nonzero(qnn.v1) || tiny_nonzero(qnn.v2) ||
                           tiny_nonzero(qnn.v3);
        fast3 = !__any(tiny3);
      }
      if (fast3) sweep3(std::true_type{});
      else sweep3(std::false_type{});
      PROF(6);
      {
        double h[3];
        CO(tx + FH, ty + FH, k0).face_scale(3, h); // ScaleMomentumFlux at the x3 face (no x3 dependence)
        fz_hi.m2 *= h[1], fz_hi.m3 *= h[2];
      }
      if (live) {
        const auto co = CO(tx + FH, ty + FH, k, S.C3[k - k0], S.S3[k - k0]);
        const CellMetric cm = cell_metric_of(co);
        const double bdt = S.C.bdt;
        const double dt_vol = div(bdt, recip(cm.vol));
        double b1[2], b2[2], b3[2];
        co.rf_weights(b1, b2, b3);
        fold(std::integral_constant<int, 3>{}, s, fz_lo, fz_hi, cm.ax3[0], cm.ax3[1], b3[0], b3[1], div(bdt, cm.dx[2]), dt_vol, true);
        double hx[3];
        scale_factors_of(co, hx);
        update(k, co, cm, hx, qc, s, u1raw, ds, gcz);
        PROF(7);
      }
      fz_lo = fz_hi, qc = qn, qn = qnn;
    }
#undef ZPUT
  }
#ifdef CURV_PROF
  PROF(8);
  if ((t & 63) == 0)
    for (int q = 0; q < 10; ++q) atomicAdd(&g_curv_prof[q], prof_acc[q]);
#endif
  if (a.dt_bits) {
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) ldt = fmin(ldt, __shfl_down(ldt, off, 64));
    if ((t & 63) == 0) S.wmin[t >> 6] = ldt;
    __syncthreads();
    if (t == 0) {
      double m = S.wmin[0];
      for (int w = 1; w < 4; ++w) m = fmin(m, S.wmin[w]);
      if (m < DBL_MAX) atomicMin(a.dt_bits, static_cast<unsigned long long>(__double_as_longlong(S.C.cfl * m)));
    }
    if constexpr (DUST) if (a.finish) { // ... and the gas fluid's limit of the same zones
      __syncthreads();
      for (int off = 32; off > 0; off >>= 1) ldt_gas = fmin(ldt_gas, __shfl_down(ldt_gas, off, 64));
      if ((t & 63) == 0) S.wmin[t >> 6] = ldt_gas;
      __syncthreads();
      if (t == 0) {
        double m = S.wmin[0];
        for (int w = 1; w < 4; ++w) m = fmin(m, S.wmin[w]);
        if (m < DBL_MAX) atomicMin(a.dt_bits, static_cast<unsigned long long>(__double_as_longlong(S.C.cfl_gas * m)));
      }
    }
  }
}

template <int SYS, int RIEMANN, int RECON, bool D3>
void launch_tile_dust(const PackView &P, const CurvK &k, bool narrow, unsigned grid, hipStream_t s) {
  if (narrow) hipLaunchKernelGGL((stage_curv_kernel<SYS, RIEMANN, RECON, D3, 16, true, true>), dim3(grid), dim3(256), 0, s, P, k);
  else hipLaunchKernelGGL((stage_curv_kernel<SYS, RIEMANN, RECON, D3, 32, true, true>), dim3(grid), dim3(256), 0, s, P, k);
}
template <int SYS, int RIEMANN, int RECON, bool D3>
void launch_tile(const PackView &P, const CurvK &k, bool narrow, unsigned grid, hipStream_t s) {
  if (k.to_cons || k.nb_n) {
    if (narrow) hipLaunchKernelGGL((stage_curv_kernel<SYS, RIEMANN, RECON, D3, 16, true>), dim3(grid), dim3(256), 0, s, P, k);
    else hipLaunchKernelGGL((stage_curv_kernel<SYS, RIEMANN, RECON, D3, 32, true>), dim3(grid), dim3(256), 0, s, P, k);
    return;
  }
  if (narrow) hipLaunchKernelGGL((stage_curv_kernel<SYS, RIEMANN, RECON, D3, 16>), dim3(grid), dim3(256), 0, s, P, k);
  else hipLaunchKernelGGL((stage_curv_kernel<SYS, RIEMANN, RECON, D3, 32>), dim3(grid), dim3(256), 0, s, P, k);
}
template <int SYS, bool D3>
void launch_sys_dust(const PackView &P, const CurvK &k, int riemann, int recon, bool narrow, unsigned grid, hipStream_t s) {
  if (riemann == ARTEMIS_LLF) {
    if (recon == ARTEMIS_PCM) launch_tile_dust<SYS, 2, 0, D3>(P, k, narrow, grid, s);
    else launch_tile_dust<SYS, 2, 1, D3>(P, k, narrow, grid, s);
  } else {
    if (recon == ARTEMIS_PCM) launch_tile_dust<SYS, 1, 0, D3>(P, k, narrow, grid, s);
    else launch_tile_dust<SYS, 1, 1, D3>(P, k, narrow, grid, s);
  }
}
template <int SYS, bool D3>
void launch_sys(const PackView &P, const CurvK &k, int riemann, int recon, bool narrow, unsigned grid, hipStream_t s) {
#define RC(RS)                                                                 \
  case RS:                                                                     \
    if (recon == ARTEMIS_PCM) launch_tile<SYS, RS, 0, D3>(P, k, narrow, grid, s); \
    else launch_tile<SYS, RS, 1, D3>(P, k, narrow, grid, s);                   \
    break;
  switch (riemann) {
    RC(0)
    RC(1)
    RC(2)
  }
#undef RC
}
} // namespace

// Gas (one species) on a non-Cartesian system, PCM / PLM, with the pointwise tasks the kernel folds in; diffusion only
// as artemis_hip_viscous_source's sums (the flux-array form stays on kernels_fused.hip's instantiation).  A dust
// species beside it, drag and N-body gravity are fine: the dust runs on its cell-centred kernel and the drag finish
// couples the two (launch_stage_cell).
bool curv_march_covers(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas) {
  if (opt(OPT_NO_CURV_MARCH)) return false;
  if (static_cast<long>(P.nk) * P.nj * P.ni >= (1L << 29)) return false;
  if (P.gas.ns != 1 || P.dust.ns > 1 || P.ng < 2) return false;
  // Cartesian packs with the pointwise sources the tuned kernel (kernels_fused.hip) and the 2-D row march do not carry --
  // gravity, viscosity as sums -- on the same march: plain PLM, every metric factor 1 (SYS = cartesian instantiations);
  // the rotating frame of a Cartesian pack is the shearing box, which this march does not have
  if (P.coords == ARTEMIS_CARTESIAN && (P.dust.ns != 0 || g.rf_omega != 0.0 || g.nbody_n || P.ndim < 2 || opt(OPT_NO_CART_MARCH)))
    return false;
  if (!g.pcm && recon_gas == ARTEMIS_PPM) return false;
  if (g.cooling) return false;
  if (g.diffusion && !g.diffusion_sums) return false;
  if (g.nbody_n && P.coords != ARTEMIS_CYLINDRICAL && P.coords != ARTEMIS_SPHERICAL3D) return false;
  if (g.gravity && g.gravity->type != ARTEMIS_GRAVITY_UNIFORM && g.gravity->type != ARTEMIS_GRAVITY_POINT &&
      g.gravity->type != ARTEMIS_GRAVITY_BINARY)
    return false;
  // the systems by dimensionality (geometry.hpp:38-56 CoordSelect): anything else keeps the older kernel
  const int nd = P.ndim;
  switch (P.coords) {
  case ARTEMIS_CARTESIAN: return nd >= 2;
  case ARTEMIS_CYLINDRICAL: return nd >= 2;
  case ARTEMIS_SPHERICAL1D: return nd == 1;
  case ARTEMIS_SPHERICAL2D: return nd == 2;
  case ARTEMIS_SPHERICAL3D: return nd == 3;
  case ARTEMIS_AXISYMMETRIC: return nd >= 1;
  default: return false;
  }
}

// The dust species beside it on the same march (DUST instantiations): PCM / PLM, HLLE / LLF
bool curv_march_covers_dust(const PackView &P, const artemis_stage_general_args_t &g, int recon_dust, int riemann_dust) {
  if (P.dust.ns != 1 || opt(OPT_NO_CURV_DUST_MARCH)) return false;
  if (!g.pcm && recon_dust == ARTEMIS_PPM) return false;
  return riemann_dust == ARTEMIS_HLLE || riemann_dust == ARTEMIS_LLF;
}

// fluid 0: the gas march; fluid 1: the dust march (same tiles, same chunks)
void launch_stage_curv(const PackView &P, const artemis_stage_general_args_t &g, int fluid, int recon_in, int riemann, hipStream_t s,
                       bool finish) {
  const bool dust = fluid != 0;
  CurvK k;
  k.gam0 = g.gam0, k.gam1 = g.gam1, k.beta_dt = g.beta_dt, k.bdt = g.bdt, k.cfl = dust ? g.cfl_dust : g.cfl_gas;
  k.bdt_ptr = g.beta_dt_dev;
  if (dust) k.prim_in = g.dust_in, k.prim_u1 = g.dust_u1, k.prim_out = g.dust_out;
  else k.prim_in = g.gas_in, k.prim_u1 = g.gas_u1, k.prim_out = g.gas_out;
  k.to_cons = (g.drag || g.defer_finish == 1) ? 1 : 0;
  k.finish = 0, k.gas_out = g.gas_out, k.cfl_gas = g.cfl_gas;
  std::memset(&k.drag, 0, sizeof k.drag);
  if (dust && finish) { // the dust march couples the fluids itself and writes both fluids' primitives (stage_finish_in_march)
    k.to_cons = 0, k.finish = 1;
    k.drag.stokes = (g.drag->model == ARTEMIS_DRAG_STOKES) ? 1 : 0, k.drag.tau = g.drag->tau[0], k.drag.scale = g.drag->scale;
    k.drag.grain_density = g.drag->grain_density, k.drag.size = g.drag->sizes[0];
  }
  k.dt_bits = k.to_cons ? nullptr : reinterpret_cast<unsigned long long *>(g.dt_dev);
  k.nb_pl = g.nbody_dev, k.nb_n = g.nbody_n, k.nb_omf = g.nbody_omf;
  k.has_u1 = (k.prim_u1 != k.prim_in) ? 1 : 0;
  const int nx = P.ie - P.is + 1, ny = P.je - P.js + 1, nz = P.ke - P.ks + 1;
  // tile shape: 32 x 8, or 16 x 16 where a 32-zone row would leave half the lanes without a zone (16-zone blocks)
  const bool narrow = (nx % 32 != 0) && (nx % 16 == 0 || nx < 32) && ny > 8;
  const int ftx = narrow ? 16 : 32, fty = 256 / ftx;
  k.nti = (nx + ftx - 1) / ftx, k.ntj = (ny + fty - 1) / fty;
  const long tiles = static_cast<long>(k.nti) * k.ntj * P.nb;
  // chunks along x3 (one priming trip each): long ones, but enough workgroups for the chip's 512 slots
  int kch = CKMAX;
  if (opt(OPT_CURV_KCHUNK) > 0) kch = std::min<int>(CKMAX, static_cast<int>(opt(OPT_CURV_KCHUNK)));
  else if (P.ndim > 2) // full rounds of the 512 slots (two workgroups per CU) x few priming trips: kernels.hpp
    kch = (nz + pick_march_chunks(nz, tiles, 512, CKMAX, 0.5) - 1) / pick_march_chunks(nz, tiles, 512, CKMAX, 0.5);
  k.nchunk = (P.ndim > 2) ? (nz + kch - 1) / kch : 1;
  k.kchunk = (nz + k.nchunk - 1) / k.nchunk;
  k.nchunk = (P.ndim > 2) ? (nz + k.kchunk - 1) / k.kchunk : 1;
  k.grav_on = (g.gravity && (g.time >= g.gravity->tstart) && (g.time < g.gravity->tstop)) ? 1 : 0;
  if (k.grav_on) k.grav = *g.gravity;
  k.rfc_on = (g.rf_omega != 0.0) ? 1 : 0, k.rf_omega = g.rf_omega;
  k.diff_on = (!dust && g.diffusion != nullptr) ? 1 : 0;
  k.dsum = dust ? nullptr : g.diffusion_sums;
  const unsigned grid = static_cast<unsigned>(tiles * k.nchunk);
  const int recon = g.pcm ? ARTEMIS_PCM : recon_in;
  const bool d3 = P.ndim > 2;
#define CURV_SYS(SYSV, D3V)                                                          \
  do {                                                                               \
    if (dust) launch_sys_dust<SYSV, D3V>(P, k, riemann, recon, narrow, grid, s);     \
    else launch_sys<SYSV, D3V>(P, k, riemann, recon, narrow, grid, s);               \
  } while (0)
  switch (P.coords) {
  case ARTEMIS_CARTESIAN: // (gas only: curv_march_covers)
    if (d3) launch_sys<ARTEMIS_CARTESIAN, true>(P, k, riemann, recon, narrow, grid, s);
    else launch_sys<ARTEMIS_CARTESIAN, false>(P, k, riemann, recon, narrow, grid, s);
    break;
  case ARTEMIS_CYLINDRICAL:
    if (d3) CURV_SYS(ARTEMIS_CYLINDRICAL, true);
    else CURV_SYS(ARTEMIS_CYLINDRICAL, false);
    break;
  case ARTEMIS_SPHERICAL1D: CURV_SYS(ARTEMIS_SPHERICAL1D, false); break;
  case ARTEMIS_SPHERICAL2D: CURV_SYS(ARTEMIS_SPHERICAL2D, false); break;
  case ARTEMIS_SPHERICAL3D: CURV_SYS(ARTEMIS_SPHERICAL3D, true); break;
  case ARTEMIS_AXISYMMETRIC:
    if (d3) CURV_SYS(ARTEMIS_AXISYMMETRIC, true);
    else CURV_SYS(ARTEMIS_AXISYMMETRIC, false);
    break;
  default: break;
  }
#undef CURV_SYS
}

#ifdef CURV_PROF
extern "C" int artemis_hip_debug_curv_prof(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_curv_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_curv_prof), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif
} // namespace artemis
