// Tuned fused RK stage for gas (one species), Cartesian, PCM/PLM (everything else: the general
// cell-centred stage in kernels_stage_cell.hip): CalculateFluxes -> ApplyUpdate ->
// FluxSource -> SetAuxillaryFields -> ConsToPrim -> PrimToCons(interior) in ONE pass
// (artemis_driver.cpp:182-261 with every optional package disabled).
//
// Shape of the computation on gfx950
//   * 2.5-D streaming: a 256-thread workgroup owns a 32(i) x 8(j) column of cells and marches
//     along k through a chunk of planes.  Each thread keeps its own cell column's rolling
//     window (planes k, k+1, k+2), the carried left state and the carried lower x3-face flux
//     in registers, so the x3 sweep needs no LDS and no redundant work inside a chunk.
//   * Per plane the x1/x2 sweeps exchange only what a neighbour cannot recompute cheaply:
//     staged primitives (with a 2-cell halo), the upper face value of the lower neighbour, and
//     the 8 face outputs of the upper neighbour - two barriers per plane (the next plane is
//     staged into the dead primitive tile during the Riemann phase), 80 KiB of LDS per
//     workgroup, two workgroups per CU.
//   * Every slope and every Riemann problem inside a tile is computed exactly once; the tile's
//     perimeter (one extra face per row/column and the two halo slopes behind it) is packed
//     into spare waves.
//   * The conserved state never comes from HBM: u0 = PrimToCons(prim_in) and
//     u1 = PrimToCons(prim_u1) are rebuilt in registers (fill_derived.cpp:212-276 is applied to
//     the whole block at the end of every stage, so this is an identity, not an approximation).
//     Pressure of stencil cells is likewise recomputed as (gm1*rho)*sie.
// HBM traffic per cell-stage: 5 reads (+5 for u1 after stage 1) + 6 writes instead of the
// reference's ~124 doubles.  Results are bit-identical to the per-task kernels.
#include <algorithm>
#include <cfloat>
#include <cstdlib>

#include <type_traits>

#include "device_math.hpp"
#include "diffusion_device.hpp"
#include "fused_device.hpp"
#include "geometry.hpp"
#include "kernels.hpp"
#include "options.hpp"
#include "pack_view.hpp"
#include "sources_device.hpp"
#include "task_device.hpp"

// -DFUSED_PROF (development builds only: ARTEMIS_HIPFLAGS_KERNELS_FUSED=-DFUSED_PROF): every wave sums the shader-clock
// cycles it spends in each phase of a plane (work and waiting alike); artemis_hip_debug_fused_prof reads the totals
#ifdef FUSED_PROF
__device__ unsigned long long g_fused_prof[16];
#define FPROF_DECL unsigned long long prof_t = __builtin_readcyclecounter(), prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define FPROF(slot)                                               \
  do {                                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    prof_acc[slot] += now_ - prof_t;                              \
    prof_t = now_;                                                \
  } while (0)
#define FPROF_ARGS , unsigned long long &prof_t, unsigned long long *prof_acc
#define FPROF_PASS , prof_t, prof_acc
#else
#define FPROF_DECL
#define FPROF(slot)
#define FPROF_ARGS
#define FPROF_PASS
#endif

namespace artemis {
namespace {

#ifndef ARTEMIS_CURV_OCC2
#define ARTEMIS_CURV_OCC2 0 // experiment: force the curvilinear instantiations to two waves per SIMD (spills)
#endif
// Wave priority of the duty waves (plane_sweeps): 1 = on, 0 = off (timing experiments).
#ifndef ARTEMIS_PERI_PRIO
#define ARTEMIS_PERI_PRIO 1
#endif
#ifndef ARTEMIS_FTY
#define ARTEMIS_FTY 8
#endif
constexpr int FTX = 32, FTY = ARTEMIS_FTY, FH = 2; // tile and halo
constexpr int QX = FTX + 2 * FH, QY = FTY + 2 * FH;
constexpr int NT = FTX * FTY;
constexpr int NW = NT / 64; // waves per workgroup

struct StageK {
  double gam0, gam1, beta_dt, bdt, cfl;
  const double *bdt_ptr; // optional device scalar beta*dt (replaces beta_dt and bdt)
  double *const *prim_in, *const *prim_u1, *const *prim_out, *const *cons_out;
  unsigned long long *dt_bits;
  // Work decomposition: up to 7 boxes of (i-tiles x j-tiles x k-range), each k-range cut into
  // chunks; workgroup `id` belongs to box q with start[q] <= id < start[q+1].
  int nbox;
  int ti0[7], nti[7], tj0[7], ntj[7], kb0[7], kb1[7], nchunk[7], kchunk[7];
  int start[8];
  int xcd_swizzle; // remap ids so that each XCD's L2 sees neighbouring tiles
  // shell-first signalling: workgroups with id < nshell publish their stores (agent-scope
  // release) and count themselves into *shell_done when finished
  int nshell;
  unsigned *shell_done;
  // detect-and-redo (Cartesian stage): zones whose stencil holds a velocity the hand-scheduled divisions cannot take
  // skip their stores and append their id (block * zones-per-block + cell index) to a list; stage_redo_kernel
  // recomputes them with IEEE arithmetic from the untouched input buffer.  [0]: bulk workgroups, [1]: shell
  // workgroups of a shell-first launch (redone on the stream that waits for the shell).  Null: no detection.
  unsigned *redo_cnt0, *redo_cnt1;
  unsigned long long *redo_list0, *redo_list1;
  unsigned redo_cap; // entries per list (>= the zones of the launch, so it only binds if a shell list is never drained)
  const unsigned *tiny_in; // artemis_stage_args_t: detection runs only if *tiny_in != 0 (null: always)
  unsigned *tiny_out;
  int outflow; // artemis_stage_args_t.outflow_faces: bit f = do not read the ghost zones behind face f (stage the edge zone)
  // ... block by block (outflow_faces_by_block): six bits per block, up to ten blocks; outflow_pb = 1 selects it
  unsigned long long outflow_blk;
  int outflow_pb;
};
ADEV int outflow_of(const int outflow, const unsigned long long blk, const int pb, const int b) {
  return pb ? static_cast<int>((blk >> (6 * b)) & 63ull) : outflow;
}

struct LdsTile {
  double Q[6][QY][QX];          // staged primitives of plane k (rho, v1, v2, v3, P, sie)
  double UPX[6][FTY][FTX + 1];  // upper x1-face value of cells i0-1 .. i0+31
  double LOX[6][FTY];           // lower x1-face value of cell i0+32
  double UPY[6][FTY + 1][FTX];  // upper x2-face value of rows j0-1 .. j0+7
  double LOY[6][FTX];           // lower x2-face value of row j0+8
  double FX[8][FTY][FTX];       // x1 faces i0+1 .. i0+32 (upper faces of the tile's cells)
  double FY[8][FTY][FTX];       // x2 faces j0+1 .. j0+8
};
static_assert(sizeof(LdsTile) <= (FTY == 8 ? 80 : 160) * 1024, "LDS budget: two 32x8 workgroups (or one 32x16) per CU");

using namespace fused;

#define FOR6(X) X(d) X(v1) X(v2) X(v3) X(p) X(e)

struct Ctx { // per-thread constants of the march
  int tx, ty, t, b, i0, j0;
  bool active, multi_d, three_d;
  unsigned col, sj, sk; // element offsets in 32 bits (arrays below 2^29 elements: launch_* check it)
  double dx1, dx2, gm1;
  double beta_dt, bdt; // artemis_integrator.hpp:66, artemis_driver.cpp:168
  GasK gk;
  Recip rdx1, rdx2;
  int hr, hc;   // LDS slot (Q row/col) of the halo cell this thread stages, or hr < 0
  unsigned hcol; // its column offset within a plane
  const double *g;
  const double *in_r, *in_1, *in_2, *in_3, *in_e;
};

// ---- curvilinear instantiations (CURV) ----------------------------------------------------------
// Same tile machinery; what changes is the geometry every step of the reference takes from
// Coords<GEOM>: the PLM_G weights (plm.hpp:54-73), ScaleMomentumFlux (fluid_fluxes.hpp:33-70), the
// face areas / volume / widths of the update, the scale factors of PrimToCons / ConsToPrim and
// FluxSource's coordinate sources -- plus the pointwise tasks the curvilinear decks switch on
// (ExternalGravity, RotatingFrameImpl from the cell's own mass fluxes, DiffusionUpdate from the stored
// diffusion fluxes), evaluated with the device functions of the cell-centred general stage: same bits.
// A thread owns one (i, j) column for the whole march, so whatever depends on (i, j) only is a
// register constant of the march; the x3 edges (and the x3 trigonometry) are refreshed per plane.
// These instantiations run ONE workgroup per CU (up to 512 registers per lane, no scratch).
struct PlmG { // the fields of PlmGeo that plm_g_shared reads (geometry.hpp)
  double dx, cr, cl, up, lo;
  Recip ra, rb, rdx;
};
ADEV PlmG compact(const PlmGeo &g) {
  PlmG c;
  c.dx = g.dx, c.cr = g.cr, c.cl = g.cl, c.up = g.up, c.lo = g.lo, c.ra = g.ra, c.rb = g.rb, c.rdx = g.rdx;
  return c;
}
// PLM_G record of cell kc along x3 for the column whose (i, j) geometry is `co0` (== plm_geo(P, b, 3, kc, j, i))
ADEV PlmGeo plm_geo_x3(const DCoords &co0, const double *g, int kc) {
  PlmGeo r;
  const double f0 = g[4] + (kc - 1) * g[5], f1 = g[4] + kc * g[5];
  const double f2 = g[4] + (kc + 1) * g[5], f3 = g[4] + (kc + 2) * g[5];
  r.xvm = 0.5 * (f0 + f1), r.xvc = 0.5 * (f1 + f2), r.xvp = 0.5 * (f2 + f3);
  r.xf0 = f1, r.xf1 = f2;
  DCoords c = co0;
  c.x3[0] = f1, c.x3[1] = f2;
  r.dx = c.width3();
  plm_geo_finish(r);
  return r;
}
struct SrcK { // pointwise tasks folded into the curvilinear instantiations
  int grav_on, rfc_on, diff_on, do_viscosity;
  double *const *dsum; // artemis_stage_general_args_t.diffusion_sums or null
  double rf_omega;
  artemis_gravity_t grav;
};
template <bool CURV>
struct SrcArg {};
template <>
struct SrcArg<true> {
  SrcK v;
};
struct LdsTileCurv : LdsTile { // + the PLM_G records of the tile perimeter (constant along the march)
  PlmG GX1[2];      // x1 records of columns i0-1 and i0+32
  PlmG GX2[2][FTX]; // x2 records of rows j0-1 and j0+8, per column
  int tiny[2];      // plane (k & 1) holds a tiny-but-nonzero velocity: its slopes take IEEE division
  double HF1[FTY][2]; // ScaleMomentumFlux factors (h2, h3) of the perimeter x1 faces (cell i0+32, row u)
  double HF2[FTX][2]; // ... of the perimeter x2 faces (cell j0+8, column cx); both constant along the march
};
template <bool CURV>
struct GeoCtx {};
template <>
struct GeoCtx<true> {
  DCoords co;        // own cell (clamped indices); x3 edges / trigonometry follow the march
  PlmG g1, g2;       // PLM_G records of the own cell along x1 and x2
  double h1[3], h2[3], h3[3]; // ScaleMomentumFlux factors at the centroid of the lower x1 / x2 / x3 face
  const double *m3;  // cos / sin of the x3 cell centres (spherical3D, axisymmetric) or null
};

ADEV void put6(double (*A)[QY][QX], int r, int c, const Cell6 &q) {
  A[0][r][c] = q.d, A[1][r][c] = q.v1, A[2][r][c] = q.v2;
  A[3][r][c] = q.v3, A[4][r][c] = q.p, A[5][r][c] = q.e;
}
#define GET6(dst, A, ...)                                                                  \
  dst.d = A[0] __VA_ARGS__, dst.v1 = A[1] __VA_ARGS__, dst.v2 = A[2] __VA_ARGS__,          \
  dst.v3 = A[3] __VA_ARGS__, dst.p = A[4] __VA_ARGS__, dst.e = A[5] __VA_ARGS__
#define PUT8(A, fl, ...)                                                                   \
  A[0] __VA_ARGS__ = fl.d, A[1] __VA_ARGS__ = fl.m1, A[2] __VA_ARGS__ = fl.m2,             \
  A[3] __VA_ARGS__ = fl.m3, A[4] __VA_ARGS__ = fl.e, A[5] __VA_ARGS__ = fl.eg,             \
  A[6] __VA_ARGS__ = fl.pf, A[7] __VA_ARGS__ = fl.vf
#define GET8(fl, A, ...)                                                                   \
  fl.d = A[0] __VA_ARGS__, fl.m1 = A[1] __VA_ARGS__, fl.m2 = A[2] __VA_ARGS__,             \
  fl.m3 = A[3] __VA_ARGS__, fl.e = A[4] __VA_ARGS__, fl.eg = A[5] __VA_ARGS__,             \
  fl.pf = A[6] __VA_ARGS__, fl.vf = A[7] __VA_ARGS__


// Stage one plane's primitives (own cell + this thread's halo cell) into S.Q.
template <class TILE>
ADEV void stage_plane(TILE &S, const Ctx &x, const Cell6 &q, const Raw5 &hal) {
  put6(S.Q, x.ty + FH, x.tx + FH, q);
  if (x.hr >= 0) put6(S.Q, x.hr, x.hc, finish_cell(hal, x.gm1));
}
// Guarded tiles (the curvilinear kernel and the flux task) note whether the plane they stage (parity `par`) holds a
// velocity the hand-scheduled divisions cannot take (geometry.hpp: tiny_nonzero -- ahead of a shock velocities decay
// like 1e-40, 1e-80, 1e-160, 1e-320): that plane's slopes and Riemann problems then take the IEEE divisions.  The
// flag was cleared one phase earlier.  The Cartesian tile is exactly 80 KB, so its two flags live in corner zones of
// the staged tile that no stencil reads and no thread stages (rows 0 x columns 0 / 1 of the halo frame).
ADEV int &tiny_flag(LdsTileCurv &S, int par) { return S.tiny[par]; }
ADEV int &tiny_flag(LdsTile &S, int par) { return reinterpret_cast<int *>(&S.Q[5][0][0])[2 * par]; }
template <class TILE>
ADEV void stage_plane_flag(TILE &S, const Ctx &x, const Cell6 &q, const Raw5 &hal, int par) {
  bool t = tiny_vel3(q.v1, q.v2, q.v3);
  if (x.hr >= 0) t = t || tiny_vel3(hal.v1, hal.v2, hal.v3);
  if (__any(t) && (x.t & 63) == 0) tiny_flag(S, par) = 1;
}
ADEV bool tiny_v(const Cell6 &q) { return tiny_vel3(q.v1, q.v2, q.v3); }

// x1/x2 sweeps of one plane through LDS.  Plane k's primitives are already staged in S.Q (by
// the previous plane's call, or by the prologue).  TWO barriers per plane:
//   P1  slopes from S.Q, publish upper face values              -- barrier --
//   P2  Riemann problems, publish face outputs; S.Q is dead now, so the NEXT plane's
//       primitives (`qn`, `hal_next`) are staged into it here    -- barrier --
// Returns the fluxes through the own cell's lower x1/x2 faces; the upper ones are left in
// S.FX / S.FY for plane_update.  The perimeter duties rotate over the waves with k so that no
// wave (and no SIMD) carries the extra Riemann pass every plane.
// DETECT (the Cartesian stage): the plane flags are kept but every division stays hand-scheduled; `flagged` tells the
// caller that this plane's tile (halo included) holds a tiny velocity, so that the zones concerned are redone exactly.
template <int RIEMANN, int RECON, bool D3, bool CURV, bool GUARD, bool DETECT, class TILE>
ADEV void plane_sweeps(TILE &S, const PackView &P, const Ctx &x, const GeoCtx<CURV> &gx, const int k,
                       const Cell6 &qc, const bool stage_next, const Cell6 &qn, const Raw5 &hal_next,
                       Flux8 &fx_lo, Flux8 &fy_lo, bool &flagged, const bool det FPROF_ARGS) {
  FPROF(0);
  constexpr bool PG = CURV && RECON == 1; // PLM_G instead of the uniform-mesh slope
  bool fastp = true; // every division hand-scheduled (no tiny velocity in this plane's tile)
  if constexpr (GUARD || DETECT) {
    if (GUARD || det) { // (det: wave-uniform, the stage's input may hold a tiny velocity at all)
      flagged = (tiny_flag(S, k & 1) != 0);
      if constexpr (GUARD) fastp = !flagged;
      if (x.t == 0) tiny_flag(S, (k + 1) & 1) = 0; // set again when the next plane is staged (after the barrier)
    }
  }
  const int tx = x.tx, ty = x.ty;
  const bool multi_d = D3 || x.multi_d; // compile-time true in the 3-D instantiation
  const int t = (x.t + 64 * (k % NW)) % NT; // duty index: wave roles rotate with k
#if ARTEMIS_PERI_PRIO
  // Three of the four waves carry a perimeter duty in this plane (slopes on two, the extra Riemann pass on one) and the
  // fourth waits for them at both barriers; each SIMD holds one wave of this workgroup and one of another.  The duty
  // waves therefore run at raised priority until their duties are done: the arbiter prefers them over the co-resident
  // workgroup's wave, the workgroup reaches its barriers sooner, and the other one fills the slots this one leaves
  // while it waits.  -4.8 % on the Sedov headline (0.986 -> 0.938 ms; priority only in the duty passes themselves:
  // -2.2 %; held through the update phase as well: -1 %; the fourth wave at an intermediate level during the sweeps: no
  // change).  Scheduling only: the same bits.
  if (t >= 64) __builtin_amdgcn_s_setprio(2);
#endif
  // ---- P1: slopes of the own cell; perimeter slopes on waves 2 (x1) and 3 (x2) -----------
  Cell6 lox, loy;
#define SLX(m, n)                                                                          \
  {                                                                                        \
    double up_, lo_;                                                                       \
    if constexpr (PG) {                                                                    \
      if (fastp)                                                                           \
        plm_g_shared<2>(S.Q[n][ty + FH][tx + FH - 1], qc.m, S.Q[n][ty + FH][tx + FH + 1],  \
                        up_, lo_, gx.g1);                                                  \
      else                                                                                 \
        plm_g_shared<0>(S.Q[n][ty + FH][tx + FH - 1], qc.m, S.Q[n][ty + FH][tx + FH + 1],  \
                        up_, lo_, gx.g1);                                                  \
    } else {                                                                               \
      const double s_ =                                                                    \
          slope_sel<RECON>(S.Q[n][ty + FH][tx + FH - 1], qc.m, S.Q[n][ty + FH][tx + FH + 1], fastp);  \
      lo_ = lo_val<RECON>(qc.m, s_), up_ = up_val<RECON>(qc.m, s_);                        \
    }                                                                                      \
    lox.m = lo_;                                                                           \
    S.UPX[n][ty][tx + 1] = up_;                                                            \
  }
  SLX(d, 0) SLX(v1, 1) SLX(v2, 2) SLX(v3, 3) SLX(p, 4) SLX(e, 5)
#undef SLX
  if (multi_d) {
#define SLY(m, n)                                                                          \
  {                                                                                        \
    double up_, lo_;                                                                       \
    if constexpr (PG) {                                                                    \
      if (fastp)                                                                           \
        plm_g_shared<2>(S.Q[n][ty + FH - 1][tx + FH], qc.m, S.Q[n][ty + FH + 1][tx + FH],  \
                        up_, lo_, gx.g2);                                                  \
      else                                                                                 \
        plm_g_shared<0>(S.Q[n][ty + FH - 1][tx + FH], qc.m, S.Q[n][ty + FH + 1][tx + FH],  \
                        up_, lo_, gx.g2);                                                  \
    } else {                                                                               \
      const double s_ =                                                                    \
          slope_sel<RECON>(S.Q[n][ty + FH - 1][tx + FH], qc.m, S.Q[n][ty + FH + 1][tx + FH], fastp);  \
      lo_ = lo_val<RECON>(qc.m, s_), up_ = up_val<RECON>(qc.m, s_);                        \
    }                                                                                      \
    loy.m = lo_;                                                                           \
    S.UPY[n][ty + 1][tx] = up_;                                                            \
  }
    SLY(d, 0) SLY(v1, 1) SLY(v2, 2) SLY(v3, 3) SLY(p, 4) SLY(e, 5)
#undef SLY
  }
  if (t >= 128 && t < 128 + 2 * FTY) { // cells i0-1 (upper value) and i0+32 (lower value)
    const int u = t - 128, row = u >> 1, side = u & 1;
    const int cx = side ? FTX + FH : FH - 1;
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const double q = S.Q[n][row + FH][cx];
      double up_, lo_;
      if constexpr (PG) {
        if (fastp) plm_g_shared<2>(S.Q[n][row + FH][cx - 1], q, S.Q[n][row + FH][cx + 1], up_, lo_, S.GX1[side]);
        else plm_g_shared<0>(S.Q[n][row + FH][cx - 1], q, S.Q[n][row + FH][cx + 1], up_, lo_, S.GX1[side]);
      } else {
        const double s_ = slope_sel<RECON>(S.Q[n][row + FH][cx - 1], q, S.Q[n][row + FH][cx + 1], fastp);
        lo_ = lo_val<RECON>(q, s_), up_ = up_val<RECON>(q, s_);
      }
      if (side) S.LOX[n][row] = lo_;
      else S.UPX[n][row][0] = up_;
    }
  }
  if (multi_d && t >= 192 && t < 256) { // rows j0-1 (upper value) and j0+8 (lower value)
    const int u = t - 192, cx = u & 31, side = u >> 5;
    const int ry = side ? FTY + FH : FH - 1;
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const double q = S.Q[n][ry][cx + FH];
      double up_, lo_;
      if constexpr (PG) {
        if (fastp) plm_g_shared<2>(S.Q[n][ry - 1][cx + FH], q, S.Q[n][ry + 1][cx + FH], up_, lo_, S.GX2[side][cx]);
        else plm_g_shared<0>(S.Q[n][ry - 1][cx + FH], q, S.Q[n][ry + 1][cx + FH], up_, lo_, S.GX2[side][cx]);
      } else {
        const double s_ = slope_sel<RECON>(S.Q[n][ry - 1][cx + FH], q, S.Q[n][ry + 1][cx + FH], fastp);
        lo_ = lo_val<RECON>(q, s_), up_ = up_val<RECON>(q, s_);
      }
      if (side) S.LOY[n][cx] = lo_;
      else S.UPY[n][0][cx] = up_;
    }
  }
  FPROF(1);
  __syncthreads();
  FPROF(2);
  // ---- P2: Riemann problems at the own lower faces; perimeter faces on wave 1 ------------
  Cell6 L;
  GET6(L, S.UPX, [ty][tx]);
  fx_lo = solve_face<RIEMANN, 1>(x.gk, L, lox, fastp);
  if constexpr (CURV) fx_lo.m2 *= gx.h1[1], fx_lo.m3 *= gx.h1[2]; // ScaleMomentumFlux (h1 == 1)
  if (tx > 0) { PUT8(S.FX, fx_lo, [ty][tx - 1]); }
  fy_lo = fx_lo;
  if (multi_d) {
    GET6(L, S.UPY, [ty][tx]);
    fy_lo = solve_face<RIEMANN, 2>(x.gk, L, loy, fastp);
    if constexpr (CURV) fy_lo.m2 *= gx.h2[1], fy_lo.m3 *= gx.h2[2];
    if (ty > 0) { PUT8(S.FY, fy_lo, [ty - 1][tx]); }
  }
  if (t >= 64 && t < 128) { // lanes 0..7: x1 face i0+32 per row; lanes 32..63: x2 face j0+8
    // ONE Riemann pass for both kinds of perimeter face: the x2 lanes hand the solver their states with the velocity
    // components rotated (v2, v3, v1) -- which is what solve_face<.., 2> does internally (hllc.hpp:67-69) -- and rotate
    // the momentum fluxes back, so the duty wave runs three solver passes per plane instead of four (same bits).
    const int u = t - 64;
    const bool isx = (u < FTY), isy = multi_d && (u >= 32);
    if (isx || isy) {
      const int cx = u - 32;
      Cell6 l, r;
      if (isx) {
        GET6(l, S.UPX, [u][FTX]);
        GET6(r, S.LOX, [u]);
      } else {
        GET6(l, S.UPY, [FTY][cx]);
        GET6(r, S.LOY, [cx]);
        double a_ = l.v1;
        l.v1 = l.v2, l.v2 = l.v3, l.v3 = a_;
        a_ = r.v1;
        r.v1 = r.v2, r.v2 = r.v3, r.v3 = a_;
      }
      Flux8 fe_ = solve_face<RIEMANN, 1>(x.gk, l, r, fastp);
      if (isx) {
        if constexpr (CURV) fe_.m2 *= S.HF1[u][0], fe_.m3 *= S.HF1[u][1]; // the face below cell (j0+u, i0+32)
        PUT8(S.FX, fe_, [u][FTX - 1]);
      } else {
        const double n_ = fe_.m1; // (normal, t1, t2) = (m2, m3, m1) of the block's frame
        fe_.m1 = fe_.m3, fe_.m3 = fe_.m2, fe_.m2 = n_;
        if constexpr (CURV) fe_.m2 *= S.HF2[cx][0], fe_.m3 *= S.HF2[cx][1]; // the face below cell (j0+8, i0+cx)
        PUT8(S.FY, fe_, [FTY - 1][cx]);
      }
    }
  }
#if ARTEMIS_PERI_PRIO
  __builtin_amdgcn_s_setprio(0); // the duties are done
#endif
  FPROF(9);
  if (stage_next) {
    stage_plane(S, x, qn, hal_next);
    if constexpr (GUARD || DETECT) {
      if (GUARD || det) stage_plane_flag(S, x, qn, hal_next, (k + 1) & 1);
    }
  }
  FPROF(3);
  __syncthreads();
  FPROF(4);
}

// Phase P3 of the flux TASK (artemis_hip_calculate_fluxes through the tile march, FLUXES = true): instead of
// updating the zone, store what CalculateFluxesImpl stores (fluid_fluxes.hpp:78-213) -- the eight outputs of the
// zone's lower x1 / x2 / x3 faces, and of the block's last face of a direction from the zone next to it.  Faces
// [s, e+1] of each direction over the active extent of the others, every face exactly once.
template <bool D3, class TILE>
ADEV void plane_store_fluxes(TILE &S, const PackView &P, const Ctx &x, const int k, const Flux8 &fx_lo,
                             const Flux8 &fy_lo, const Flux8 &fz_lo, const Flux8 &fz_hi) {
  if (!x.active) return;
  const int tx = x.tx, ty = x.ty;
  const bool multi_d = D3 || x.multi_d;
  const FluidView &f = P.gas;
  const long c = static_cast<long>(x.col + static_cast<unsigned>(k) * x.sk);
  const int b6 = x.b * 6;
  auto put = [&](int d, const Flux8 &fl, long at) {
    f.flux[d][b6 + 0][at] = fl.d;
    f.flux[d][b6 + 1][at] = fl.m1, f.flux[d][b6 + 2][at] = fl.m2, f.flux[d][b6 + 3][at] = fl.m3;
    f.flux[d][b6 + 4][at] = fl.e;  // IEN shares the IPR slot (hllc.hpp:72)
    f.flux[d][b6 + 5][at] = fl.eg; // IEG shares the ISE slot (hllc.hpp:73)
    f.pflux[d][x.b][at] = fl.pf;
    f.vface[d][x.b][at] = fl.vf;
  };
  put(0, fx_lo, c);
  if (x.i0 + tx == P.ie) {
    Flux8 hi;
    GET8(hi, S.FX, [ty][tx]);
    put(0, hi, c + 1);
  }
  if (multi_d) {
    put(1, fy_lo, c);
    if (x.j0 + ty == P.je) {
      Flux8 hi;
      GET8(hi, S.FY, [ty][tx]);
      put(1, hi, c + x.sj);
    }
  }
  if constexpr (D3) {
    put(2, fz_lo, c);
    if (k == P.ke) put(2, fz_hi, c + x.sk);
  }
}

// Phase P3: gather the upper-face fluxes published by the neighbours, then the whole per-cell
// chain update -> sources -> aux -> c2p -> p2c -> store (and the CFL reduction).
// `bad`: the zone's stencil holds a tiny velocity (the caller's plane flag and own-column bits).  Such a zone -- and one
// whose updated momenta come out tiny -- keeps its hands off memory: no store, no contribution to dt; its id goes to
// the redo list and stage_redo_kernel computes it with IEEE arithmetic.
ADEV void redo_append(const StageK &a, const PackView &P, int b, unsigned c, bool shell_wg) {
  unsigned *cnt = shell_wg ? a.redo_cnt1 : a.redo_cnt0;
  unsigned long long *list = shell_wg ? a.redo_list1 : a.redo_list0;
  const unsigned at = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (at < a.redo_cap)
    list[at] = static_cast<unsigned long long>(b) * (static_cast<unsigned long long>(P.nk) * P.nj * P.ni) + static_cast<unsigned long long>(c);
}
template <bool HAS_U1, bool WRITE_CONS, bool WITH_DT, bool D3, class TILE>
ADEV void plane_update(TILE &S, const PackView &P, const StageK &a, const Ctx &x, const int k,
                       const Cell6 &qc, const Flux8 &fx_lo, const Flux8 &fy_lo, const Flux8 &fz_lo,
                       const Flux8 &fz_hi, const Raw5 &u1raw, double &ldt, const bool bad_in, const bool shell_wg) {
  const int tx = x.tx, ty = x.ty;
  const bool multi_d = D3 || x.multi_d;
  constexpr bool three_d = D3;
  const double gm1 = x.gm1;
  const FluidView &f = P.gas;
  Flux8 fx_hi, fy_hi = fx_lo;
  GET8(fx_hi, S.FX, [ty][tx]);
  if (multi_d) { GET8(fy_hi, S.FY, [ty][tx]); }
  if (!x.active) return;
  const double *g = x.g + opaque(4); // (scalar loads at their use: fused_device.hpp kload)
  const double dx1 = x.dx1, dx2 = x.dx2;
  const double g4 = kload(g), g5 = kload(g + 1);
  const double dx3 = (g4 + (k + 1) * g5) - (g4 + k * g5);
  const double ax1 = dx2 * dx3, ax2 = dx1 * dx3, ax3 = dx1 * dx2; // geometry.hpp:199-216
  const double vol = dx1 * dx2 * dx3;                             // geometry.hpp:219-225
  const int b = x.b;
  const unsigned c = x.col + static_cast<unsigned>(k) * x.sk;
  // u0 = PrimToCons(prim_in), u1 = PrimToCons(prim_u1)  (fill_derived.cpp:226-255)
  const double D0 = qc.d;
  const double M10 = qc.d * qc.v1 * 1.0, M20 = qc.d * qc.v2 * 1.0, M30 = qc.d * qc.v3 * 1.0;
  const double G0 = qc.e * qc.d;
  const double E0 = G0 + 0.5 * qc.d * (sqr(qc.v1) + sqr(qc.v2) + sqr(qc.v3));
  double D1 = D0, M11 = M10, M21 = M20, M31 = M30, G1 = G0, E1 = E0;
  if constexpr (HAS_U1) {
    const double r1 = u1raw.d, e1 = u1raw.e, a1 = u1raw.v1, a2 = u1raw.v2, a3 = u1raw.v3;
    D1 = r1, M11 = r1 * a1 * 1.0, M21 = r1 * a2 * 1.0, M31 = r1 * a3 * 1.0;
    G1 = e1 * r1;
    E1 = G1 + 0.5 * r1 * (sqr(a1) + sqr(a2) + sqr(a3));
  }
  // ApplyUpdate (artemis_integrator.hpp:88-106); the six divisions by the cell volume and the
  // three bdt/vol of FluxSource share one refined reciprocal.
  const Recip rvol = recip(vol);
  const Recip rdx3 = recip(dx3);
  auto upd = [&](double u0, double u1, double f1l, double f1h, double f2l, double f2h, double f3l,
                 double f3h) {
    double divf = (ax1 * f1l - ax1 * f1h);
    if (multi_d) divf += (ax2 * f2l - ax2 * f2h);
    if (three_d) divf += (ax3 * f3l - ax3 * f3h);
    return a.gam0 * u0 + a.gam1 * u1 + div(divf * x.beta_dt, rvol);
  };
  const double D = upd(D0, D1, fx_lo.d, fx_hi.d, fy_lo.d, fy_hi.d, fz_lo.d, fz_hi.d);
  double M1 = upd(M10, M11, fx_lo.m1, fx_hi.m1, fy_lo.m1, fy_hi.m1, fz_lo.m1, fz_hi.m1);
  double M2 = upd(M20, M21, fx_lo.m2, fx_hi.m2, fy_lo.m2, fy_hi.m2, fz_lo.m2, fz_hi.m2);
  double M3 = upd(M30, M31, fx_lo.m3, fx_hi.m3, fy_lo.m3, fy_hi.m3, fz_lo.m3, fz_hi.m3);
  const double E = upd(E0, E1, fx_lo.e, fx_hi.e, fy_lo.e, fy_hi.e, fz_lo.e, fz_hi.e);
  double G = upd(G0, G1, fx_lo.eg, fx_hi.eg, fy_lo.eg, fy_hi.eg, fz_lo.eg, fz_hi.eg);
  // FluxSource (fluid_fluxes.hpp:365-392)
  const double bdt_vol = div(x.bdt, rvol);
  M1 += div(x.bdt, x.rdx1) * (fx_lo.pf - fx_hi.pf);
  G -= bdt_vol * 0.5 * (fx_lo.pf + fx_hi.pf) * (ax1 * fx_hi.vf - ax1 * fx_lo.vf);
  if (multi_d) {
    M2 += div(x.bdt, x.rdx2) * (fy_lo.pf - fy_hi.pf);
    G -= bdt_vol * 0.5 * (fy_lo.pf + fy_hi.pf) * (ax2 * fy_hi.vf - ax2 * fy_lo.vf);
  }
  if (three_d) {
    M3 += div(x.bdt, rdx3) * (fz_lo.pf - fz_hi.pf);
    G -= bdt_vol * 0.5 * (fz_lo.pf + fz_hi.pf) * (ax3 * fz_hi.vf - ax3 * fz_lo.vf);
  }
  // SetAuxillaryFields (fill_derived.cpp:54-73, artemis_utils.hpp:43-62) and ConsToPrim
  // (fill_derived.cpp:137-151) divide six times by the same floored density.
  const double w_d = (D > f.dfloor) ? D : f.dfloor; // == max(D, dfloor) of artemis_utils.hpp:49
  const Recip rd = recip(w_d);
  {
    const double ke = div(0.5 * (sqr(M1 / 1.0) + sqr(M2 / 1.0) + sqr(M3 / 1.0)), rd);
    const double ue = E - ke;
    double sie = div((ue > f.de_switch * E) ? ue : G, rd);
    sie = amax(sie, f.siefloor);
    G = sie * w_d;
    const double uflr = f.siefloor * w_d;
    G = (G > uflr) ? G : uflr;
  }
  if (a.redo_cnt0 && (bad_in || tiny_mom3(M1, M2, M3))) {
    redo_append(a, P, b, c, shell_wg);
    return;
  }
  // the PrimToCons floors that follow (fill_derived.cpp:229,244) are idempotent here
  const double w1 = div(M1, rd), w2 = div(M2, rd), w3 = div(M3, rd);
  double w_s = div(G, rd);
  w_s = (w_s > f.siefloor) ? w_s : f.siefloor;
  const double w_p = amax(0.0, gm1 * w_d * w_s); // fill_derived.cpp:247
  if (a.tiny_out) { // report a vanishing velocity to the stage that will read this state (artemis_stage_args_t.tiny_out)
    if (tiny_vel3(w1, w2, w3)) __hip_atomic_fetch_or(a.tiny_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  double *const *po = a.prim_out + opaque(b * 6);
  gst(kload(po + 0), c, w_d);
  gst(kload(po + 1), c, w1);
  gst(kload(po + 2), c, w2);
  gst(kload(po + 3), c, w3);
  gst(kload(po + 4), c, w_p);
  gst(kload(po + 5), c, w_s);
  if constexpr (WRITE_CONS) { // PrimToCons (fill_derived.cpp:226-255)
    const double u_u = w_s * w_d;
    double *const *co = a.cons_out + opaque(b * 6);
    gst(kload(co + 0), c, w_d);
    gst(kload(co + 1), c, w_d * w1 * 1.0);
    gst(kload(co + 2), c, w_d * w2 * 1.0);
    gst(kload(co + 3), c, w_d * w3 * 1.0);
    gst(kload(co + 4), c, u_u + 0.5 * w_d * (sqr(w1) + sqr(w2) + sqr(w3)));
    gst(kload(co + 5), c, u_u);
  }
  if constexpr (WITH_DT) { // Gas::EstimateTimestepMesh on the new state (gas.cpp:411-433)
    const double bulk = (gm1 + 1.0) * gm1 * w_d * w_s;
    const double cs = sqrt_pos(div(bulk, rd));
    double denom = 0.0;
    denom += div(fabs(w1) + cs, x.rdx1); // 1.0*dx == dx
    if (multi_d) denom += div(fabs(w2) + cs, x.rdx2);
    if (three_d) denom += div(fabs(w3) + cs, rdx3);
    ldt = amin(ldt, div(1.0, denom));
  }
}

// The cell's diffusion fluxes (one species: 3 momentum components + energy through the lower / upper face of
// each direction), fetched at the top of a trip so that their latency hides behind the plane's sweeps.
struct DFlux24 {
  double lo[3][4], hi[3][4];
  double sum[5]; // ... or the five sums DiffusionUpdate subtracts, precomputed by artemis_hip_viscous_source
};
ADEV DFlux24 load_dsum(double *const *dsum, int b, unsigned c) {
  DFlux24 r{};
#pragma unroll
  for (int q = 0; q < 5; ++q) r.sum[q] = gld(dsum[b * 5 + q], c);
  return r;
}
ADEV DFlux24 load_dflux(const PackView &P, int b, long c, bool multi_d, bool three_d) {
  DFlux24 r;
  const long up[3] = {c + 1, c + (multi_d ? P.sj : 0), c + (three_d ? P.sk : 0)};
  const int dd[3] = {0, multi_d ? 1 : 0, three_d ? 2 : 0}; // inactive directions: a finite stand-in (times 0)
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double *q = P.gas.dflux[dd[d]][b * 4 + v];
      r.lo[d][v] = gld(q, c), r.hi[d][v] = gld(q, up[d]);
    }
  return r;
}

// Phase P3 of the curvilinear instantiations: the per-cell chain of the cell-centred general stage
// (kernels_stage_cell.hip, same device functions, same order) fed with the tile's face fluxes:
// ApplyUpdate -> FluxSource (pressure + coordinate sources) -> DiffusionUpdate -> ExternalGravity ->
// RotatingFrameImpl -> SetAuxillaryFields -> ConsToPrim (-> EstimateTimestepMesh of the new state).
template <bool HAS_U1, bool WITH_DT, bool D3, class TILE>
ADEV void plane_update_curv(TILE &S, const PackView &P, const StageK &a, const SrcK &sk, const Ctx &x,
                            const GeoCtx<true> &gx, const int k, const Cell6 &qc, const Flux8 &fx_lo,
                            const Flux8 &fy_lo_in, const Flux8 &fz_lo_in, const Flux8 &fz_hi_in,
                            const Raw5 &u1raw, const DFlux24 &df, double &ldt) {
  const int tx = x.tx, ty = x.ty;
  const bool multi_d = D3 || x.multi_d;
  constexpr bool three_d = D3;
  Flux8 fx_hi, fy_hi, fy_lo = fy_lo_in, fz_lo = fz_lo_in, fz_hi = fz_hi_in;
  GET8(fx_hi, S.FX, [ty][tx]);
  fy_hi = fx_hi;
  if (multi_d) { GET8(fy_hi, S.FY, [ty][tx]); }
  if (!x.active) return;
  const FluidView &f = P.gas;
  const int b = x.b;
  const unsigned c = x.col + static_cast<unsigned>(k) * x.sk;
  DCoords co = gx.co; // (c3 / s3 of this plane were fetched at the top of the trip)
  co.x3[0] = x.g[4] + k * x.g[5], co.x3[1] = x.g[4] + (k + 1) * x.g[5];
  const CellMetric g = cell_metric_of(co);
  double hx[3];
  scale_factors_of(co, hx);
  FluidPrim w;
  w.rho = qc.d, w.v1 = qc.v1, w.v2 = qc.v2, w.v3 = qc.v3, w.sie = qc.e;
  GasCons u0 = prim_to_cons_gas(f, w.rho, w.v1, w.v2, w.v3, w.sie, hx);
  GasCons u1 = u0;
  if constexpr (HAS_U1) u1 = prim_to_cons_gas(f, u1raw.d, u1raw.v1, u1raw.v2, u1raw.v3, u1raw.e, hx);
  // ---- ApplyUpdate (artemis_integrator.hpp:88-106)
  const Recip rvol = recip(g.vol);
  // Momenta can be tiny-but-nonzero ahead of a shock, where only IEEE division is right (see plm_g_shared): the
  // wave checks its three momentum numerators once and takes `/` for them if any lane needs it.
  auto flux_div = [&](double f1l, double f1h, double f2l, double f2h, double f3l, double f3h) {
    double divf = (g.ax1[0] * f1l - g.ax1[1] * f1h);
    if (multi_d) divf += (g.ax2[0] * f2l - g.ax2[1] * f2h);
    if (three_d) divf += (g.ax3[0] * f3l - g.ax3[1] * f3h);
    return divf * x.beta_dt;
  };
  const double nd = flux_div(fx_lo.d, fx_hi.d, fy_lo.d, fy_hi.d, fz_lo.d, fz_hi.d);
  const double n1m = flux_div(fx_lo.m1, fx_hi.m1, fy_lo.m1, fy_hi.m1, fz_lo.m1, fz_hi.m1);
  const double n2m = flux_div(fx_lo.m2, fx_hi.m2, fy_lo.m2, fy_hi.m2, fz_lo.m2, fz_hi.m2);
  const double n3m = flux_div(fx_lo.m3, fx_hi.m3, fy_lo.m3, fy_hi.m3, fz_lo.m3, fz_hi.m3);
  const double ne = flux_div(fx_lo.e, fx_hi.e, fy_lo.e, fy_hi.e, fz_lo.e, fz_hi.e);
  const double neg = flux_div(fx_lo.eg, fx_hi.eg, fy_lo.eg, fy_hi.eg, fz_lo.eg, fz_hi.eg);
  double q1m, q2m, q3m;
  if (__any(tiny_nonzero(n1m) || tiny_nonzero(n2m) || tiny_nonzero(n3m))) {
    q1m = n1m / g.vol, q2m = n2m / g.vol, q3m = n3m / g.vol;
  } else {
    q1m = div(n1m, rvol), q2m = div(n2m, rvol), q3m = div(n3m, rvol);
  }
  u0.d = a.gam0 * u0.d + a.gam1 * u1.d + div(nd, rvol);
  u0.m1 = a.gam0 * u0.m1 + a.gam1 * u1.m1 + q1m;
  u0.m2 = a.gam0 * u0.m2 + a.gam1 * u1.m2 + q2m;
  u0.m3 = a.gam0 * u0.m3 + a.gam1 * u1.m3 + q3m;
  u0.e = a.gam0 * u0.e + a.gam1 * u1.e + div(ne, rvol);
  u0.eg = a.gam0 * u0.eg + a.gam1 * u1.eg + div(neg, rvol);
  // ---- FluxSource (fluid_fluxes.hpp:361-415); divisions through refined reciprocals (device_math.hpp: the
  // bits of `/`), the ones of (i, j)-only denominators being constants of the march
  const double dt = x.bdt;
  const double dt_vol = div(dt, rvol);
  u0.m1 += div(dt, g.dx[0]) * (fx_lo.pf - fx_hi.pf);
  u0.eg -= dt_vol * 0.5 * (fx_lo.pf + fx_hi.pf) * (g.ax1[1] * fx_hi.vf - g.ax1[0] * fx_lo.vf);
  if (multi_d) {
    u0.m2 += div(dt, g.dx[1]) * (fy_lo.pf - fy_hi.pf);
    u0.eg -= dt_vol * 0.5 * (fy_lo.pf + fy_hi.pf) * (g.ax2[1] * fy_hi.vf - g.ax2[0] * fy_lo.vf);
  }
  if (three_d) {
    u0.m3 += div(dt, g.dx[2]) * (fz_lo.pf - fz_hi.pf);
    u0.eg -= dt_vol * 0.5 * (fz_lo.pf + fz_hi.pf) * (g.ax3[1] * fz_hi.vf - g.ax3[0] * fz_lo.vf);
  }
  {
    const double rdt = w.rho * dt;
    double vf[3];
    rotation_velocity(co, P.omf, vf);
    if (co.x1dep())
      u0.m1 += rdt * (0.0 * sqr(w.v1 + vf[0]) + co.dh2dx1() * sqr(w.v2 + vf[1]) + co.dh3dx1() * sqr(w.v3 + vf[2]));
    if (co.x2dep() && multi_d)
      u0.m2 += rdt * (0.0 * sqr(w.v1 + vf[0]) + 0.0 * sqr(w.v2 + vf[1]) + co.dh3dx2() * sqr(w.v3 + vf[2]));
  }
  if (sk.diff_on) { // Gas::DiffusionUpdate (artemis_driver.cpp:218-221)
    double dm[3], de, deg;
    if (sk.dsum) {
      dm[0] = df.sum[0], dm[1] = df.sum[1], dm[2] = df.sum[2], de = df.sum[3], deg = df.sum[4];
    } else {
      const DiffCell dcell = diffusion_cell_of(co, g, hx, P.ndim);
      const double v[3] = {w.v1, w.v2, w.v3};
      auto F = [&](int d, int var, int u) { return u ? df.hi[d][var] : df.lo[d][var]; };
      diffusion_update_core(dcell, F, 0, 1, sk.do_viscosity, dt, v, dm, de, deg);
    }
    u0.m1 -= dm[0], u0.m2 -= dm[1], u0.m3 -= dm[2];
    u0.e -= de;
    u0.eg -= deg;
  }
  if (sk.grav_on) gravity_gas(gravity_accel(sk.grav, co, P.ndim, dt), dt, hx, w, u0);
  if (sk.rfc_on) {
    const RotFrame rfc = rotating_frame_terms(co, sk.rf_omega, dt);
    const double mlo[3] = {fx_lo.d, multi_d ? fy_lo.d : 0.0, three_d ? fz_lo.d : 0.0};
    const double mup[3] = {fx_hi.d, multi_d ? fy_hi.d : 0.0, three_d ? fz_hi.d : 0.0};
    const double ax2[2] = {multi_d ? g.ax2[0] : 0.0, multi_d ? g.ax2[1] : 0.0};
    const double ax3[2] = {three_d ? g.ax3[0] : 0.0, three_d ? g.ax3[1] : 0.0};
    rotating_frame_gas(rfc, multi_d, three_d, mlo, mup, g.ax1, ax2, ax3, g.vol, u0);
  }
  // ---- SetAuxillaryFields (fill_derived.cpp:58-71) + ConsToPrim (:132-146)
  const double w_d = (u0.d > f.dfloor) ? u0.d : f.dfloor;
  const double u_d2 = amax(u0.d, f.dfloor);
  const Recip rd2 = recip(u_d2);
  const bool tiny_m = __any(tiny_nonzero(u0.m1) || tiny_nonzero(u0.m2) || tiny_nonzero(u0.m3)); // momenta: see above
  double rv2, rv3;
  if (tiny_m) rv2 = u0.m2 / hx[1], rv3 = u0.m3 / hx[2];
  else rv2 = div(u0.m2, hx[1]), rv3 = div(u0.m3, hx[2]);
  const double rv1 = u0.m1 / 1.0; // hx[0] == 1
  const double ke = div(0.5 * (sqr(rv1) + sqr(rv2) + sqr(rv3)), rd2);
  const double ue_cons = u0.e - ke;
  double sie = (ue_cons > f.de_switch * u0.e) ? div(ue_cons, rd2) : div(u0.eg, rd2);
  sie = amax(sie, f.siefloor);
  double u_u = sie * w_d;
  const double uflr = f.siefloor * w_d;
  u_u = (u_u > uflr) ? u_u : uflr;
  const Recip rwd = recip(w_d);
  double n1, n2, n3;
  if (tiny_m) n1 = u0.m1 / (w_d * hx[0]), n2 = u0.m2 / (w_d * hx[1]), n3 = u0.m3 / (w_d * hx[2]);
  else n1 = div(u0.m1, rwd), n2 = div(u0.m2, w_d * hx[1]), n3 = div(u0.m3, w_d * hx[2]); // w_d * 1.0 == w_d
  double w_s = div(u_u, rwd);
  w_s = (w_s > f.siefloor) ? w_s : f.siefloor;
  gst(a.prim_out[b * 6 + 0], c, w_d);
  gst(a.prim_out[b * 6 + 1], c, n1);
  gst(a.prim_out[b * 6 + 2], c, n2);
  gst(a.prim_out[b * 6 + 3], c, n3);
  gst(a.prim_out[b * 6 + 4], c, amax(0.0, x.gm1 * w_d * w_s)); // fill_derived.cpp:247 (consumers recompute it anyway)
  gst(a.prim_out[b * 6 + 5], c, w_s);
  if constexpr (WITH_DT) { // Gas::EstimateTimestepMesh on the new state (gas.cpp:411-433)
    const double bulk = (x.gm1 + 1.0) * x.gm1 * w_d * w_s;
    const double cs = sqrt_pos(div(bulk, rwd));
    double denom = div(fabs(n1) + cs, co.width1());
    if (multi_d) denom += div(fabs(n2) + cs, co.width2());
    if (three_d) denom += div(fabs(n3) + cs, co.width3());
    ldt = amin(ldt, div(1.0, denom));
  }
}

template <int RIEMANN, int RECON, bool HAS_U1, bool WRITE_CONS, bool WITH_DT, bool D3, bool CURV = false, bool FLUXES = false>
__global__ __launch_bounds__(NT, (CURV && !ARTEMIS_CURV_OCC2) ? 1 : 2) void stage_fused_kernel(const PackView P, const StageK a,
                                                                       const SrcArg<CURV> src) {
  __shared__ std::conditional_t<CURV, LdsTileCurv, LdsTile> S;
  Ctx x;
  x.tx = threadIdx.x, x.ty = threadIdx.y, x.t = x.ty * FTX + x.tx;
  int id = blockIdx.x;
  const bool shell_wg = static_cast<int>(blockIdx.x) < a.nshell;
  if (a.xcd_swizzle && id >= a.nshell) {
    // Workgroup ids are dealt round-robin over the 8 XCDs, each with its own L2.  Give every XCD one
    // contiguous run of the bulk tiles (x1 fastest, then x2, then x3 chunks), so the tiles that share
    // halo zones and the cache lines a 32-zone row straddles are resident on the same L2.  A bijection
    // of the bulk ids; shell workgroups keep the lowest ids (the comm stream waits for them).
    const int r = id - a.nshell, nbulk = static_cast<int>(gridDim.x) - a.nshell;
    const int q = nbulk >> 3, rem = nbulk & 7, xcd = r & 7;
    id = a.nshell + xcd * q + min(xcd, rem) + (r >> 3);
  }
  // box lookup with static indices only (a dynamically indexed by-value argument would be
  // spilled to scratch); everything here is wave-uniform scalar work
  int bstart = a.start[0], bti0 = a.ti0[0], bnti = a.nti[0], btj0 = a.tj0[0], bntj = a.ntj[0];
  int bkb0 = a.kb0[0], bkb1 = a.kb1[0], bnchunk = a.nchunk[0], bkchunk = a.kchunk[0];
#pragma unroll
  for (int r = 1; r < 7; ++r) {
    if (r < a.nbox && id >= a.start[r]) {
      bstart = a.start[r], bti0 = a.ti0[r], bnti = a.nti[r], btj0 = a.tj0[r], bntj = a.ntj[r];
      bkb0 = a.kb0[r], bkb1 = a.kb1[r], bnchunk = a.nchunk[r], bkchunk = a.kchunk[r];
    }
  }
  int local = id - bstart;
  const int ti = bti0 + local % bnti;
  local /= bnti;
  const int tj = btj0 + local % bntj;
  local /= bntj;
  const int chunk = local % bnchunk;
  x.b = local / bnchunk;
  x.three_d = D3, x.multi_d = D3 || P.ndim > 1; // D3: the 3-D march; otherwise one plane (1-D / 2-D blocks)
  x.i0 = P.is + ti * FTX, x.j0 = P.js + tj * FTY;
  const int i = x.i0 + x.tx, j = x.j0 + x.ty;
  x.active = (i <= P.ie) && (j <= P.je);
  // clamped indices: inactive lanes still serve as neighbours and face owners
  int il = min(i, P.ni - 1);
  int jl = min(j, P.nj - 1);
  const int outflow = outflow_of(a.outflow, a.outflow_blk, a.outflow_pb, x.b); // (wave-uniform: the block is)
  if (outflow & 8) jl = min(jl, P.je);
  if (outflow & 2) il = min(il, P.ie); // (ragged tiles: the lanes past the last zone stand in for the outflow ghosts)
  const int k0 = bkb0 + chunk * bkchunk;
  const int k1 = min(bkb1, k0 + bkchunk - 1);
  if (k0 > k1) { // empty chunk (cannot happen with the box builder, kept for safety)
    if (shell_wg && x.t == 0)
      __hip_atomic_fetch_add(a.shell_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  x.gm1 = P.gm1;
  x.gk = gas_constants(P.gm1);
  x.beta_dt = a.beta_dt, x.bdt = a.bdt;
  if (a.bdt_ptr) x.beta_dt = x.bdt = *a.bdt_ptr; // dt lives on the device: scalar load
  x.g = P.geom + 6 * x.b;
  x.in_r = a.prim_in[x.b * 6 + 0], x.in_1 = a.prim_in[x.b * 6 + 1];
  x.in_2 = a.prim_in[x.b * 6 + 2], x.in_3 = a.prim_in[x.b * 6 + 3];
  x.in_e = a.prim_in[x.b * 6 + 5];
  x.sj = static_cast<unsigned>(P.sj), x.sk = static_cast<unsigned>(P.sk);
  x.col = static_cast<unsigned>(jl) * x.sj + static_cast<unsigned>(il);
  // geometry.hpp:65-72: widths along x1 and x2 do not change along the march
  x.dx1 = (x.g[0] + (i + 1) * x.g[1]) - (x.g[0] + i * x.g[1]);
  x.dx2 = (x.g[2] + (j + 1) * x.g[3]) - (x.g[2] + j * x.g[3]);
  x.rdx1 = recip(x.dx1), x.rdx2 = recip(x.dx2);
  // halo duty: threads 0..127 stage the x2 halo rows (Q rows 0,1,10,11), threads 128..159 the x1
  // halo columns (Q cols 0,1,34,35); each owns one halo column for the whole march
  x.hr = -1, x.hc = -1, x.hcol = 0;
  if (x.t < 4 * FTX) {
    const int rr = x.t >> 5;
    x.hr = (rr < 2) ? rr : FTY + rr;
    x.hc = (x.t & 31) + FH;
  } else if (x.t >= 128 && x.t < 128 + 4 * FTY) {
    const int u = x.t - 128, cc = u & 3;
    x.hr = (u >> 2) + FH;
    x.hc = (cc < 2) ? cc : FTX + cc;
  }
  if (x.hr >= 0) {
    int gi = min(max(x.i0 - FH + x.hc, 0), P.ni - 1);
    if (outflow & 1) gi = max(gi, P.is); // outflow: the ghost zones hold the edge zone's value -- stage that
    if (outflow & 2) gi = min(gi, P.ie);
    int gj = min(max(x.j0 - FH + x.hr, 0), P.nj - 1);
    if (outflow & 4) gj = max(gj, P.js);
    if (outflow & 8) gj = min(gj, P.je);
    x.hcol = static_cast<unsigned>(gj) * x.sj + static_cast<unsigned>(gi);
  }
  double ldt = DBL_MAX;
  FPROF_DECL;
  GeoCtx<CURV> gx;
  if constexpr (CURV) {
    // geometry of the own column; table rows are read with indices clamped into the block (lanes beyond
    // the last face owner compute unused values)
    const int jg = max(1, min(jl, P.nj - 2));
    gx.co = make_coords(P, x.b, k0, jl, il);
    gx.co.face_scale(1, gx.h1), gx.co.face_scale(2, gx.h2), gx.co.face_scale(3, gx.h3);
    gx.m3 = nullptr;
    if (P.metric && (P.coords == ARTEMIS_SPHERICAL3D || P.coords == ARTEMIS_AXISYMMETRIC))
      gx.m3 = P.metric + x.b * metric_block_stride(P.nj, P.nk) + static_cast<long>(MT_ROWS) * (P.nj + 1);
    { // perimeter faces' scale factors (no x3 dependence: geometry_core.hpp face_scale)
      double hs[3];
      if (x.t < FTY) {
        make_coords(P, x.b, k0, min(x.j0 + x.t, P.nj - 1), min(x.i0 + FTX, P.ni - 1)).face_scale(1, hs);
        S.HF1[x.t][0] = hs[1], S.HF1[x.t][1] = hs[2];
      } else if (x.t >= 64 && x.t < 64 + FTX) {
        make_coords(P, x.b, k0, min(x.j0 + FTY, P.nj - 1), min(x.i0 + (x.t - 64), P.ni - 1)).face_scale(2, hs);
        S.HF2[x.t - 64][0] = hs[1], S.HF2[x.t - 64][1] = hs[2];
      }
    }
    if constexpr (RECON == 1) {
      gx.g1 = compact(plm_geo(P, x.b, 1, k0, jl, il));
      gx.g2 = gx.g1;
      if (x.multi_d) gx.g2 = compact(plm_geo(P, x.b, 2, k0, jg, il));
      // perimeter records (read after the first barrier below)
      if (x.t < 2) S.GX1[x.t] = compact(plm_geo(P, x.b, 1, k0, jl, x.t ? min(x.i0 + FTX, P.ni - 2) : x.i0 - 1));
      if (x.multi_d && x.t >= 64 && x.t < 128) {
        const int u = x.t - 64, side = u >> 5, cx = u & 31;
        const int jr = max(1, min(side ? x.j0 + FTY : x.j0 - 1, P.nj - 2));
        S.GX2[side][cx] = compact(plm_geo(P, x.b, 2, k0, jr, min(x.i0 + cx, P.ni - 1)));
      }
    }
  }
  // the plane flags: PLM_G's cubic numerators need them in the curvilinear kernel; the flux TASK keeps every output
  // bit exact with them (the fused Cartesian stage does without: DESIGN.md section 4)
  constexpr bool GUARD = (CURV && RECON == 1) || FLUXES;
  // the Cartesian stage: detect-and-redo instead of a second code path (which costs registers it does not have)
#ifndef ARTEMIS_DETECT
#define ARTEMIS_DETECT 1
#endif
  constexpr bool DETECT = !CURV && !FLUXES && (ARTEMIS_DETECT != 0);
  // ... and only in stages whose input may hold such a velocity (wave-uniform; artemis_stage_args_t.tiny_in)
  bool detect = DETECT && a.redo_cnt0 != nullptr;
  if constexpr (DETECT) {
    if (detect && a.tiny_in) detect = (*a.tiny_in != 0u);
  }
  if constexpr (GUARD || DETECT) {
    if (x.t == 0) tiny_flag(S, 0) = tiny_flag(S, 1) = 0;
    __syncthreads(); // the flags are cleared before any wave sets one for the first staged plane (and the tables are in)
  }

  const double *u1_r = a.prim_u1[x.b * 6 + 0], *u1_1 = a.prim_u1[x.b * 6 + 1];
  const double *u1_2 = a.prim_u1[x.b * 6 + 2], *u1_3 = a.prim_u1[x.b * 6 + 3];
  const double *u1_e = a.prim_u1[x.b * 6 + 5];
  Raw5 u1raw;
  u1raw.d = u1raw.v1 = u1raw.v2 = u1raw.v3 = u1raw.e = 0.0;
  if constexpr (!D3) {
    const Cell6 qc = load_cell(x.in_r, x.in_1, x.in_2, x.in_3, x.in_e, x.col + k0 * x.sk, x.gm1);
    if constexpr (HAS_U1) u1raw = load_raw(u1_r, u1_1, u1_2, u1_3, u1_e, x.col + k0 * x.sk);
    Flux8 fz, fx_lo, fy_lo;
    fz.d = fz.m1 = fz.m2 = fz.m3 = fz.e = fz.eg = fz.pf = fz.vf = 0.0;
    Raw5 hal = u1raw;
    if (x.hr >= 0) hal = load_raw(x.in_r, x.in_1, x.in_2, x.in_3, x.in_e, x.hcol + k0 * x.sk);
    stage_plane(S, x, qc, hal);
    if constexpr (GUARD || DETECT) {
      if (GUARD || detect) stage_plane_flag(S, x, qc, hal, k0 & 1);
    }
    __syncthreads();
    bool flagged = false;
    plane_sweeps<RIEMANN, RECON, false, CURV, GUARD, DETECT>(S, P, x, gx, k0, qc, false, qc, hal, fx_lo, fy_lo, flagged, detect FPROF_PASS);
    if constexpr (CURV) {
      DFlux24 df{};
      if (gx.m3) gx.co.c3 = gx.m3[MT3_COS * (P.nk + 1) + k0], gx.co.s3 = gx.m3[MT3_SIN * (P.nk + 1) + k0];
      if (src.v.diff_on) df = src.v.dsum ? load_dsum(src.v.dsum, x.b, x.col + static_cast<unsigned>(k0) * x.sk)
                                         : load_dflux(P, x.b, static_cast<long>(x.col + static_cast<unsigned>(k0) * x.sk), x.multi_d, false);
      plane_update_curv<HAS_U1, WITH_DT, false>(S, P, a, src.v, x, gx, k0, qc, fx_lo, fy_lo, fz, fz, u1raw, df, ldt);
    }
    else if constexpr (FLUXES) plane_store_fluxes<false>(S, P, x, k0, fx_lo, fy_lo, fz, fz);
    else plane_update<HAS_U1, WRITE_CONS, WITH_DT, false>(S, P, a, x, k0, qc, fx_lo, fy_lo, fz, fz, u1raw, ldt,
                                                          flagged, shell_wg);
  } else {
    // x3 state carried in registers: planes k, k+1, the upper face value of cell k and the
    // flux through face k.
    // planes behind an outflow x3 face are the first / last active plane (StageK::outflow): wave-uniform index arithmetic
    auto kpl = [&](int kk) {
      if (outflow & 16) kk = max(kk, P.ks);
      if (outflow & 32) kk = min(kk, P.ke);
      return static_cast<unsigned>(kk);
    };
    Cell6 qc = load_cell(x.in_r, x.in_1, x.in_2, x.in_3, x.in_e, x.col + kpl(k0 - 1) * x.sk, x.gm1);
    Cell6 qn = load_cell(x.in_r, x.in_1, x.in_2, x.in_3, x.in_e, x.col + k0 * x.sk, x.gm1);
    Cell6 zl;
    // DETECT: the x3 sweep of zone k reads planes k-2 .. k+2 of its own column.  Planes k+1 and k+2 are in registers
    // when zone k is updated and are tested there; planes k-1 and k-2 are remembered as the (wave-uniform, hence
    // scalar: the kernel has no vector register to spare) plane flags of the two trips before, bit 0 = plane k-1 --
    // tile-wide and therefore conservative.  The chunk's first zones see planes below the chunk through `below`.
    [[maybe_unused]] unsigned fhist = 0;
    [[maybe_unused]] bool ahead1 = false; // the own column's plane k+1 holds a tiny velocity
    {
      const Cell6 qmm =
          load_cell(x.in_r, x.in_1, x.in_2, x.in_3, x.in_e, x.col + kpl(k0 - 2) * x.sk, x.gm1);
      if constexpr (DETECT) {
        if (detect) fhist = __any(tiny_v(qmm) || tiny_v(qc)) ? 3u : 0u, ahead1 = tiny_v(qn); // planes k0-2, k0-1 (not staged by this chunk); k0
      }
      if constexpr (CURV && RECON == 1) {
        const PlmGeo g3 = plm_geo_x3(gx.co, x.g, k0 - 1);
        double unused_;
#define ZL0(m) plm_g_shared<0>(qmm.m, qc.m, qn.m, zl.m, unused_, g3);
        FOR6(ZL0)
#undef ZL0
      } else {
        const bool f0 = !(FLUXES && __any(tiny_v(qmm) || tiny_v(qc) || tiny_v(qn)));
#define ZL0(m) zl.m = up_val<RECON>(qc.m, slope_sel<RECON>(qmm.m, qc.m, qn.m, f0));
        FOR6(ZL0)
#undef ZL0
      }
    }
    Flux8 fz_lo;
    fz_lo.d = fz_lo.m1 = fz_lo.m2 = fz_lo.m3 = fz_lo.e = fz_lo.eg = fz_lo.pf = fz_lo.vf = 0.0;
    Raw5 hal = u1raw; // halo cell of plane k+1 (staged by trip k)
    for (int k = k0 - 1; k <= k1; ++k) { // the first trip only primes fz_lo (face k0)
      // Issue this trip's HBM loads first; they are consumed after the plane's LDS phases, so
      // their latency hides behind the x1/x2 sweeps (barriers do not drain vmcnt).
      const Raw5 rnn = load_raw(x.in_r, x.in_1, x.in_2, x.in_3, x.in_e, x.col + kpl(k + 2) * x.sk);
      if constexpr (HAS_U1) {
        if (k >= k0) u1raw = load_raw(u1_r, u1_1, u1_2, u1_3, u1_e, x.col + static_cast<unsigned>(k) * x.sk);
      }
      // halo cell of plane k+1: staged at the end of this trip's P2, so its latency hides behind
      // the slopes and Riemann problems of plane k
      if (x.hr >= 0 && k < k1) hal = load_raw(x.in_r, x.in_1, x.in_2, x.in_3, x.in_e, x.hcol + (k + 1) * x.sk);
      DFlux24 df{};
      if constexpr (CURV) {
        if (src.v.diff_on && k >= k0)
          df = src.v.dsum ? load_dsum(src.v.dsum, x.b, x.col + static_cast<unsigned>(k) * x.sk)
                          : load_dflux(P, x.b, static_cast<long>(x.col + static_cast<unsigned>(k) * x.sk), true, true);
        // cos / sin of this plane's x3 centre (spherical3D, axisymmetric), also ahead of their use
        if (gx.m3 && k >= k0) gx.co.c3 = gx.m3[MT3_COS * (P.nk + 1) + k], gx.co.s3 = gx.m3[MT3_SIN * (P.nk + 1) + k];
      }
      Flux8 fx_lo, fy_lo;
      bool flagged = false;
      if (k >= k0) {
        plane_sweeps<RIEMANN, RECON, true, CURV, GUARD, DETECT>(S, P, x, gx, k, qc, k < k1, qn, hal, fx_lo, fy_lo, flagged, detect FPROF_PASS);
      } else { // priming trip: stage the first plane
        stage_plane(S, x, qn, hal);
        if constexpr (GUARD || DETECT) {
          if (GUARD || detect) stage_plane_flag(S, x, qn, hal, k0 & 1);
        }
        __syncthreads();
      }
      // x3 sweep, registers only: slope of cell k+1, face k+1
      const Cell6 qnn = finish_cell(rnn, x.gm1);
#ifdef FUSED_PROF
      if (qnn.d == -1.2345) prof_acc[8]++; // (the prefetch has arrived when the clock is read)
#endif
      FPROF(5);
      [[maybe_unused]] bool ahead = false;
      if constexpr (DETECT) { // planes k+1, k+2 of the own column (k+1 was tested as this trip's k+2 one trip ago)
        if (detect) {
          const bool a2 = tiny_v(qnn);
          ahead = ahead1 || a2;
          ahead1 = a2;
        }
      }
      // face k+1 reads the own column's planes k-1 .. k+2: the three in registers decide for the wave (the slope of
      // cell k entered zl in the previous trip under that trip's check)
      const bool fast_col = !(FLUXES && __any(tiny_v(qc) || tiny_v(qn) || tiny_v(qnn)));
      Cell6 zr, zl_next;
      if constexpr (CURV && RECON == 1) {
        const PlmGeo g3 = plm_geo_x3(gx.co, x.g, k + 1);
        // the own column's three cells decide for the wave whether the hand-scheduled divisions are safe
        const bool tiny3 = tiny_nonzero(qc.v1) || tiny_nonzero(qc.v2) || tiny_nonzero(qc.v3) || tiny_nonzero(qn.v1) ||
                           tiny_nonzero(qn.v2) || tiny_nonzero(qn.v3) || tiny_nonzero(qnn.v1) || tiny_nonzero(qnn.v2) ||
                           tiny_nonzero(qnn.v3);
        if (!__any(tiny3)) {
#define ZSL(m) plm_g_shared<2>(qc.m, qn.m, qnn.m, zl_next.m, zr.m, g3);
          FOR6(ZSL)
#undef ZSL
        } else {
#define ZSL(m) plm_g_shared<0>(qc.m, qn.m, qnn.m, zl_next.m, zr.m, g3);
          FOR6(ZSL)
#undef ZSL
        }
      } else {
#define ZSL(m)                                                                             \
  {                                                                                        \
    const double s_ = slope_sel<RECON>(qc.m, qn.m, qnn.m, fast_col);                       \
    zr.m = lo_val<RECON>(qn.m, s_);                                                        \
    zl_next.m = up_val<RECON>(qn.m, s_);                                                   \
  }
        FOR6(ZSL)
#undef ZSL
      }
      Flux8 fz_hi = solve_face<RIEMANN, 3>(x.gk, zl, zr, fast_col);
      if constexpr (CURV) fz_hi.m2 *= gx.h3[1], fz_hi.m3 *= gx.h3[2]; // ScaleMomentumFlux at the x3 face
      FPROF(6);
      if (k >= k0) {
        if constexpr (CURV) plane_update_curv<HAS_U1, WITH_DT, true>(S, P, a, src.v, x, gx, k, qc, fx_lo, fy_lo, fz_lo, fz_hi, u1raw, df, ldt);
        else if constexpr (FLUXES) plane_store_fluxes<true>(S, P, x, k, fx_lo, fy_lo, fz_lo, fz_hi);
        else plane_update<HAS_U1, WRITE_CONS, WITH_DT, true>(S, P, a, x, k, qc, fx_lo, fy_lo, fz_lo, fz_hi, u1raw, ldt,
                                                             flagged || (fhist & 3u) != 0u || ahead, shell_wg);
      }
      FPROF(7);
      fz_lo = fz_hi, zl = zl_next, qc = qn, qn = qnn;
      if constexpr (DETECT) {
        if (k >= k0) fhist = (fhist << 1) | (flagged ? 1u : 0u);
      }
    }
  }

#ifdef FUSED_PROF
  FPROF(8);
  if ((x.t & 63) == 0)
    for (int q = 0; q < 10; ++q) atomicAdd(&g_fused_prof[q], prof_acc[q]);
#endif
  if constexpr (WITH_DT) {
    __syncthreads(); // LOX is reused as reduction scratch
    for (int off = 32; off > 0; off >>= 1) ldt = fmin(ldt, __shfl_down(ldt, off, 64));
    double *wmin = &S.LOX[0][0];
    if ((x.t & 63) == 0) wmin[x.t >> 6] = ldt;
    __syncthreads();
    if (x.t == 0) {
      double m = wmin[0];
      for (int w = 1; w < NW; ++w) m = fmin(m, wmin[w]);
      if (m < DBL_MAX)
        atomicMin(a.dt_bits, static_cast<unsigned long long>(__double_as_longlong(a.cfl * m)));
    }
  }
  if (shell_wg) {
    // MI355X: per-CU L1 and per-XCD L2 are not coherent across CUs/XCDs.  Every wave drains its
    // stores, the workgroup meets, then ONE lane releases at agent scope (writes back this
    // XCD's dirty L2 lines) and bumps the counter the comm stream's wait kernel polls.
    // (Write-through sc1 stores instead of the fence measured 2 % slower: 8-byte sc1 stores.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (x.t == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(a.shell_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Comm-stream side of the shell-first hand-off: one wave polls the counter (relaxed, agent
// scope, with s_sleep) until `target` workgroups have published, then acquires.  Kernels queued
// behind it on the same stream (halo packs) start with clean caches and see the shell's stores.
__global__ void wait_counter_kernel(unsigned *counter, unsigned target, unsigned *timeout_flag,
                                    unsigned long long spin_limit) {
  if (threadIdx.x != 0) return;
  unsigned long long spins = 0;
  while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(32);
    if (++spins > spin_limit) { // ~seconds: never hang the GPU on a logic error
      if (timeout_flag) *timeout_flag = 1u;
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// ---- detect-and-redo: the exact path of the zones the Cartesian stage kernel deferred ------------------------------
// One thread per listed zone: the whole stage of that zone -- 2 ndim faces from register stencils, ApplyUpdate,
// FluxSource, SetAuxillaryFields, ConsToPrim, PrimToCons, the zone's timestep -- with IEEE `/` and sqrt and the
// expression trees of the per-task kernels (device_math.hpp plm_dqm / hllc_gas / riemann_gas, kernels_stage_cell.hip's
// gas branch): the bits of the per-task chain whatever the magnitudes.  Reads prim_in (stencil) and
// prim_u1 (the zone itself, not yet overwritten: the stage kernel skipped its stores) and writes prim_out / cons_out.
// Normally the list is empty (the kernel reads the count and returns); velocities of 1e-61 and below next to a shock
// precursor are what fills it.
struct RedoK {
  double gam0, gam1, beta_dt, bdt, cfl;
  const double *bdt_ptr;
  double *const *prim_in, *const *prim_u1, *const *prim_out, *const *cons_out;
  unsigned long long *dt_bits;
  unsigned *tiny_out, *tiny_clear;
  int outflow;
  unsigned long long outflow_blk;
  int outflow_pb;
  unsigned *cnt;   // entries in the list; reset to zero by the last workgroup of this kernel (no memset between stages)
  unsigned *done;  // its ticket counter
  unsigned cap;
  const unsigned long long *list;
  int has_u1;
};
template <int RIEMANN, int RECON>
__global__ __launch_bounds__(256) void stage_redo_kernel(const PackView P, const RedoK a) {
  const unsigned n = min(__hip_atomic_load(a.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a.cap);
  const GasK gk = gas_constants(P.gm1);
  const FluidView &f = P.gas;
  const unsigned long long N = static_cast<unsigned long long>(P.nk) * P.nj * P.ni;
  double beta_dt = a.beta_dt, bdt = a.bdt;
  if (a.bdt_ptr) beta_dt = bdt = *a.bdt_ptr;
  const bool multi_d = P.ndim > 1, three_d = P.ndim > 2;
  for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
    const unsigned long long id = a.list[q];
    const int b = static_cast<int>(id / N);
    const long c = static_cast<long>(id % N);
    const int k = static_cast<int>(c / P.sk), j = static_cast<int>((c % P.sk) / P.sj), i = static_cast<int>(c % P.sj);
    const double *qr = a.prim_in[b * 6 + 0], *q1 = a.prim_in[b * 6 + 1], *q2 = a.prim_in[b * 6 + 2];
    const double *q3 = a.prim_in[b * 6 + 3], *qe = a.prim_in[b * 6 + 5];
    const double *g = P.geom + 6 * b;
    const double dx1 = (g[0] + (i + 1) * g[1]) - (g[0] + i * g[1]);
    const double dx2 = (g[2] + (j + 1) * g[3]) - (g[2] + j * g[3]);
    const double dx3 = (g[4] + (k + 1) * g[5]) - (g[4] + k * g[5]);
    const double ax1 = dx2 * dx3, ax2 = dx1 * dx3, ax3 = dx1 * dx2, vol = dx1 * dx2 * dx3; // geometry.hpp:199-225
    Flux8 lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) lo[d].d = lo[d].m1 = lo[d].m2 = lo[d].m3 = lo[d].e = lo[d].eg = lo[d].pf = lo[d].vf = 0.0, hi[d] = lo[d];
    auto faces = [&](auto DIRTAG, long st, Flux8 &flo, Flux8 &fhi) {
      constexpr int DIR = decltype(DIRTAG)::value;
      Cell6 w[5];
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        long cm = c + (m - 2) * st;
        const int outflow = outflow_of(a.outflow, a.outflow_blk, a.outflow_pb, b);
        if (outflow) { // the stage kernel's rule: the ghost zones behind an outflow face are the edge zone (not read from memory)
          const int at = (DIR == 1) ? i : ((DIR == 2) ? j : k), lo = (DIR == 1) ? P.is : ((DIR == 2) ? P.js : P.ks);
          const int hi = (DIR == 1) ? P.ie : ((DIR == 2) ? P.je : P.ke);
          int am = at + (m - 2);
          if (((outflow >> (2 * (DIR - 1))) & 1) && am < lo) am = lo;
          if (((outflow >> (2 * (DIR - 1) + 1)) & 1) && am > hi) am = hi;
          cm = c + (am - at) * st;
        }
        w[m] = load_cell(qr, q1, q2, q3, qe, cm, P.gm1);
      }
      Cell6 Ll, Rl, Lu, Ru;
#define RD(v)                                                                                          \
  {                                                                                                    \
    double sm = 0.0, sc = 0.0, sp = 0.0;                                                               \
    if constexpr (RECON == 1) sm = plm_dqm(w[0].v, w[1].v, w[2].v), sc = plm_dqm(w[1].v, w[2].v, w[3].v), sp = plm_dqm(w[2].v, w[3].v, w[4].v); \
    Ll.v = up_val<RECON>(w[1].v, sm), Rl.v = lo_val<RECON>(w[2].v, sc);                                \
    Lu.v = up_val<RECON>(w[2].v, sc), Ru.v = lo_val<RECON>(w[3].v, sp);                                \
  }
      FOR6(RD)
#undef RD
      flo = solve_face<RIEMANN, DIR>(gk, Ll, Rl, false);
      fhi = solve_face<RIEMANN, DIR>(gk, Lu, Ru, false);
    };
    faces(std::integral_constant<int, 1>{}, 1, lo[0], hi[0]);
    if (multi_d) faces(std::integral_constant<int, 2>{}, P.sj, lo[1], hi[1]);
    if (three_d) faces(std::integral_constant<int, 3>{}, P.sk, lo[2], hi[2]);
    // u0 = PrimToCons(prim_in), u1 = PrimToCons(prim_u1)  (fill_derived.cpp:226-255)
    const Cell6 qc = load_cell(qr, q1, q2, q3, qe, c, P.gm1);
    const double D0 = qc.d, M10 = qc.d * qc.v1 * 1.0, M20 = qc.d * qc.v2 * 1.0, M30 = qc.d * qc.v3 * 1.0;
    const double G0 = qc.e * qc.d;
    const double E0 = G0 + 0.5 * qc.d * (sqr(qc.v1) + sqr(qc.v2) + sqr(qc.v3));
    double D1 = D0, M11 = M10, M21 = M20, M31 = M30, G1 = G0, E1 = E0;
    if (a.has_u1) {
      const double r1 = a.prim_u1[b * 6 + 0][c], a1 = a.prim_u1[b * 6 + 1][c], a2 = a.prim_u1[b * 6 + 2][c];
      const double a3 = a.prim_u1[b * 6 + 3][c], e1 = a.prim_u1[b * 6 + 5][c];
      D1 = r1, M11 = r1 * a1 * 1.0, M21 = r1 * a2 * 1.0, M31 = r1 * a3 * 1.0;
      G1 = e1 * r1;
      E1 = G1 + 0.5 * r1 * (sqr(a1) + sqr(a2) + sqr(a3));
    }
    auto upd = [&](double u0, double u1, double f1l, double f1h, double f2l, double f2h, double f3l, double f3h) {
      double divf = (ax1 * f1l - ax1 * f1h); // artemis_integrator.hpp:88-106
      if (multi_d) divf += (ax2 * f2l - ax2 * f2h);
      if (three_d) divf += (ax3 * f3l - ax3 * f3h);
      return a.gam0 * u0 + a.gam1 * u1 + divf * beta_dt / vol;
    };
    const double D = upd(D0, D1, lo[0].d, hi[0].d, lo[1].d, hi[1].d, lo[2].d, hi[2].d);
    double M1 = upd(M10, M11, lo[0].m1, hi[0].m1, lo[1].m1, hi[1].m1, lo[2].m1, hi[2].m1);
    double M2 = upd(M20, M21, lo[0].m2, hi[0].m2, lo[1].m2, hi[1].m2, lo[2].m2, hi[2].m2);
    double M3 = upd(M30, M31, lo[0].m3, hi[0].m3, lo[1].m3, hi[1].m3, lo[2].m3, hi[2].m3);
    const double E = upd(E0, E1, lo[0].e, hi[0].e, lo[1].e, hi[1].e, lo[2].e, hi[2].e);
    double G = upd(G0, G1, lo[0].eg, hi[0].eg, lo[1].eg, hi[1].eg, lo[2].eg, hi[2].eg);
    // FluxSource (fluid_fluxes.hpp:365-392)
    M1 += bdt / dx1 * (lo[0].pf - hi[0].pf);
    G -= bdt / vol * 0.5 * (lo[0].pf + hi[0].pf) * (ax1 * hi[0].vf - ax1 * lo[0].vf);
    if (multi_d) {
      M2 += bdt / dx2 * (lo[1].pf - hi[1].pf);
      G -= bdt / vol * 0.5 * (lo[1].pf + hi[1].pf) * (ax2 * hi[1].vf - ax2 * lo[1].vf);
    }
    if (three_d) {
      M3 += bdt / dx3 * (lo[2].pf - hi[2].pf);
      G -= bdt / vol * 0.5 * (lo[2].pf + hi[2].pf) * (ax3 * hi[2].vf - ax3 * lo[2].vf);
    }
    // SetAuxillaryFields (fill_derived.cpp:54-73, artemis_utils.hpp:43-62) + ConsToPrim (:137-151)
    const double w_d = (D > f.dfloor) ? D : f.dfloor;
    {
      const double ke = 0.5 * (sqr(M1 / 1.0) + sqr(M2 / 1.0) + sqr(M3 / 1.0)) / w_d;
      const double ue = E - ke;
      double sie = ((ue > f.de_switch * E) ? ue : G) / w_d;
      sie = amax(sie, f.siefloor);
      G = sie * w_d;
      const double uflr = f.siefloor * w_d;
      G = (G > uflr) ? G : uflr;
    }
    const double w1 = M1 / w_d, w2 = M2 / w_d, w3 = M3 / w_d;
    double w_s = G / w_d;
    w_s = (w_s > f.siefloor) ? w_s : f.siefloor;
    const double w_p = amax(0.0, P.gm1 * w_d * w_s); // fill_derived.cpp:247
    if (a.tiny_out && tiny_vel3(w1, w2, w3)) __hip_atomic_fetch_or(a.tiny_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a.prim_out[b * 6 + 0][c] = w_d, a.prim_out[b * 6 + 1][c] = w1, a.prim_out[b * 6 + 2][c] = w2;
    a.prim_out[b * 6 + 3][c] = w3, a.prim_out[b * 6 + 4][c] = w_p, a.prim_out[b * 6 + 5][c] = w_s;
    if (a.cons_out) { // PrimToCons (fill_derived.cpp:226-255)
      const double u_u = w_s * w_d;
      a.cons_out[b * 6 + 0][c] = w_d, a.cons_out[b * 6 + 1][c] = w_d * w1 * 1.0, a.cons_out[b * 6 + 2][c] = w_d * w2 * 1.0;
      a.cons_out[b * 6 + 3][c] = w_d * w3 * 1.0;
      a.cons_out[b * 6 + 4][c] = u_u + 0.5 * w_d * (sqr(w1) + sqr(w2) + sqr(w3));
      a.cons_out[b * 6 + 5][c] = u_u;
    }
    if (a.dt_bits) { // Gas::EstimateTimestepMesh on the new state (gas.cpp:411-433)
      const double bulk = (P.gm1 + 1.0) * P.gm1 * w_d * w_s;
      const double cs = sqrt(bulk / w_d);
      double denom = 0.0;
      denom += (fabs(w1) + cs) / dx1;
      if (multi_d) denom += (fabs(w2) + cs) / dx2;
      if (three_d) denom += (fabs(w3) + cs) / dx3;
      atomicMin(a.dt_bits, static_cast<unsigned long long>(__double_as_longlong(a.cfl * (1.0 / denom))));
    }
  }
  // every workgroup has read the count by now; the last one to get here empties the list for the next stage
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == gridDim.x - 1) {
      __hip_atomic_store(a.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.tiny_clear) __hip_atomic_store(a.tiny_clear, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.done, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Planes per chunk of the x3 march.  The launch proceeds in rounds of `slots` resident workgroups (256 CUs x 2, or
// x 1 for the curvilinear instantiations) and a half-empty last round costs a full one; every chunk also pays one
// priming trip.  Among 8..16 planes take the best product of round fill and march efficiency (256^3: 16 planes,
// 4096 workgroups = 8 full rounds; 192 x 128^2: 8 planes, 3 full rounds instead of 1.5).
int pick_chunk(int planes, long tiles, long slots, int cmax = 16) {
  if (opt(OPT_FUSED_KCHUNK) > 0) return static_cast<int>(opt(OPT_FUSED_KCHUNK)); // tuning knob
  int best = 16;
  double score = -1.0;
  // (<= 16 where the launch has a boundary shell, which is one chunk thick; launches without one take up to 128 planes
  // on the Cartesian path and 64 on the curvilinear one -- 256 x 128^2: two chunks, one full round of 256 workgroups)
  for (int c = cmax; c >= 8; --c) {
    const int n0 = std::max(1, planes / c);
    const int kc = (planes + n0 - 1) / n0; // what add_box makes of it
    const int nchunk = (planes + kc - 1) / kc;
    const long wgs = tiles * nchunk;
    const double fill = static_cast<double>(wgs) / static_cast<double>(((wgs + slots - 1) / slots) * slots);
    const double sc = fill * kc / (kc + 1.0);
    if (sc > score + 1e-9) score = sc, best = c;
  }
  return best;
}

template <int RIEMANN, int RECON>
int launch_cfg(const PackView &P, const StageK &k, bool has_u1, bool cons, bool dt, hipStream_t s) {
  const dim3 grid(k.start[k.nbox]);
  const dim3 block(FTX, FTY);
#define GO(U, C, D)                                                                        \
  do {                                                                                     \
    if (P.ndim > 2) hipLaunchKernelGGL((stage_fused_kernel<RIEMANN, RECON, U, C, D, true>), grid, block, 0, s, P, k, SrcArg<false>{}); \
    else hipLaunchKernelGGL((stage_fused_kernel<RIEMANN, RECON, U, C, D, false>), grid, block, 0, s, P, k, SrcArg<false>{}); \
  } while (0)
  if (has_u1) {
    if (cons) { if (dt) GO(true, true, true); else GO(true, true, false); }
    else { if (dt) GO(true, false, true); else GO(true, false, false); }
  } else {
    if (cons) { if (dt) GO(false, true, true); else GO(false, true, false); }
    else { if (dt) GO(false, false, true); else GO(false, false, false); }
  }
#undef GO
  return 0;
}

} // namespace

struct Betas {
  double b[3];
  int n;
};
__global__ void advance_dt_kernel(double *st, double tlim, const Betas be) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double time = st[0], dt = st[1];
  const double est = st[2];
  time += dt;
  double ndt = dt;
  if (ndt < 0.1 * DBL_MAX) ndt *= 2.0;
  ndt = (est < ndt) ? est : ndt; // std::min(ndt, est)
  if (tlim > 0.0 && time < tlim && (tlim - time) < ndt) ndt = tlim - time;
  st[0] = time, st[1] = ndt, st[2] = DBL_MAX;
  for (int q = 0; q < be.n; ++q) st[3 + q] = be.b[q] * ndt;
}
void launch_advance_dt(double *state, double tlim, int nstages, const double *beta, hipStream_t s) {
  Betas be;
  be.n = nstages;
  for (int q = 0; q < 3; ++q) be.b[q] = (q < nstages) ? beta[q] : 0.0;
  hipLaunchKernelGGL(advance_dt_kernel, dim3(1), dim3(64), 0, s, state, tlim, be);
}

void launch_wait_counter(unsigned *counter, unsigned target, unsigned *timeout_flag, hipStream_t s) {
  unsigned long long limit = 1ull << 26; // ARTEMIS_WAIT_SPIN_LIMIT: diagnostics / the timeout test
  if (opt(OPT_WAIT_SPIN_LIMIT) > 0) limit = opt(OPT_WAIT_SPIN_LIMIT);
  hipLaunchKernelGGL(wait_counter_kernel, dim3(1), dim3(64), 0, s, counter, target, timeout_flag, limit);
}

// Redo lists of the calling thread: [0] bulk, [1] shell; each a counter and room for every zone of the largest pack
// seen.  Allocated on first use (outside any stream capture: the host driver's first plain steps), grown on demand.
namespace {
struct RedoBufs {
  unsigned *cnt = nullptr; // two entry counters + two ticket counters (zero between launches)
  unsigned long long *list[2] = {nullptr, nullptr};
  size_t cap = 0;
};
thread_local RedoBufs g_redo;
bool redo_enabled() { return opt(OPT_NO_REDO) == 0; }
bool ensure_redo(size_t zones) {
  if (g_redo.cnt && g_redo.cap >= zones) return true;
  (void)hipDeviceSynchronize();
  if (g_redo.list[0]) (void)hipFree(g_redo.list[0]);
  if (g_redo.list[1]) (void)hipFree(g_redo.list[1]);
  g_redo.list[0] = g_redo.list[1] = nullptr, g_redo.cap = 0;
  if (!g_redo.cnt) {
    if (hipMalloc(reinterpret_cast<void **>(&g_redo.cnt), 4 * sizeof(unsigned)) != hipSuccess) return false;
    if (hipMemset(g_redo.cnt, 0, 4 * sizeof(unsigned)) != hipSuccess) return false;
  }
  for (int q = 0; q < 2; ++q)
    if (hipMalloc(reinterpret_cast<void **>(&g_redo.list[q]), zones * sizeof(unsigned long long)) != hipSuccess) return false;
  g_redo.cap = zones;
  return true;
}
size_t pack_zones(const PackView &P) {
  return static_cast<size_t>(P.nb) * (P.ie - P.is + 1) * (P.je - P.js + 1) * (P.ke - P.ks + 1);
}
// the lists of this call: the caller's scratch ([16 unsigned | list 0 | list 1], artemis_hip_redo_scratch_bytes) or
// the calling thread's own
RedoBufs redo_bufs_of(const PackView &P, const artemis_stage_args_t &a) {
  if (!a.redo_scratch) return g_redo;
  RedoBufs B;
  const size_t zones = pack_zones(P);
  B.cnt = static_cast<unsigned *>(a.redo_scratch);
  B.list[0] = reinterpret_cast<unsigned long long *>(static_cast<char *>(a.redo_scratch) + 64);
  B.list[1] = B.list[0] + zones;
  B.cap = zones;
  return B;
}
template <int RIEMANN, int RECON>
void launch_redo_cfg(const PackView &P, const RedoK &r, hipStream_t s) {
  // (an empty pass costs by its grid: 5.5 us at 128 workgroups, every stage; the lists are short when they are not empty)
  hipLaunchKernelGGL((stage_redo_kernel<RIEMANN, RECON>), dim3(32), dim3(256), 0, s, P, r);
}
// artemis_stage_args_t.outflow_faces_by_block (a HOST array) packed into six bits per block; 0: the pack-wide mask applies
static int pack_outflow(const PackView &P, const artemis_stage_args_t &a, unsigned long long &blk) {
  blk = 0;
  if (!a.outflow_faces_by_block) return 0;
  for (int b = 0; b < P.nb && b < 10; ++b) blk |= static_cast<unsigned long long>(a.outflow_faces_by_block[b] & 63) << (6 * b);
  return 1;
}
void launch_redo(const PackView &P, const artemis_stage_args_t &a, int riemann, int recon, int which, hipStream_t s) {
  RedoK r;
  r.gam0 = a.gam0, r.gam1 = a.gam1, r.beta_dt = a.beta_dt, r.bdt = a.bdt, r.cfl = a.cfl;
  r.bdt_ptr = a.beta_dt_dev;
  r.prim_in = a.prim_in, r.prim_u1 = a.prim_u1, r.prim_out = a.prim_out, r.cons_out = a.cons_out;
  r.dt_bits = reinterpret_cast<unsigned long long *>(a.dt_dev);
  r.tiny_out = a.tiny_out, r.tiny_clear = (which == 0) ? a.tiny_clear : nullptr;
  r.outflow = a.outflow_faces & 63;
  r.outflow_pb = pack_outflow(P, a, r.outflow_blk);
  const RedoBufs B = redo_bufs_of(P, a);
  r.cnt = B.cnt + which, r.done = B.cnt + 2 + which, r.cap = static_cast<unsigned>(std::min<size_t>(B.cap, 0xffffffffu));
  r.list = B.list[which];
  r.has_u1 = (a.prim_u1 != a.prim_in) ? 1 : 0;
#define RC(RS)                                                                             \
  case RS:                                                                                 \
    if (recon == ARTEMIS_PCM) launch_redo_cfg<RS, 0>(P, r, s);                             \
    else launch_redo_cfg<RS, 1>(P, r, s);                                                  \
    break;
  switch (riemann) {
    RC(0)
    RC(1)
    RC(2)
  }
#undef RC
}
} // namespace
// The shell list of the calling thread's last shell-first launch (artemis_stage_args_t.shell_done), on the stream
// that has waited for the shell counter.
int launch_stage_fused_redo_shell(const PackView &P, const artemis_stage_args_t &a, int riemann, int recon, hipStream_t s) {
  if (!redo_enabled() || (!a.redo_scratch && !g_redo.cnt)) return 0;
  launch_redo(P, a, riemann, recon, 1, s);
  return 0;
}

// the tile march addresses cells with 32-bit byte offsets (fused_device.hpp gld / gst)
static bool offsets_fit(const PackView &P) { return static_cast<long>(P.nk) * P.nj * P.ni < (1L << 29); }
int launch_stage_fused(const PackView &P, const artemis_stage_args_t &a, int riemann, int recon,
                       hipStream_t s) {
  if (recon == ARTEMIS_PPM) return 3; // PPM stays on the per-task path
  if (!offsets_fit(P)) return 6;
  StageK k;
  k.gam0 = a.gam0, k.gam1 = a.gam1, k.beta_dt = a.beta_dt, k.bdt = a.bdt, k.cfl = a.cfl;
  k.bdt_ptr = a.beta_dt_dev;
  k.prim_in = a.prim_in, k.prim_u1 = a.prim_u1, k.prim_out = a.prim_out, k.cons_out = a.cons_out;
  k.dt_bits = reinterpret_cast<unsigned long long *>(a.dt_dev);
  const int nz = P.ke - P.ks + 1;
  const int NTI = (P.ie - P.is + FTX) / FTX, NTJ = (P.je - P.js + FTY) / FTY;
  // A launch without a boundary shell (no overlapped exchange: one rank, or links handled after the kernel) takes long
  // chunks -- up to 128 planes: at 256^3 two chunks per tile column, 512 workgroups, exactly one round of the chip's
  // slots, and the priming trip once per 128 planes instead of once per 16 (0.941 -> 0.916 ms).  Launches with a shell
  // keep <= 16 for the shell boxes (the x3 shell is one chunk thick) and give the bulk box long chunks of its own.
  const bool has_shell = a.shell_done != nullptr || a.region != 0;
  const int target_chunk = pick_chunk(nz, static_cast<long>(NTI) * NTJ * P.nb, 512, has_shell ? 16 : 128);
  k.nbox = 0;
  k.start[0] = 0;
  int box_chunk = target_chunk; // (the bulk box of a launch with a shell takes long chunks of its own, below)
  auto add_box = [&](int ti0, int nti, int tj0, int ntj, int kb0, int kb1) {
    if (nti <= 0 || ntj <= 0 || kb1 < kb0) return;
    const int q = k.nbox++;
    k.ti0[q] = ti0, k.nti[q] = nti, k.tj0[q] = tj0, k.ntj[q] = ntj, k.kb0[q] = kb0, k.kb1[q] = kb1;
    const int planes = kb1 - kb0 + 1;
    k.nchunk[q] = (P.ndim > 2) ? std::max(1, planes / box_chunk) : 1;
    k.kchunk[q] = (planes + k.nchunk[q] - 1) / k.nchunk[q];
    k.nchunk[q] = (planes + k.kchunk[q] - 1) / k.kchunk[q];
    k.start[q + 1] = k.start[q] + nti * ntj * k.nchunk[q] * P.nb;
  };
  // region 0: whole block.  region 1: boundary shell = everything a neighbour's ghost slab is cut
  // from, rounded out to whole tiles (x1, x2) / nghost planes (x3).  region 2: the complement.
  const int g = P.ng;
  // Which faces' boundary cells form the shell (bit f of shell_faces; 0 = all six).  x3 part:
  // `zt` planes at a flagged end.  Separate launches (regions 1/2) keep it as thin as the ghost
  // depth; the single-launch shell-first mode uses whole chunks, which cost nothing extra there
  // and avoid 2-plane workgroups that pay the priming trip for nothing.
  const int faces = (a.shell_faces & 63) ? (a.shell_faces & 63) : 63;
  const int xlo = (faces >> 0) & 1, xhi = (faces >> 1) & 1;
  const int ylo = (P.ndim > 1) ? (faces >> 2) & 1 : 0, yhi = (P.ndim > 1) ? (faces >> 3) & 1 : 0;
  const int zlo = (P.ndim > 2) ? (faces >> 4) & 1 : 0, zhi = (P.ndim > 2) ? (faces >> 5) & 1 : 0;
  const int zt = a.shell_done ? std::max(g, std::min(target_chunk, nz / 3)) : g;
  // the shell/bulk split exists only if every flagged side can be cut off and something is
  // left; otherwise the whole block is "shell" and the bulk is empty (region 1 must always
  // contain every flagged boundary cell)
  const bool split = (NTI > xlo + xhi) && (NTJ > ylo + yhi) && (nz > (zlo + zhi) * zt);
  const int km0 = P.ks + zlo * zt, km1 = P.ke - zhi * zt;  // middle k-range
  const int tjm0 = ylo, ntjm = NTJ - ylo - yhi;            // middle j-tiles
  const int tim0 = xlo, ntim = NTI - xlo - xhi;            // middle i-tiles
  auto add_shell = [&]() {
    if (zlo) add_box(0, NTI, 0, NTJ, P.ks, P.ks + zt - 1);
    if (zhi) add_box(0, NTI, 0, NTJ, P.ke - zt + 1, P.ke);
    if (ylo) add_box(0, NTI, 0, 1, km0, km1);
    if (yhi) add_box(0, NTI, NTJ - 1, 1, km0, km1);
    if (xlo) add_box(0, 1, tjm0, ntjm, km0, km1);
    if (xhi) add_box(NTI - 1, 1, tjm0, ntjm, km0, km1);
  };
  k.nshell = 0, k.shell_done = nullptr;
  if (a.shell_done) {
    // whole block in ONE launch, boundary shell first: shell workgroups get the lowest ids and
    // count themselves into *shell_done (target returned through *shell_target)
    if (split) {
      add_shell();
      k.nshell = k.start[k.nbox];
      box_chunk = pick_chunk(km1 - km0 + 1, static_cast<long>(ntim) * ntjm * P.nb, 512, 128); // the bulk: long chunks
      add_box(tim0, ntim, tjm0, ntjm, km0, km1);
    } else {
      add_box(0, NTI, 0, NTJ, P.ks, P.ke);
      k.nshell = k.start[k.nbox];
    }
    k.shell_done = a.shell_done;
    if (a.shell_target) *a.shell_target = static_cast<unsigned>(k.nshell);
  } else if (a.region == 0 || (a.region == 1 && !split)) {
    add_box(0, NTI, 0, NTJ, P.ks, P.ke);
  } else if (a.region == 1) {
    add_shell();
  } else if (split) {
    box_chunk = pick_chunk(km1 - km0 + 1, static_cast<long>(ntim) * ntjm * P.nb, 512, 128); // the bulk: long chunks
    add_box(tim0, ntim, tjm0, ntjm, km0, km1);
  }
  if (k.nbox == 0) return 0; // nothing to do in this region
  // detect-and-redo: zones next to vanishing velocities are deferred to the exact kernel (below)
  k.redo_cnt0 = k.redo_cnt1 = nullptr, k.redo_list0 = k.redo_list1 = nullptr, k.redo_cap = 0;
  k.tiny_in = a.tiny_in, k.tiny_out = a.tiny_out;
  k.outflow = a.outflow_faces & 63;
  k.outflow_pb = pack_outflow(P, a, k.outflow_blk);
  const bool redo = redo_enabled();
  if (redo) {
    if (!a.redo_scratch && !ensure_redo(pack_zones(P))) return 5;
    const RedoBufs B = redo_bufs_of(P, a);
    k.redo_cnt0 = B.cnt, k.redo_cnt1 = B.cnt + 1;
    k.redo_list0 = B.list[0], k.redo_list1 = B.list[1];
    k.redo_cap = static_cast<unsigned>(std::min<size_t>(B.cap, 0xffffffffu));
  }
  // XCD-aware id remap of the bulk workgroups: no effect on the run time (the kernel is VALU-bound) but
  // it removes the halo / straddled-line re-reads between XCDs from the HBM traffic (profiles/r02*pmc*)
  k.xcd_swizzle = opt(OPT_FUSED_NO_SWIZZLE) ? 0 : 1;
  const bool has_u1 = (a.prim_u1 != a.prim_in);
  const bool cons = (a.cons_out != nullptr);
  const bool dt = (a.dt_dev != nullptr);
  int rc = 4;
#define RC(RS)                                                                             \
  case RS:                                                                                 \
    rc = (recon == ARTEMIS_PCM) ? launch_cfg<RS, 0>(P, k, has_u1, cons, dt, s)             \
                                : launch_cfg<RS, 1>(P, k, has_u1, cons, dt, s);            \
    break;
  switch (riemann) {
    RC(0)
    RC(1)
    RC(2)
  }
#undef RC
  // the bulk list on the launch's own stream; the shell list of a shell-first launch belongs to the stream that waits
  // for the shell (launch_stage_fused_redo_shell)
  if (rc == 0 && redo) launch_redo(P, a, riemann, recon, 0, s);
  return rc;
}

// ---- Gas::CalculateFluxes through the tile march -------------------------------------------------
// The per-task flux kernel (kernels_unfused.hip) reads every stencil value from L1 / L2 and solves each face
// from scratch in a one-thread-per-zone launch; this is the same LDS-staged march as the fused stage with the
// update phase replaced by the stores of the task, for the decks the Cartesian march covers: one gas species,
// PCM / PLM, blocks at least a tile wide.  Same device functions as the fused stage, which is bit-identical to the
// per-task chain, so the task's outputs do not change.
bool fused_flux_covers(const PackView &P, int recon) {
  if (opt(OPT_NO_TILED_FLUX)) return false;
  if (static_cast<long>(P.nk) * P.nj * P.ni >= (1L << 29)) return false;
  if (P.coords != ARTEMIS_CARTESIAN || P.gas.ns != 1 || P.ng < 2 || P.ndim < 2) return false;
  if (recon != ARTEMIS_PCM && recon != ARTEMIS_PLM) return false;
  return (P.ie - P.is + 1) >= FTX && (P.je - P.js + 1) >= FTY;
}
namespace {
template <int RIEMANN, int RECON>
void launch_flux_cfg(const PackView &P, const StageK &k, hipStream_t s) {
  const dim3 grid(k.start[k.nbox]);
  const dim3 block(FTX, FTY);
  if (P.ndim > 2)
    hipLaunchKernelGGL((stage_fused_kernel<RIEMANN, RECON, false, false, false, true, false, true>), grid, block, 0, s, P, k, SrcArg<false>{});
  else
    hipLaunchKernelGGL((stage_fused_kernel<RIEMANN, RECON, false, false, false, false, false, true>), grid, block, 0, s, P, k, SrcArg<false>{});
}
} // namespace
int launch_flux_fused(const PackView &P, int riemann, int recon, hipStream_t s) {
  StageK k;
  k.gam0 = k.gam1 = k.beta_dt = k.bdt = k.cfl = 0.0;
  k.bdt_ptr = nullptr;
  k.prim_in = P.gas.prim, k.prim_u1 = P.gas.prim, k.prim_out = nullptr, k.cons_out = nullptr;
  k.dt_bits = nullptr;
  const int nz = P.ke - P.ks + 1;
  const int NTI = (P.ie - P.is + FTX) / FTX, NTJ = (P.je - P.js + FTY) / FTY;
  const int target_chunk = pick_chunk(nz, static_cast<long>(NTI) * NTJ * P.nb, 512, 128); // (no shell in the flux task)
  k.nbox = 1, k.start[0] = 0;
  k.ti0[0] = 0, k.nti[0] = NTI, k.tj0[0] = 0, k.ntj[0] = NTJ, k.kb0[0] = P.ks, k.kb1[0] = P.ke;
  k.nchunk[0] = (P.ndim > 2) ? std::max(1, nz / target_chunk) : 1;
  k.kchunk[0] = (nz + k.nchunk[0] - 1) / k.nchunk[0];
  k.nchunk[0] = (nz + k.kchunk[0] - 1) / k.kchunk[0];
  k.start[1] = NTI * NTJ * k.nchunk[0] * P.nb;
  k.nshell = 0, k.shell_done = nullptr;
  k.redo_cnt0 = k.redo_cnt1 = nullptr, k.redo_list0 = k.redo_list1 = nullptr, k.redo_cap = 0;
  k.tiny_in = nullptr, k.tiny_out = nullptr;
  k.outflow = 0, k.outflow_blk = 0, k.outflow_pb = 0;
  k.xcd_swizzle = opt(OPT_FUSED_NO_SWIZZLE) ? 0 : 1;
#define RC(RS)                                                                             \
  case RS:                                                                                 \
    if (recon == ARTEMIS_PCM) launch_flux_cfg<RS, 0>(P, k, s);                             \
    else launch_flux_cfg<RS, 1>(P, k, s);                                                  \
    return 0;
  switch (riemann) {
    RC(0)
    RC(1)
    RC(2)
  }
#undef RC
  return 4;
}

// ---- curvilinear decks through artemis_hip_stage_general -----------------------------------------
// Gas (one species) on any non-Cartesian system, PCM / PLM, with the pointwise tasks plane_update_curv folds
// in; everything else stays on the cell-centred kernels.
bool fused_curv_covers(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas) {
  if (static_cast<long>(P.nk) * P.nj * P.ni >= (1L << 29)) return false;
  if (P.coords == ARTEMIS_CARTESIAN || P.gas.ns != 1 || P.dust.ns != 0 || P.ng < 2) return false;
  if (!g.pcm && recon_gas == ARTEMIS_PPM) return false;
  if (g.drag || g.cooling || g.nbody_n || g.defer_finish) return false;
  if (g.gravity && g.gravity->type != ARTEMIS_GRAVITY_UNIFORM && g.gravity->type != ARTEMIS_GRAVITY_POINT &&
      g.gravity->type != ARTEMIS_GRAVITY_BINARY)
    return false;
  return true;
}

namespace {
template <int RIEMANN, int RECON>
void launch_curv(const PackView &P, const StageK &k, const SrcArg<true> &src, bool has_u1, bool dt, hipStream_t s) {
  const dim3 grid(k.start[k.nbox]);
  const dim3 block(FTX, FTY);
#define GO(U, D)                                                                           \
  do {                                                                                     \
    if (P.ndim > 2) hipLaunchKernelGGL((stage_fused_kernel<RIEMANN, RECON, U, false, D, true, true>), grid, block, 0, s, P, k, src); \
    else hipLaunchKernelGGL((stage_fused_kernel<RIEMANN, RECON, U, false, D, false, true>), grid, block, 0, s, P, k, src); \
  } while (0)
  if (has_u1) { if (dt) GO(true, true); else GO(true, false); }
  else { if (dt) GO(false, true); else GO(false, false); }
#undef GO
}
} // namespace

void launch_stage_fused_curv(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas, int riemann_gas,
                             hipStream_t s) {
  StageK k;
  k.gam0 = g.gam0, k.gam1 = g.gam1, k.beta_dt = g.beta_dt, k.bdt = g.bdt, k.cfl = g.cfl_gas;
  k.bdt_ptr = g.beta_dt_dev;
  k.prim_in = g.gas_in, k.prim_u1 = g.gas_u1, k.prim_out = g.gas_out, k.cons_out = nullptr;
  k.dt_bits = reinterpret_cast<unsigned long long *>(g.dt_dev);
  const int nz = P.ke - P.ks + 1;
  const int NTI = (P.ie - P.is + FTX) / FTX, NTJ = (P.je - P.js + FTY) / FTY;
  const int target_chunk = pick_chunk(nz, static_cast<long>(NTI) * NTJ * P.nb, 256, 64);
  k.nbox = 1, k.start[0] = 0;
  k.ti0[0] = 0, k.nti[0] = NTI, k.tj0[0] = 0, k.ntj[0] = NTJ, k.kb0[0] = P.ks, k.kb1[0] = P.ke;
  k.nchunk[0] = (P.ndim > 2) ? std::max(1, nz / target_chunk) : 1;
  k.kchunk[0] = (nz + k.nchunk[0] - 1) / k.nchunk[0];
  k.nchunk[0] = (nz + k.kchunk[0] - 1) / k.kchunk[0];
  k.start[1] = NTI * NTJ * k.nchunk[0] * P.nb;
  k.nshell = 0, k.shell_done = nullptr;
  k.redo_cnt0 = k.redo_cnt1 = nullptr, k.redo_list0 = k.redo_list1 = nullptr, k.redo_cap = 0;
  k.tiny_in = nullptr, k.tiny_out = nullptr;
  k.outflow = 0, k.outflow_blk = 0, k.outflow_pb = 0;
  k.xcd_swizzle = opt(OPT_FUSED_NO_SWIZZLE) ? 0 : 1;
  SrcArg<true> src;
  src.v.grav_on = (g.gravity && (g.time >= g.gravity->tstart) && (g.time < g.gravity->tstop)) ? 1 : 0;
  if (src.v.grav_on) src.v.grav = *g.gravity;
  src.v.rfc_on = (g.rf_omega != 0.0) ? 1 : 0, src.v.rf_omega = g.rf_omega;
  src.v.diff_on = (g.diffusion != nullptr) ? 1 : 0;
  src.v.do_viscosity = (g.diffusion && g.diffusion->visc.type != ARTEMIS_DIFF_OFF) ? 1 : 0;
  src.v.dsum = src.v.diff_on ? g.diffusion_sums : nullptr;
  const bool has_u1 = (g.gas_u1 != g.gas_in);
  const bool dt = (g.dt_dev != nullptr);
  const int recon = g.pcm ? ARTEMIS_PCM : recon_gas;
#define RC(RS)                                                                             \
  case RS:                                                                                 \
    if (recon == ARTEMIS_PCM) launch_curv<RS, 0>(P, k, src, has_u1, dt, s);                \
    else launch_curv<RS, 1>(P, k, src, has_u1, dt, s);                                     \
    break;
  switch (riemann_gas) {
    RC(0)
    RC(1)
    RC(2)
  }
#undef RC
}

#ifdef FUSED_PROF
extern "C" int artemis_hip_debug_fused_prof(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_fused_prof), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif
} // namespace artemis
