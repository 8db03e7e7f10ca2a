// Fused RK stage for 2-D Cartesian blocks with gas (one species) AND dust (0..2 species) in ONE kernel:
//   Gas/Dust::CalculateFluxes -> ApplyUpdate -> Gas::FluxSource -> ExternalGravity -> RotatingFrameForce
//   (shearing box) -> DragSource (simple_dust) -> SetAuxillaryFields -> ConsToPrim -> EstimateTimestepMesh
// (artemis_driver.cpp:182-255, :279-297) -- SURVEY config 3 (inputs/ssheet: dusty shearing sheet with drag).
// Same contract as artemis_hip_stage_general (primitives in, primitives out, u0 / u1 rebuilt from
// primitives, nothing but the FillGhost variables read) and the same bits: every expression tree is the
// one the cell-centred general stage evaluates (kernels_stage_cell.hip, kernels_sources.hip).
//
// gfx950 shape -- a 1.5-D stream, no LDS, no barrier:
//   * one wave owns a strip of 60 columns (lanes 2..61; lanes 0,1,62,63 carry the two halo columns each side)
//     and marches along x2 through a chunk of rows; row loads are 512-byte coalesced segments;
//   * the x2 sweep lives in registers (rolling rows j, j+1, j+2, the carried left state and the carried
//     flux through the lower x2 face): every x2 slope and Riemann problem is computed exactly once;
//   * the x1 sweep exchanges through the wave's lane shuffles only: a lane limits its own slopes from its
//     neighbours' values, solves the Riemann problem at its LOWER x1 face with the upper face value of the
//     lane to its left, and takes the flux through its upper face from the lane to its right -- every x1
//     slope and Riemann problem inside the strip is computed once as well (the cell-centred general stage
//     solves every face twice and reads every stencil from L2);
//   * gas and dust share the march, so the coupled drag update, SetAuxillaryFields and ConsToPrim happen in
//     registers: no conserved-state round trip through HBM and no finish kernel.
// HBM traffic per cell-stage: 5 + 4 nd reads (+ the same for u1 after stage 1) and 5 + 4 nd writes --
// the algorithmic 8 B x 5 x (6 + 4 nd) minus the pressures and conserved states that are never materialised.
#include <cfloat>

#include "device_math.hpp"
#include "fused_device.hpp"
#include "geometry.hpp"
#include "kernels.hpp"
#include "options.hpp"
#include "pack_view.hpp"
#include "sources_device.hpp"
#include "task_device.hpp"

namespace artemis {
namespace {
using namespace fused;

constexpr int OWN = 60; // owned columns per wave
constexpr int HALO = 2;

struct S2Args {
  double gam0, gam1, beta_dt, bdt;
  const double *bdt_ptr;
  double *const *gin, *const *gu1, *const *gout;
  double *const *din, *const *du1, *const *dout;
  int grav_on, rf_on, drag_on, has_u1;
  int damp_on; // any <gas|dust/damping> rate non-zero (otherwise every ramp is dt * (0 + 0) = +0)
  artemis_gravity_t grav;
  double rf_omega, rf_qshear;
  artemis_drag_t drag;
  double cfl_gas, cfl_dust;
  unsigned long long *dt_bits;
  int rows, nstrip, nchunk;
  // detect-and-redo (as the tile march, kernels_fused.hip): a wave whose five-row window holds a gas or dust velocity
  // below 2^-200, or whose updated momenta come out below 2^-480, does not store that row but lists it (block, strip,
  // row); the EXACT instantiation of this kernel -- IEEE divisions throughout, launched over the list -- computes it.
  unsigned *redo_cnt, *redo_done;
  unsigned long long *redo_list;
  unsigned redo_cap;
  // the `strat` problem's user conditions applied to the rows as they are loaded (artemis_stage_general_args_t.strat_faces):
  // the kernel then reads no ghost zone of the block
  int bc_strat;
  double bc_q, bc_om0;
};

struct Dust4 {
  double d, v1, v2, v3;
};
struct DFlux {
  double d, m1, m2, m3;
};

// Neighbour exchange inside the wave with DPP wave shifts (one VALU move per dword, no LDS crossbar, no
// lgkmcnt wait): lane i reads lane i - 1 (wave_shr:1) or lane i + 1 (wave_shl:1); the wave's end lanes keep
// their own value (they are halo lanes whose results are never used).
ADEV double lane_below(double v) { // the value held by lane - 1
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false));
}
ADEV double lane_above(double v) { // the value held by lane + 1
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false));
}

template <class IDX>
ADEV Dust4 load_dust(const double *r, const double *v1, const double *v2, const double *v3, IDX c) {
  Dust4 q;
  q.d = gld(r, c), q.v1 = gld(v1, c), q.v2 = gld(v2, c), q.v3 = gld(v3, c);
  return q;
}
template <int RIEMANN, int DIR>
ADEV DFlux solve_dust(const Dust4 &L, const Dust4 &R) {
  Prim4 l, r;
  l.d = L.d, r.d = R.d;
  if constexpr (DIR == 1) l.vx = L.v1, l.vy = L.v2, l.vz = L.v3, r.vx = R.v1, r.vy = R.v2, r.vz = R.v3;
  else l.vx = L.v2, l.vy = L.v3, l.vz = L.v1, r.vx = R.v2, r.vy = R.v3, r.vz = R.v1;
  FaceFlux F;
  riemann_dust<RIEMANN>(l, r, F);
  DFlux o;
  o.d = F.fd;
  if constexpr (DIR == 1) o.m1 = F.fmx, o.m2 = F.fmy, o.m3 = F.fmz;
  else o.m2 = F.fmx, o.m3 = F.fmy, o.m1 = F.fmz;
  return o;
}

#define G6(X) X(d) X(v1) X(v2) X(v3) X(p) X(e)
#define D4(X) X(d) X(v1) X(v2) X(v3)

// x1 sweep of one row through the wave: fluxes through the lower and the upper x1 face of the lane's cell
template <int RIEMANN, int RECON, bool FAST>
ADEV void x1_gas(const GasK &gk, const Cell6 &q, Flux8 &lo, Flux8 &up) {
  constexpr bool fast = FAST;
  Cell6 L, R;
#define SW(m)                                                                    \
  {                                                                              \
    const double s_ = slope_sel<RECON>(lane_below(q.m), q.m, lane_above(q.m), fast); \
    R.m = lo_val<RECON>(q.m, s_);                                                \
    L.m = lane_below(up_val<RECON>(q.m, s_));                                    \
  }
  G6(SW)
#undef SW
  lo = solve_face<RIEMANN, 1>(gk, L, R, fast);
  up.d = lane_above(lo.d), up.m1 = lane_above(lo.m1), up.m2 = lane_above(lo.m2), up.m3 = lane_above(lo.m3);
  up.e = lane_above(lo.e), up.eg = lane_above(lo.eg), up.pf = lane_above(lo.pf), up.vf = lane_above(lo.vf);
}
template <int RIEMANN, int RECON, bool FAST>
ADEV void x1_dust(const Dust4 &q, DFlux &lo, DFlux &up) {
  constexpr bool fast = FAST;
  Dust4 L, R;
#define SW(m)                                                                    \
  {                                                                              \
    const double s_ = slope_sel<RECON>(lane_below(q.m), q.m, lane_above(q.m), fast); \
    R.m = lo_val<RECON>(q.m, s_);                                                \
    L.m = lane_below(up_val<RECON>(q.m, s_));                                    \
  }
  D4(SW)
#undef SW
  lo = solve_dust<RIEMANN, 1>(L, R);
  up.d = lane_above(lo.d), up.m1 = lane_above(lo.m1), up.m2 = lane_above(lo.m2), up.m3 = lane_above(lo.m3);
}

// Division by UNFLOORED state (post-update densities): when every lane of the wave holds a comfortably normal
// positive denominator -- always, in a healthy run -- the shared-reciprocal form is used (same bits as `/`);
// otherwise the whole wave takes the IEEE division, so zero / negative / denormal densities behave exactly
// as in the reference arithmetic.  The branch is wave-uniform.
ADEV bool normal_pos(double x) { return x > 1.0e-280 && x < 1.0e280; }

// Cartesian ToCylWithVec of a cell centre (geometry.hpp:289-306): radius and the first components of the
// three basis rows, as DragSource uses them for the damping target velocity
struct CylV {
  double R, e1, e2, e3;
};
ADEV CylV cyl_vec_cart(const double xv[3]) {
  CylV c;
  const double R = sqrt(xv[0] * xv[0] + xv[1] * xv[1]);
  c.R = R, c.e1 = xv[0] / (R + 1e-99), c.e2 = xv[1] / (R + 1e-99), c.e3 = 0.0;
  return c;
}
ADEV void damping_ramps2(const artemis_damping_t &p, const artemis_drag_t &D, int ndim, const double xv[3], double dt,
                         double f[3]) { // drag.hpp:68-117, as kernels_sources.hip
  const int multi_d = (ndim >= 2), three_d = (ndim == 3);
  f[0] = dt * (p.irate[0] * ((xv[0] < p.ix[0]) * sqr((xv[0] - p.ix[0]) / (p.ix[0] - D.xmin[0]))) +
               p.orate[0] * ((xv[0] > p.ox[0]) * sqr((xv[0] - p.ox[0]) / (p.ox[0] - D.xmax[0]))));
  f[1] = multi_d * dt *
         (p.irate[1] * ((xv[1] < p.ix[1]) * sqr((xv[1] - p.ix[1]) / (p.ix[1] - D.xmin[1]))) +
          p.orate[1] * ((xv[1] > p.ox[1]) * sqr((xv[1] - p.ox[1]) / (p.ox[1] - D.xmax[1]))));
  f[2] = three_d * dt *
         (p.irate[2] * ((xv[2] < p.ix[2]) * sqr((xv[2] - p.ix[2]) / (p.ix[2] - D.xmin[2]))) +
          p.orate[2] * ((xv[2] > p.ox[2]) * sqr((xv[2] - p.ox[2]) / (p.ox[2] - D.xmax[2]))));
}

// ---- the `strat` problem's conditions on a row in registers (pgen/strat.hpp:158-466, as kernels_unfused.hip's
// strat_bc_kernel fills them into the ghost zones) ------------------------------------------------------------------------
// x1 `extrap` (:188-226, :262-299): density, sie and v3 of the first active zone, v1 copied unless it points into the
// domain, v2 continued linearly in x1v through the first two active zones.  la / lb: the lanes that hold those two zones
// (wave-uniform); `mine`: this lane is a ghost column of that side; x, x0, x1: the zone centres.
ADEV double rd_lane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <class Q>
ADEV void strat_x1(Q &q, bool with_e, double &e_slot, const bool mine, const int upper, const int la, const int lb, const double x,
                   const double x0, const double x1) {
  const double d = rd_lane(q.d, la), v1 = rd_lane(q.v1, la), v2 = rd_lane(q.v2, la), v3 = rd_lane(q.v3, la);
  const double v2n = rd_lane(q.v2, lb);
  const double dx = upper ? (x0 - x1) : (x1 - x0);
  const double vx1 = upper ? ((v1 < 0.0) ? 0.0 : v1) : ((v1 > 0.0) ? 0.0 : v1);
  const double vx2 = upper ? (v2 + (v2 - v2n) * (x - x0) / dx) : (v2 + (v2n - v2) * (x - x0) / dx);
  double es = 0.0;
  if (with_e) es = rd_lane(e_slot, la);
  if (mine) {
    q.d = d, q.v1 = vx1, q.v2 = vx2, q.v3 = v3;
    if (with_e) e_slot = es;
  }
}
// x2 `inflow` (:352-392, :437-466): a copy of the first active row with v2 = -q Om0 x1v where the shear carries material
// into the box (lower face: x1f >= 0, upper face: x1f < 0) and one-way outflow elsewhere
ADEV double strat_x2_v2(const double v2, const int upper, const double xf, const double vy0) {
  return upper ? ((xf < 0) ? ((v2 < 0.0) ? 0.0 : v2) : vy0) : ((xf >= 0) ? ((v2 > 0.) ? 0.0 : v2) : vy0);
}

// One wave per SIMD (launch bound 1): the march keeps ~360 registers live (three rows of both fluids, the carried
// face states and fluxes, one HLLC problem in flight); at two waves per SIMD the same code spills 200-400 bytes per
// lane to scratch and measures 23 % slower (3.07 vs 3.98e9 zone-cycles/s on config 3 at 4096^2).  A single wave
// still hides HBM latency: each trip issues the loads of row j + 2 first and consumes them ~1500 instructions later.
template <int RG, int RD, int RECON, int ND, bool HAS_U1, bool DRAG, bool EXACT = false>
__global__ __launch_bounds__(256, 1) void stage2d_kernel(const PackView P, const S2Args a) {
  const int lane = threadIdx.x & 63;
  const long wave = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const long per_block = static_cast<long>(a.nstrip) * a.nchunk;
  double ldt_g = DBL_MAX, ldt_d = DBL_MAX;
  constexpr bool fast = !EXACT; // compile-time: the EXACT instantiation divides with `/` and takes plm_dqm / hllc_gas
  // EXACT: one listed (block, strip, row) per wave and trip of this loop; otherwise one (block, strip, chunk) per wave
  const long nwork = EXACT ? static_cast<long>(min(__hip_atomic_load(a.redo_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a.redo_cap))
                           : per_block * P.nb;
  for (long work = wave; work < nwork; work += EXACT ? static_cast<long>(gridDim.x) * 4 : nwork) {
    int b, strip, j0, j1;
    if constexpr (EXACT) {
      const unsigned long long e = a.redo_list[work];
      const unsigned bs = static_cast<unsigned>(e >> 32);
      b = static_cast<int>(bs / static_cast<unsigned>(a.nstrip)), strip = static_cast<int>(bs % static_cast<unsigned>(a.nstrip));
      j0 = j1 = static_cast<int>(e & 0xffffffffu);
    } else {
      b = static_cast<int>(work / per_block);
      const int w = static_cast<int>(work - b * per_block);
      strip = w % a.nstrip;
      const int chunk = w / a.nstrip;
      j0 = P.js + chunk * a.rows, j1 = min(P.je, j0 + a.rows - 1);
    }
    const int i = P.is + strip * OWN + lane - HALO;
    const int il = min(max(i, 0), P.ni - 1);
    const bool owned = (lane >= HALO) && (lane < HALO + OWN) && (i <= P.ie);
    double beta_dt = a.beta_dt, bdt = a.bdt;
    if (a.bdt_ptr) beta_dt = bdt = *a.bdt_ptr;
    const GasK gk = gas_constants(P.gm1);
    const double gm1 = P.gm1;
    const FluidView &fg = P.gas, &fd = P.dust;
    const double *g_r = a.gin[b * 6 + 0], *g_1 = a.gin[b * 6 + 1], *g_2 = a.gin[b * 6 + 2], *g_3 = a.gin[b * 6 + 3];
    const double *g_e = a.gin[b * 6 + 5];
    const double *u_r = a.gu1[b * 6 + 0], *u_1 = a.gu1[b * 6 + 1], *u_2 = a.gu1[b * 6 + 2], *u_3 = a.gu1[b * 6 + 3];
    const double *u_e = a.gu1[b * 6 + 5];
    const double *d_r[ND > 0 ? ND : 1], *d_1[ND > 0 ? ND : 1], *d_2[ND > 0 ? ND : 1], *d_3[ND > 0 ? ND : 1];
    const double *e_r[ND > 0 ? ND : 1], *e_1[ND > 0 ? ND : 1], *e_2[ND > 0 ? ND : 1], *e_3[ND > 0 ? ND : 1];
#pragma unroll
    for (int n = 0; n < ND; ++n) {
      d_r[n] = a.din[b * 4 * ND + n], d_1[n] = a.din[b * 4 * ND + ND + 3 * n + 0];
      d_2[n] = a.din[b * 4 * ND + ND + 3 * n + 1], d_3[n] = a.din[b * 4 * ND + ND + 3 * n + 2];
      e_r[n] = a.du1[b * 4 * ND + n], e_1[n] = a.du1[b * 4 * ND + ND + 3 * n + 0];
      e_2[n] = a.du1[b * 4 * ND + ND + 3 * n + 1], e_3[n] = a.du1[b * 4 * ND + ND + 3 * n + 2];
    }
    // element offsets in 32 bits (fused_device.hpp gld / gst: SGPR base + one VGPR byte offset per cell); stage2d_covers
    // refuses blocks of 2^29 zones or more
    const unsigned sj = static_cast<unsigned>(P.sj);
    const unsigned col = static_cast<unsigned>(il); // k = 0 in a 2-D block
    // the block's edge table, read ONCE: inside the march (stores in flight) a read of it would be a vector load of a
    // uniform address with a vmcnt(0) wait behind it, i.e. a wait for the row prefetch that was just issued
    const double *geo = P.geom + 6 * b;
    const double gl[6] = {geo[0], geo[1], geo[2], geo[3], geo[4], geo[5]};
    // ---- the user conditions of the block's faces, in registers (a.bc_strat): row r of the march is read from the
    // nearest ACTIVE row; ghost columns take the x1 condition of that row's first two active zones (which sit in lanes of
    // this wave), then a ghost row takes the x2 condition -- parthenon's order (x1 over the entire x2 extent, then x2 over
    // the entire x1 extent).  Wave-uniform branches; waves away from the block's edges skip all of it.
    const bool bcs = a.bc_strat != 0;
    // (the strip and the row are wave-uniform: said so explicitly, the tests below are then scalar compares and branches
    //  and nothing of the conditions' arithmetic is alive outside the rare waves / rows that need it)
    const int strip_u = __builtin_amdgcn_readfirstlane(strip);
    const bool edge_x = bcs && (strip_u == 0 || (P.ie - P.is - strip_u * OWN + HALO) < 63);
    auto row_of = [&](int r) { return bcs ? min(max(r, P.js), P.je) : r; };
    auto bc_row = [&](Raw5 &g, Dust4 *dq, int r) {
      const int ru = __builtin_amdgcn_readfirstlane(r);
      const bool ghost_row = ru < P.js || ru > P.je;
      if (!(edge_x || ghost_row)) return;
      auto x1v = [&](int ii) { return 0.5 * ((gl[0] + ii * gl[1]) + (gl[0] + (ii + 1) * gl[1])); };
      const double bx = x1v(i);
      const int lane_is = HALO - strip_u * OWN;               // the lane that holds column is (strip 0: lane 2)
      const int lane_ie = P.ie - P.is - strip_u * OWN + HALO; // ... column ie
      double none = 0.0;
      if (strip_u == 0) {
        const double x0 = x1v(P.is), x1 = x1v(P.is + 1);
        strat_x1(g, true, g.e, i < P.is, 0, lane_is, lane_is + 1, bx, x0, x1);
#pragma unroll
        for (int n = 0; n < ND; ++n) strat_x1(dq[n], false, none, i < P.is, 0, lane_is, lane_is + 1, bx, x0, x1);
      }
      if (lane_ie < 63) {
        const double x0 = x1v(P.ie), x1 = x1v(P.ie - 1);
        strat_x1(g, true, g.e, i > P.ie, 1, lane_ie, lane_ie - 1, bx, x0, x1);
#pragma unroll
        for (int n = 0; n < ND; ++n) strat_x1(dq[n], false, none, i > P.ie, 1, lane_ie, lane_ie - 1, bx, x0, x1);
      }
      if (ghost_row) {
        const double bxf = gl[0] + i * gl[1], bvy0 = -a.bc_q * a.bc_om0 * bx;
        g.v2 = strat_x2_v2(g.v2, ru > P.je, bxf, bvy0);
#pragma unroll
        for (int n = 0; n < ND; ++n) dq[n].v2 = strat_x2_v2(dq[n].v2, ru > P.je, bxf, bvy0);
      }
    };
    // one row of both fluids as the march wants it (the three rows of the priming)
    auto load_row = [&](int r, Cell6 &gq, Dust4 *dq) {
      const unsigned c = col + static_cast<unsigned>(row_of(r)) * sj;
      Raw5 q = load_raw(g_r, g_1, g_2, g_3, g_e, c);
#pragma unroll
      for (int n = 0; n < ND; ++n) dq[n] = load_dust(d_r[n], d_1[n], d_2[n], d_3[n], c);
      if (bcs) bc_row(q, dq, r);
      gq = finish_cell(q, gm1);
    };
    // ---- prime the x2 march: rows j0-2, j0-1, j0 --------------------------------------------------------
    Cell6 qc, qn, zl;
    Dust4 dc[ND > 0 ? ND : 1], dn[ND > 0 ? ND : 1], dzl[ND > 0 ? ND : 1];
    load_row(j0 - 1, qc, dc);
    load_row(j0, qn, dn);
    // Guard (exactness next to vanishing velocities): one bit per row, newest in bit 0 -- some lane
    // of this wave holds a gas or dust velocity below 2^-200 in that row.  A trip whose five-row window (the x2 stencil
    // of row j; the x1 stencil lives in the wave's own lanes) has a bit set takes IEEE divisions throughout.
    unsigned th = 0;
    const bool detect = !EXACT && a.redo_cnt != nullptr; // wave-uniform
    auto tiny6 = [](const Cell6 &q) { return tiny_vel3(q.v1, q.v2, q.v3); };
    auto tiny4 = [](const Dust4 &q) { return tiny_vel3(q.v1, q.v2, q.v3); };
    Flux8 fy_lo;
    fy_lo.d = fy_lo.m1 = fy_lo.m2 = fy_lo.m3 = fy_lo.e = fy_lo.eg = fy_lo.pf = fy_lo.vf = 0.0;
    DFlux dy_lo[ND > 0 ? ND : 1];
    {
      Cell6 qmm;
      Dust4 dmm[ND > 0 ? ND : 1];
      load_row(j0 - 2, qmm, dmm);
      if (detect) th = (__any(tiny6(qmm)) ? 4u : 0u) | (__any(tiny6(qc)) ? 2u : 0u) | (__any(tiny6(qn)) ? 1u : 0u);
#define ZL0(m) zl.m = up_val<RECON>(qc.m, slope_sel<RECON>(qmm.m, qc.m, qn.m, fast));
      G6(ZL0)
#undef ZL0
#pragma unroll
      for (int n = 0; n < ND; ++n) {
        if (detect) th |= (__any(tiny4(dmm[n])) ? 4u : 0u) | (__any(tiny4(dc[n])) ? 2u : 0u) | (__any(tiny4(dn[n])) ? 1u : 0u);
#define ZL0(m) dzl[n].m = up_val<RECON>(dc[n].m, slope_sel<RECON>(dmm[n].m, dc[n].m, dn[n].m, fast));
        D4(ZL0)
#undef ZL0
        dy_lo[n].d = dy_lo[n].m1 = dy_lo[n].m2 = dy_lo[n].m3 = 0.0;
      }
    }
    ShearAcc sa{}; // the shearing-box terms depend on the column only (rotating_frame_impl.hpp:43-60)
    if (a.rf_on) sa = shear_terms(gl, 2, 0, i, a.rf_omega, a.rf_qshear);
    // ... and the block's output arrays
    double *const o_r = a.gout[b * 6 + 0], *const o_1 = a.gout[b * 6 + 1], *const o_2 = a.gout[b * 6 + 2];
    double *const o_3 = a.gout[b * 6 + 3], *const o_e = a.gout[b * 6 + 5];
    double *p_r[ND > 0 ? ND : 1], *p_1[ND > 0 ? ND : 1], *p_2[ND > 0 ? ND : 1], *p_3[ND > 0 ? ND : 1];
#pragma unroll
    for (int n = 0; n < ND; ++n) {
      p_r[n] = a.dout[b * 4 * ND + n], p_1[n] = a.dout[b * 4 * ND + ND + 3 * n + 0];
      p_2[n] = a.dout[b * 4 * ND + ND + 3 * n + 1], p_3[n] = a.dout[b * 4 * ND + ND + 3 * n + 2];
    }
    for (int j = j0 - 1; j <= j1; ++j) { // the first trip only primes the flux through face j0
      const unsigned cnn = col + static_cast<unsigned>(row_of(j + 2)) * sj, ccur = col + static_cast<unsigned>(j) * sj;
      const bool live = (j >= j0); // wave-uniform
      // this trip's HBM loads first; consumed after the sweeps
      // (every load of the trip here, unconditionally, and nothing reads one before the sweeps: the single wave of a SIMD
      // has nobody to hide a memory wait behind)
      const Raw5 rnn = load_raw(g_r, g_1, g_2, g_3, g_e, cnn);
      Dust4 dnn[ND > 0 ? ND : 1];
#pragma unroll
      for (int n = 0; n < ND; ++n) dnn[n] = load_dust(d_r[n], d_1[n], d_2[n], d_3[n], cnn);
      Raw5 g1raw{};           // start-of-step state of row j (stages after the first)
      Dust4 d1raw[ND > 0 ? ND : 1];
      if constexpr (HAS_U1) {
        g1raw = load_raw(u_r, u_1, u_2, u_3, u_e, ccur);
#pragma unroll
        for (int n = 0; n < ND; ++n) d1raw[n] = load_dust(e_r[n], e_1[n], e_2[n], e_3[n], ccur);
      }
      auto GD = [&](double num, const Recip &r) { return fast ? div(num, r) : num / r.b; };
      CellMetric g;
      {
        const CellGeom cg = cell_geom(gl, 0, j, i); // task_device.hpp cell_metric<false>, from the preloaded edge table
        g.ax1[0] = g.ax1[1] = cg.dx2 * cg.dx3, g.ax2[0] = g.ax2[1] = cg.dx1 * cg.dx3, g.ax3[0] = g.ax3[1] = cg.dx1 * cg.dx2;
        g.vol = cg.dx1 * cg.dx2 * cg.dx3;
        g.dx[0] = cg.dx1, g.dx[1] = cg.dx2, g.dx[2] = cg.dx3;
      }
      const double hx[3] = {1.0, 1.0, 1.0};
      // Division: the cell epilogue of the general stage is ~90 IEEE divisions per cell (27 instructions each).
      // Denominators that are geometry or floored state get ONE refined reciprocal shared by every quotient
      // (device_math.hpp `div`: the bits of `/` for operands within 2^+-400); divisions by unfloored state
      // (post-update densities) stay IEEE.
      const Recip rvol = recip(g.vol), rdx0 = recip(g.dx[0]), rdx1 = recip(g.dx[1]);
      // Fluxes are folded into ApplyUpdate's divergence and FluxSource's two terms the moment they exist
      // (same products, same order of additions as the cell-centred stage) so that no face record outlives
      // its sweep: the march is register-bound.
      // ---- gas: x1 sweep of row j through the wave --------------------------------------------------------
      double divf[6] = {0, 0, 0, 0, 0, 0}, tm1 = 0.0, tm2 = 0.0, teg1 = 0.0, teg2 = 0.0;
      if (live) {
        Flux8 lo, up;
        x1_gas<RG, RECON, fast>(gk, qc, lo, up);
        divf[0] = (g.ax1[0] * lo.d - g.ax1[1] * up.d), divf[1] = (g.ax1[0] * lo.m1 - g.ax1[1] * up.m1);
        divf[2] = (g.ax1[0] * lo.m2 - g.ax1[1] * up.m2), divf[3] = (g.ax1[0] * lo.m3 - g.ax1[1] * up.m3);
        divf[4] = (g.ax1[0] * lo.e - g.ax1[1] * up.e), divf[5] = (g.ax1[0] * lo.eg - g.ax1[1] * up.eg);
        tm1 = GD(bdt, rdx0) * (lo.pf - up.pf); // FluxSource (fluid_fluxes.hpp:361-392)
        teg1 = GD(bdt, rvol) * 0.5 * (lo.pf + up.pf) * (g.ax1[1] * up.vf - g.ax1[0] * lo.vf);
      }
      // ---- gas: x2 sweep, registers only: slope of row j+1, face j+1 ---------------------------------------
      // (the prefetched row is first read here, behind the x1 sweep)
      Raw5 rnb = rnn;
      if (bcs) bc_row(rnb, dnn, j + 2);
      const Cell6 qnn = finish_cell(rnb, gm1);
      if (detect) {
        bool t = tiny6(qnn);
#pragma unroll
        for (int n = 0; n < ND; ++n) t = t || tiny4(dnn[n]);
        th = (th << 1) | (__any(t) ? 1u : 0u); // bits 0..4 = rows j+2 .. j-2
      }
      bool skip_row = (th & 31u) != 0u; // wave-uniform: this row goes to the exact kernel
      {
        Cell6 zr, zl_next;
#define ZS(m)                                                \
  {                                                          \
    const double s_ = slope_sel<RECON>(qc.m, qn.m, qnn.m, fast); \
    zr.m = lo_val<RECON>(qn.m, s_);                          \
    zl_next.m = up_val<RECON>(qn.m, s_);                     \
  }
        G6(ZS)
#undef ZS
        const Flux8 up = solve_face<RG, 2>(gk, zl, zr, fast);
        if (live) {
          divf[0] += (g.ax2[0] * fy_lo.d - g.ax2[1] * up.d), divf[1] += (g.ax2[0] * fy_lo.m1 - g.ax2[1] * up.m1);
          divf[2] += (g.ax2[0] * fy_lo.m2 - g.ax2[1] * up.m2), divf[3] += (g.ax2[0] * fy_lo.m3 - g.ax2[1] * up.m3);
          divf[4] += (g.ax2[0] * fy_lo.e - g.ax2[1] * up.e), divf[5] += (g.ax2[0] * fy_lo.eg - g.ax2[1] * up.eg);
          tm2 = GD(bdt, rdx1) * (fy_lo.pf - up.pf);
          teg2 = GD(bdt, rvol) * 0.5 * (fy_lo.pf + up.pf) * (g.ax2[1] * up.vf - g.ax2[0] * fy_lo.vf);
        }
        fy_lo = up, zl = zl_next;
      }
      // ---- sources of the cell -----------------------------------------------------------------------------
      GravAcc ga{};
      DCoords co;
      if (live) {
        if (a.grav_on || DRAG) co = coords_of(ARTEMIS_CARTESIAN, gl, nullptr, P.nj, P.nk, 0, j, i);
        if (a.grav_on) ga = gravity_accel(a.grav, co, 2, bdt);
      }
      // ---- gas: ApplyUpdate, FluxSource, gravity, shearing box ----------------------------------------------
      GasCons u0{};
      if (live) {
        FluidPrim wg;
        wg.rho = qc.d, wg.v1 = qc.v1, wg.v2 = qc.v2, wg.v3 = qc.v3, wg.sie = qc.e;
        u0 = prim_to_cons_gas(fg, qc.d, qc.v1, qc.v2, qc.v3, qc.e, hx);
        GasCons u1 = u0;
        if constexpr (HAS_U1)
          u1 = prim_to_cons_gas(fg, g1raw.d, g1raw.v1, g1raw.v2, g1raw.v3, g1raw.e, hx);
        u0.d = a.gam0 * u0.d + a.gam1 * u1.d + GD(divf[0] * beta_dt, rvol);
        u0.m1 = a.gam0 * u0.m1 + a.gam1 * u1.m1 + GD(divf[1] * beta_dt, rvol);
        u0.m2 = a.gam0 * u0.m2 + a.gam1 * u1.m2 + GD(divf[2] * beta_dt, rvol);
        u0.m3 = a.gam0 * u0.m3 + a.gam1 * u1.m3 + GD(divf[3] * beta_dt, rvol);
        u0.e = a.gam0 * u0.e + a.gam1 * u1.e + GD(divf[4] * beta_dt, rvol);
        u0.eg = a.gam0 * u0.eg + a.gam1 * u1.eg + GD(divf[5] * beta_dt, rvol);
        u0.m1 += tm1;
        u0.eg -= teg1;
        u0.m2 += tm2;
        u0.eg -= teg2;
        if (a.grav_on) gravity_gas(ga, bdt, hx, wg, u0);
        if (a.rf_on) shear_gas(sa, bdt, wg, u0);
      }
      qc = qn, qn = qnn;
      // ---- dust: the same march per species ------------------------------------------------------------------
      DustCons ud[ND > 0 ? ND : 1];
#pragma unroll
      for (int n = 0; n < ND; ++n) {
        double dv[4] = {0, 0, 0, 0};
        if (live) {
          DFlux lo, up;
          x1_dust<RD, RECON, fast>(dc[n], lo, up);
          dv[0] = (g.ax1[0] * lo.d - g.ax1[1] * up.d), dv[1] = (g.ax1[0] * lo.m1 - g.ax1[1] * up.m1);
          dv[2] = (g.ax1[0] * lo.m2 - g.ax1[1] * up.m2), dv[3] = (g.ax1[0] * lo.m3 - g.ax1[1] * up.m3);
        }
        {
          Dust4 dzr, dzl_next;
#define ZS(m)                                                          \
  {                                                                    \
    const double s_ = slope_sel<RECON>(dc[n].m, dn[n].m, dnn[n].m, fast); \
    dzr.m = lo_val<RECON>(dn[n].m, s_);                                \
    dzl_next.m = up_val<RECON>(dn[n].m, s_);                           \
  }
          D4(ZS)
#undef ZS
          const DFlux up = solve_dust<RD, 2>(dzl[n], dzr);
          if (live) {
            dv[0] += (g.ax2[0] * dy_lo[n].d - g.ax2[1] * up.d), dv[1] += (g.ax2[0] * dy_lo[n].m1 - g.ax2[1] * up.m1);
            dv[2] += (g.ax2[0] * dy_lo[n].m2 - g.ax2[1] * up.m2), dv[3] += (g.ax2[0] * dy_lo[n].m3 - g.ax2[1] * up.m3);
          }
          dy_lo[n] = up, dzl[n] = dzl_next;
        }
        if (live) {
          FluidPrim wd;
          wd.rho = dc[n].d, wd.v1 = dc[n].v1, wd.v2 = dc[n].v2, wd.v3 = dc[n].v3, wd.sie = 0.0;
          DustCons v0 = prim_to_cons_dust(fd, dc[n].d, dc[n].v1, dc[n].v2, dc[n].v3, hx), v1 = v0;
          if constexpr (HAS_U1) v1 = prim_to_cons_dust(fd, d1raw[n].d, d1raw[n].v1, d1raw[n].v2, d1raw[n].v3, hx);
          v0.d = a.gam0 * v0.d + a.gam1 * v1.d + GD(dv[0] * beta_dt, rvol);
          v0.m1 = a.gam0 * v0.m1 + a.gam1 * v1.m1 + GD(dv[1] * beta_dt, rvol);
          v0.m2 = a.gam0 * v0.m2 + a.gam1 * v1.m2 + GD(dv[2] * beta_dt, rvol);
          v0.m3 = a.gam0 * v0.m3 + a.gam1 * v1.m3 + GD(dv[3] * beta_dt, rvol);
          if (a.grav_on) gravity_dust(ga, bdt, hx, wd, v0);
          if (a.rf_on) shear_dust(sa, bdt, wd, v0);
          ud[n] = v0;
        }
        dc[n] = dn[n], dn[n] = dnn[n];
      }
      // ---- the cell (j, i): drag, aux, c2p, dt ----------------------------------------------------------------
      if (live && owned) {
        // ---- DragSource simple_dust (drag.hpp:296-482) + SetAuxillaryFields + ConsToPrim, in registers -----
        double en = u0.e, mnew[3] = {u0.m1, u0.m2, u0.m3};
        double dmom[ND > 0 ? ND : 1][3];
#pragma unroll
        for (int n = 0; n < ND; ++n) dmom[n][0] = ud[n].m1, dmom[n][1] = ud[n].m2, dmom[n][2] = ud[n].m3;
        if (detect) { // updated momenta whose quotients the hand-scheduled division would not round like `/`
          bool tm = tiny_mom3(mnew[0], mnew[1], mnew[2]);
#pragma unroll
          for (int n = 0; n < ND; ++n) tm = tm || tiny_mom3(dmom[n][0], dmom[n][1], dmom[n][2]);
          if (__any(tm)) skip_row = true;
          if (skip_row) { // (uniform over the active lanes) one entry per wave and row
            if (lane == __builtin_ctzll(__ballot(1))) {
              const unsigned at = __hip_atomic_fetch_add(a.redo_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (at < a.redo_cap)
                a.redo_list[at] = (static_cast<unsigned long long>(static_cast<unsigned>(b) * static_cast<unsigned>(a.nstrip) + static_cast<unsigned>(strip)) << 32) |
                                  static_cast<unsigned>(j);
            }
            continue; // no store, no dt contribution: the exact kernel does both
          }
        }
        if constexpr (DRAG) {
          const artemis_drag_t &D = a.drag;
          const double xv[3] = {co.x1v(), co.x2v(), co.x3v()};
          const CylV cv = cyl_vec_cart(xv);
          double bg[3] = {0.0, 0.0, 0.0}, bd[3] = {0.0, 0.0, 0.0};
          if (a.damp_on) { // (wave-uniform; twelve divisions per cell otherwise spent on zeros)
            damping_ramps2(D.gas, D, 2, xv, bdt, bg);
            damping_ramps2(D.dust, D, 2, xv, bdt, bd);
          }
          const double dg = u0.d;
          const double mg[3] = {u0.m1, u0.m2, u0.m3};
          bool healthy = normal_pos(dg);
#pragma unroll
          for (int n = 0; n < ND; ++n) healthy = healthy && normal_pos(ud[n].d);
          const bool fastdiv = __all(healthy || !owned) != 0;
          double vg[3];
          if (fastdiv) {
            const Recip rg = recip(dg); // hx * dg == dg (hx = 1)
            vg[0] = GD(mg[0], rg), vg[1] = GD(mg[1], rg), vg[2] = GD(mg[2], rg);
          } else {
            vg[0] = mg[0] / (hx[0] * dg), vg[1] = mg[1] / (hx[1] * dg), vg[2] = mg[2] / (hx[2] * dg);
          }
          double sieg; // GetSpecificInternalEnergy (artemis_utils.hpp:43-62)
          {
            const double u_d = amax(dg, fg.dfloor);
            const Recip ru = recip(u_d); // floored: positive
            const double rv1 = mg[0] / hx[0], rv2 = mg[1] / hx[1], rv3 = mg[2] / hx[2];
            const double ke = GD(0.5 * (sqr(rv1) + sqr(rv2) + sqr(rv3)), ru);
            const double e_cons = u0.e;
            const double ue_cons = e_cons - ke;
            sieg = GD((ue_cons > fg.de_switch * e_cons) ? ue_cons : u0.eg, ru);
            sieg = amax(sieg, fg.siefloor);
          }
          const double mu = 0.0; // damp_to_visc is not routed here (stage2d_covers refuses it)
          const double vR = -1.5 * mu / (cv.R * dg);
          const double vt[3] = {cv.e1 * vR, cv.e2 * vR, cv.e3 * vR};
          double fdd[3] = {0., 0., 0.}, fvd[3] = {0., 0., 0.};
          double vth = 0.0;
          const bool stokes = (D.model == ARTEMIS_DRAG_STOKES);
          if (stokes) vth = sqrt(8.0 / M_PI * gm1 * sieg);
          const double vdt[3] = {0.0, 0.0, 0.0};
          // per species: velocity (IEEE divisions by the unfloored density, evaluated once: the second pass of
          // the reference re-evaluates the same expressions), coupling strength, and rhop = dens alpha /
          // (1 + alpha + bd) whose denominator is >= 1
          double vdv[ND > 0 ? ND : 1][3], alph[ND > 0 ? ND : 1], rhopv[ND > 0 ? ND : 1][3];
#pragma unroll
          for (int n = 0; n < ND; ++n) {
            const double dens = ud[n].d;
            if (fastdiv) {
              const Recip rdn = recip(dens);
              vdv[n][0] = GD(dmom[n][0], rdn), vdv[n][1] = GD(dmom[n][1], rdn), vdv[n][2] = GD(dmom[n][2], rdn);
            } else {
              vdv[n][0] = dmom[n][0] / (hx[0] * dens), vdv[n][1] = dmom[n][1] / (hx[1] * dens), vdv[n][2] = dmom[n][2] / (hx[2] * dens);
            }
            double tc = D.tau[n];
            if (stokes) tc = D.scale * D.grain_density / dg * D.sizes[n] / vth;
            const double alpha = bdt * ((tc <= 0.0) ? DBL_MAX : 1.0 / tc);
            alph[n] = alpha;
            for (int d = 0; d < 3; d++) {
              const double rhop = GD(dens * alpha, recip(1.0 + alpha + bd[d]));
              rhopv[n][d] = rhop;
              fdd[d] += rhop * (1.0 + bd[d]);
              fvd[d] += rhop * (vdv[n][d] + bd[d] * vdt[d]);
            }
          }
          double vgp[3];
          for (int d = 0; d < 3; d++) {
            const double num = (dg * (vg[d] + bg[d] * vt[d]) + fvd[d]), den = (dg * (1.0 + bg[d]) + fdd[d]);
            vgp[d] = (__all(normal_pos(den) || !owned) != 0) ? GD(num, recip(den)) : num / den;
          }
          double delta_g[3] = {0.0, 0.0, 0.0};
          for (int d = 0; d < 3; d++) fvd[d] = 0.;
#pragma unroll
          for (int n = 0; n < ND; ++n) {
            const double dens = ud[n].d;
            const double *vd = vdv[n];
            const double alpha = alph[n];
            double newm[3];
            for (int d = 0; d < 3; d++) {
              double delta_d = 0.;
              const double rhop = rhopv[n][d];
              const double delta = rhop * ((vgp[d] - vd[d] + bd[d] * (vgp[d] - vdt[d])));
              delta_d += delta;
              delta_g[d] -= delta;
              delta_d -= GD(bd[d] * dens, recip(1. + alpha + bd[d])) * (vd[d] - vdt[d] + alpha * (vgp[d] - vdt[d]));
              fvd[d] += rhop * (vd[d] - vt[d] + bd[d] * (vdt[d] - vt[d]));
              newm[d] = dmom[n][d] + hx[d] * delta_d;
            }
            dmom[n][0] = newm[0], dmom[n][1] = newm[1], dmom[n][2] = newm[2];
          }
          for (int d = 0; d < 3; d++) {
            const double prefac = GD(dg * bg[d], recip(1.0 + bg[d] + fdd[d])); // denominator >= 1
            delta_g[d] -= prefac * (dg * (vg[d] - vt[d]) + fvd[d]);
            mnew[d] = mg[d] + hx[d] * delta_g[d];
            en += 0.5 * (vg[d] + vgp[d]) * delta_g[d];
          }
        }
        // dust ConsToPrim (fill_derived.cpp:155-164)
        const unsigned c = static_cast<unsigned>(j) * sj + static_cast<unsigned>(i);
#pragma unroll
        for (int n = 0; n < ND; ++n) {
          const double dens = ud[n].d;
          const double w_d = (dens > fd.dfloor) ? dens : fd.dfloor;
          const Recip rwd = recip(w_d); // floored; w_d * hx == w_d (hx = 1)
          const double w1 = GD(dmom[n][0], rwd), w2 = GD(dmom[n][1], rwd), w3 = GD(dmom[n][2], rwd);
          gst(p_r[n], c, w_d);
          gst(p_1[n], c, w1);
          gst(p_2[n], c, w2);
          gst(p_3[n], c, w3);
          if (a.dt_bits) { // Dust::EstimateTimestepMesh (dust.cpp:256-272)
            double denom = 0.0;
            denom += GD(fabs(w1), rdx0); // 1.0 * dx == dx
            denom += GD(fabs(w2), rdx1);
            ldt_d = amin(ldt_d, 1.0 / denom); // (IEEE: a dust at rest gives 1 / 0 = inf like the reference)
          }
        }
        // gas SetAuxillaryFields (fill_derived.cpp:58-71) + ConsToPrim (:132-146)
        {
          const double dgas = u0.d;
          const double u_d = (dgas > fg.dfloor) ? dgas : fg.dfloor;
          const double u_d2 = amax(dgas, fg.dfloor);
          const Recip r2 = recip(u_d2), rw = recip(u_d); // floored densities (equal unless the state is NaN)
          const double rv1 = mnew[0] / hx[0], rv2 = mnew[1] / hx[1], rv3 = mnew[2] / hx[2];
          const double ke = GD(0.5 * (sqr(rv1) + sqr(rv2) + sqr(rv3)), r2);
          const double ue_cons = en - ke;
          double sie = GD((ue_cons > fg.de_switch * en) ? ue_cons : u0.eg, r2);
          sie = amax(sie, fg.siefloor);
          double u_u = sie * u_d;
          const double uflr = fg.siefloor * u_d;
          u_u = (u_u > uflr) ? u_u : uflr;
          const double w_d = u_d;
          const double w1 = GD(mnew[0], rw), w2 = GD(mnew[1], rw), w3 = GD(mnew[2], rw);
          double w_s = GD(u_u, rw);
          w_s = (w_s > fg.siefloor) ? w_s : fg.siefloor;
          gst(o_r, c, w_d);
          gst(o_1, c, w1), gst(o_2, c, w2), gst(o_3, c, w3);
          gst(o_e, c, w_s);
          if (a.dt_bits) { // Gas::EstimateTimestepMesh (gas.cpp:411-433)
            const double bulk = (gm1 + 1.0) * gm1 * w_d * w_s;
            const double cs = sqrt_pos(GD(bulk, rw)); // positive: floored density and sie
            double denom = 0.0;
            denom += GD(fabs(w1) + cs, rdx0);
            denom += GD(fabs(w2) + cs, rdx1);
            ldt_g = amin(ldt_g, GD(1.0, recip(denom)));
          }
        }
      }
    }
  }
  if (a.dt_bits) { // wave minimum, one atomic per wave and fluid
    for (int off = 32; off > 0; off >>= 1) {
      ldt_g = fmin(ldt_g, __shfl_down(ldt_g, off, 64));
      ldt_d = fmin(ldt_d, __shfl_down(ldt_d, off, 64));
    }
    if (lane == 0) {
      if (ldt_g < DBL_MAX) atomicMin(a.dt_bits, static_cast<unsigned long long>(__double_as_longlong(a.cfl_gas * ldt_g)));
      if (ND > 0 && ldt_d < DBL_MAX)
        atomicMin(a.dt_bits, static_cast<unsigned long long>(__double_as_longlong(a.cfl_dust * ldt_d)));
    }
  }
  if constexpr (EXACT) { // every workgroup has read the count; the last one empties the list for the next stage
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(a.redo_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (ticket == gridDim.x - 1) {
        __hip_atomic_store(a.redo_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.redo_done, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// the rows the calling thread's launches deferred (one counter + ticket and one list, grown on demand outside captures)
struct Redo2d {
  unsigned *cnt = nullptr;
  unsigned long long *list = nullptr;
  size_t cap = 0;
};
thread_local Redo2d g_redo2d;
bool ensure_redo2d(size_t rows) {
  if (g_redo2d.cnt && g_redo2d.cap >= rows) return true;
  (void)hipDeviceSynchronize();
  if (g_redo2d.list) (void)hipFree(g_redo2d.list);
  g_redo2d.list = nullptr, g_redo2d.cap = 0;
  if (!g_redo2d.cnt) {
    if (hipMalloc(reinterpret_cast<void **>(&g_redo2d.cnt), 2 * sizeof(unsigned)) != hipSuccess) return false;
    if (hipMemset(g_redo2d.cnt, 0, 2 * sizeof(unsigned)) != hipSuccess) return false;
  }
  if (hipMalloc(reinterpret_cast<void **>(&g_redo2d.list), rows * sizeof(unsigned long long)) != hipSuccess) return false;
  g_redo2d.cap = rows;
  return true;
}

template <int RG, int RD, int RECON, int ND>
void launch_flags(const PackView &P, const S2Args &a, hipStream_t s) {
  const long waves = static_cast<long>(a.nstrip) * a.nchunk * P.nb;
  const dim3 grid(static_cast<unsigned>((waves + 3) / 4)), block(256);
  // the exact kernel: its waves stride over the listed rows (normally none).  An empty pass of this register-heavy
  // kernel costs by its grid -- 7.4 us at 256 workgroups, 5 % of a 1024^2 stage --, so the grid is sized by the main
  // launch: one exact workgroup per 32 main ones, between 8 and 256
  const int rg_env = static_cast<int>(opt(OPT_STAGE2D_RGRID));
  const unsigned rg = rg_env > 0 ? static_cast<unsigned>(rg_env) : std::min(256u, std::max(8u, grid.x / 32u));
  const dim3 rgrid(rg);
#define GO(U, D)                                                                                                    \
  do {                                                                                                              \
    hipLaunchKernelGGL((stage2d_kernel<RG, RD, RECON, ND, U, D, false>), grid, block, 0, s, P, a);                  \
    if (a.redo_cnt) hipLaunchKernelGGL((stage2d_kernel<RG, RD, RECON, ND, U, D, true>), rgrid, block, 0, s, P, a);  \
  } while (0)
  if constexpr (ND > 0) {
    if (a.has_u1) {
      if (a.drag_on) GO(true, true);
      else GO(true, false);
    } else {
      if (a.drag_on) GO(false, true);
      else GO(false, false);
    }
  } else { // (drag couples gas to dust: no dust, no drag instantiation)
    if (a.has_u1) GO(true, false);
    else GO(false, false);
  }
#undef GO
}
template <int RG, int RD, int RECON>
void launch_nd(const PackView &P, const S2Args &a, hipStream_t s) {
  if (P.dust.ns == 0) launch_flags<RG, RD, RECON, 0>(P, a, s);
  else if (P.dust.ns == 1) launch_flags<RG, RD, RECON, 1>(P, a, s);
  else launch_flags<RG, RD, RECON, 2>(P, a, s);
}
template <int RG, int RD>
void launch_rc(const PackView &P, int recon, const S2Args &a, hipStream_t s) {
  if (recon == ARTEMIS_PCM) launch_nd<RG, RD, 0>(P, a, s);
  else launch_nd<RG, RD, 1>(P, a, s);
}

} // namespace

// Does the 2-D row-march stage cover this call of artemis_hip_stage_general?
bool stage2d_covers(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas, int riemann_gas, int recon_dust,
                    int riemann_dust) {
  if (P.coords != ARTEMIS_CARTESIAN || P.ndim != 2 || P.ng < 2) return false;
  if (static_cast<long>(P.nj) * P.ni >= (1L << 29)) return false; // 32-bit cell offsets
  if (P.gas.ns != 1 || P.dust.ns > 2) return false;
  if (recon_gas == ARTEMIS_PPM || (P.dust.ns && (recon_dust != recon_gas || riemann_dust == ARTEMIS_HLLC))) return false;
  if (g.diffusion || g.cooling || g.nbody_n || g.defer_finish) return false;
  if (g.strat_faces && (g.strat_faces != 15 || P.nb != 1 || P.ie - P.is < 1)) return false; // (all four faces of ONE block)
  if (g.drag && (g.drag->type != ARTEMIS_DRAG_SIMPLE_DUST || g.drag->damp_visc || P.dust.ns == 0)) return false;
  if (g.gravity && g.gravity->type != ARTEMIS_GRAVITY_UNIFORM && g.gravity->type != ARTEMIS_GRAVITY_POINT &&
      g.gravity->type != ARTEMIS_GRAVITY_BINARY)
    return false;
  (void)riemann_gas;
  return true;
}

void launch_stage2d(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas, int riemann_gas, int riemann_dust,
                    hipStream_t s) {
  S2Args a;
  a.gam0 = g.gam0, a.gam1 = g.gam1, a.beta_dt = g.beta_dt, a.bdt = g.bdt, a.bdt_ptr = g.beta_dt_dev;
  a.gin = g.gas_in, a.gu1 = g.gas_u1, a.gout = g.gas_out, a.din = g.dust_in, a.du1 = g.dust_u1, a.dout = g.dust_out;
  a.has_u1 = (g.gas_u1 != g.gas_in) ? 1 : 0;
  a.grav_on = (g.gravity && (g.time >= g.gravity->tstart) && (g.time < g.gravity->tstop)) ? 1 : 0;
  if (a.grav_on) a.grav = *g.gravity;
  a.rf_on = (g.rf_omega != 0.0) ? 1 : 0, a.rf_omega = g.rf_omega, a.rf_qshear = g.rf_qshear;
  a.drag_on = g.drag ? 1 : 0;
  a.damp_on = 0;
  if (g.drag) {
    a.drag = *g.drag;
    for (int d = 0; d < 3; ++d)
      if (g.drag->gas.irate[d] != 0.0 || g.drag->gas.orate[d] != 0.0 || g.drag->dust.irate[d] != 0.0 || g.drag->dust.orate[d] != 0.0)
        a.damp_on = 1;
    // (a ramp with zero rates is dt * (0 * finite + 0 * finite): the thresholds default to -+DBL_MAX and the mesh
    // bounds are finite, so the skipped value is exactly +0 for the positive dt of a step)
  }
  a.cfl_gas = g.cfl_gas, a.cfl_dust = g.cfl_dust;
  a.bc_strat = g.strat_faces ? 1 : 0, a.bc_q = g.strat_qshear, a.bc_om0 = g.strat_omega;
  a.dt_bits = reinterpret_cast<unsigned long long *>(g.dt_dev);
  const int nx1 = P.ie - P.is + 1, nx2 = P.je - P.js + 1;
  a.nstrip = (nx1 + OWN - 1) / OWN;
  // Rows per chunk.  One wave per SIMD (launch bound 1): 256 CUs x 4 waves run at a time, so the launch proceeds in
  // rounds of 1024 strip-chunks, every wave of a round marching rows + 1 trips (one priming trip) after loading three
  // rows -- and a last round that is nearly empty costs as much as a full one.  Take the chunk length with the least
  // rounds x (rows + 2): 4096^2 -> 40 rows, 7 rounds; 1024^2 -> 19 rows, ONE round of 972 waves (round 2 halved the
  // chunks until there were 4096 waves, which made 3078 chunks of 6 rows = four rounds of 8 trips, 1.5 x the time).
  int rows = 32;
  if (opt(OPT_STAGE2D_ROWS) > 0) {
    rows = static_cast<int>(opt(OPT_STAGE2D_ROWS));
  } else {
    const long slots = 1024;
    long best = -1;
    for (int r = std::min(64, nx2); r >= std::min(4, nx2); --r) {
      const long waves = static_cast<long>(a.nstrip) * ((nx2 + r - 1) / r) * P.nb;
      const long cost = ((waves + slots - 1) / slots) * (r + 2);
      if (best < 0 || cost < best) best = cost, rows = r;
    }
  }
  a.rows = rows, a.nchunk = (nx2 + rows - 1) / rows;
  a.redo_cnt = a.redo_done = nullptr, a.redo_list = nullptr, a.redo_cap = 0;
  if (!opt(OPT_NO_REDO) && ensure_redo2d(static_cast<size_t>(P.nb) * a.nstrip * nx2)) {
    a.redo_cnt = g_redo2d.cnt, a.redo_done = g_redo2d.cnt + 1, a.redo_list = g_redo2d.list;
    a.redo_cap = static_cast<unsigned>(std::min<size_t>(g_redo2d.cap, 0xffffffffu));
  }
  const int recon = g.pcm ? ARTEMIS_PCM : recon_gas;
  const int rd = (P.dust.ns && riemann_dust == ARTEMIS_LLF) ? 2 : 1;
#define RR(G_)                                             \
  case G_:                                                 \
    if (rd == 1) launch_rc<G_, 1>(P, recon, a, s);         \
    else launch_rc<G_, 2>(P, recon, a, s);                 \
    break;
  switch (riemann_gas) {
    RR(0)
    RR(1)
    RR(2)
  }
#undef RR
}

} // namespace artemis
