// Device glue for geometry_core.hpp: Coords of a PackView cell, PLM_G and its per-cell inputs.
#pragma once
#include "device_math.hpp"
#include "geometry_core.hpp"
#include "pack_view.hpp"

namespace artemis {

GDEV DCoords make_coords(const PackView &P, int b, int k, int j, int i) {
  const double *m = P.metric ? P.metric + b * metric_block_stride(P.nj, P.nk) : nullptr;
  return coords_of(P.coords, P.geom + 6 * b, m, P.nj, P.nk, k, j, i);
}

// x_d centroid of the cell at index idx along direction dir, other indices irrelevant
// (x1v depends on i only, x2v on j only, x3v on k only).
GDEV double centroid_along(const PackView &P, int b, int dir, int idx) {
  DCoords c = (dir == 1)   ? make_coords(P, b, 0, 0, idx)
              : (dir == 2) ? make_coords(P, b, 0, idx, 0)
                           : make_coords(P, b, idx, 0, 0);
  return (dir == 1) ? c.x1v() : ((dir == 2) ? c.x2v() : c.x3v());
}

// PLM_G (plm.hpp:54-73), Mignone (2013) weights.  Returns both face values of cell i.
GDEV void plm_g(double q_im1, double q_i, double q_ip1, double &ql_ip1, double &qr_i, double x_im1,
                double x_i, double x_ip1, double xf0, double xf1, double dx) {
  const double dql = (q_i - q_im1) * dx / (x_i - x_im1);
  const double dqr = (q_ip1 - q_i) * dx / (x_ip1 - x_i);
  const double dq2 = dql * dqr;
  const double cr = (x_ip1 - x_i) / (xf1 - x_i);
  const double cl = (x_i - x_im1) / (x_i - xf0);
  const double dqm =
      (dq2 <= 0.0) ? 0.0
                   : dq2 * (cr * dql + cl * dqr) / (dql * dql + dqr * dqr + dq2 * (cl + cr - 2.0));
  ql_ip1 = q_i + dqm * (xf1 - x_i) / dx;
  qr_i = q_i - dqm * (x_i - xf0) / dx;
}

// What PLM_G needs for the cell (k,j,i) along dir (plm.hpp:93-101, :127-135, :161-169).  The
// geometric quotients cr, cl and the refined reciprocals of the three geometric denominators are
// formed once per cell and shared by every reconstructed variable (device_math.hpp: `div` with a
// shared reciprocal returns the bits of `/`).
struct PlmGeo {
  double xvm, xvc, xvp, xf0, xf1, dx;
  double cr, cl, up, lo; // (x_ip1-x_i)/(xf1-x_i), (x_i-x_im1)/(x_i-xf0), xf1-x_i, x_i-xf0
  Recip ra, rb, rdx;     // 1/(x_i-x_im1), 1/(x_ip1-x_i), 1/dx
};
__device__ __forceinline__ void plm_geo_finish(PlmGeo &g) {
  g.cr = (g.xvp - g.xvc) / (g.xf1 - g.xvc);
  g.cl = (g.xvc - g.xvm) / (g.xvc - g.xf0);
  g.up = g.xf1 - g.xvc, g.lo = g.xvc - g.xf0;
  g.ra = recip(g.xvc - g.xvm), g.rb = recip(g.xvp - g.xvc), g.rdx = recip(g.dx);
}
// PLM_G with the shared geometry: same expression tree as plm_g above.  G = PlmGeo, or any record with the
// fields dx, cr, cl, up, lo, ra, rb, rdx (the fused kernel keeps a compact copy per thread / in LDS).
// MODE selects how the five divisions are carried out -- the quotients are the same bits wherever the
// hand-scheduled division is valid (device_math.hpp):
//   0  IEEE `/` everywhere (always valid);
//   1  the three geometric denominators through their shared refined reciprocals, the limited slope's own
//      quotient with `/` (the per-task and cell-centred kernels);
//   2  every division hand-scheduled.  ONLY where the caller has checked that no velocity of the stencil is
//      tiny-but-nonzero: ahead of a shock the differences decay like 1e-40, 1e-80, 1e-160 ..., the cubic
//      numerator and the squares in the denominator reach the subnormal range while dq2 is still positive,
//      and 0 / 2e-320 must stay 0 (v_rcp_f64 of a subnormal is not usable).
template <int MODE = 1, class G>
__device__ __forceinline__ void plm_g_shared(double q_im1, double q_i, double q_ip1, double &ql_ip1,
                                             double &qr_i, const G &g) {
  double dql, dqr;
  if constexpr (MODE == 0) {
    dql = (q_i - q_im1) * g.dx / g.ra.b, dqr = (q_ip1 - q_i) * g.dx / g.rb.b;
  } else {
    dql = div((q_i - q_im1) * g.dx, g.ra), dqr = div((q_ip1 - q_i) * g.dx, g.rb);
  }
  const double dq2 = dql * dqr;
  const double num = dq2 * (g.cr * dql + g.cl * dqr);
  const double den = dql * dql + dqr * dqr + dq2 * (g.cl + g.cr - 2.0);
  double quo;
  if constexpr (MODE == 2) quo = div(num, den);
  else quo = num / den;
  const double dqm = (dq2 <= 0.0) ? 0.0 : quo;
  if constexpr (MODE == 0) {
    ql_ip1 = q_i + dqm * g.up / g.rdx.b;
    qr_i = q_i - dqm * g.lo / g.rdx.b;
  } else {
    ql_ip1 = q_i + div(dqm * g.up, g.rdx);
    qr_i = q_i - div(dqm * g.lo, g.rdx);
  }
}
// a velocity the hand-scheduled PLM_G (MODE 2) cannot take: non-zero and below 2^-200 (differences of
// admitted values are then 0 or at least 2^-252, their cubes normal)
__device__ __forceinline__ bool tiny_nonzero(double v) { return v != 0.0 && fabs(v) < 0x1p-200; }
GDEV PlmGeo plm_geo(const PackView &P, int b, int dir, int k, int j, int i) {
  PlmGeo g;
  const DCoords c = make_coords(P, b, k, j, i);
  const int idx = (dir == 1) ? i : ((dir == 2) ? j : k);
  g.xvm = centroid_along(P, b, dir, idx - 1);
  g.xvp = centroid_along(P, b, dir, idx + 1);
  if (dir == 1) g.xvc = c.x1v(), g.xf0 = c.x1[0], g.xf1 = c.x1[1], g.dx = c.width1();
  else if (dir == 2) g.xvc = c.x2v(), g.xf0 = c.x2[0], g.xf1 = c.x2[1], g.dx = c.width2();
  else g.xvc = c.x3v(), g.xf0 = c.x3[0], g.xf1 = c.x3[1], g.dx = c.width3();
  plm_geo_finish(g);
  return g;
}

// The same record from the table of artemis_hip_plm_table_fill (pack_view.hpp: nine rows per block and direction):
// the index-only fields are loads, the cell width along the sweep -- Coords::width<dir> = h (x_f1 - x_f0) with
// h = 1, x1v or x1v sin(x2v) -- is formed from the tabulated x1 centroid exactly as width2 / width3 form it, and
// its refined reciprocal follows.  xvm / xvc / xvp / xf0 / xf1 are not filled: plm_g_shared does not read them.
__device__ __forceinline__ PlmGeo plm_geo_tab(const PackView &P, int b, int dir, int k, int j, int i) {
  PlmGeo g;
  const int L = P.plm_len;
  const int idx = (dir == 1) ? i : ((dir == 2) ? j : k);
  const double *t = P.plm_tab + (static_cast<long>(b) * 3 + (dir - 1)) * PLM_TAB_ROWS * L + idx;
  g.cr = t[0], g.cl = t[L], g.up = t[2 * L], g.lo = t[3 * L];
  g.ra.b = t[4 * L], g.ra.y = t[5 * L], g.rb.b = t[6 * L], g.rb.y = t[7 * L];
  g.xvm = g.xvc = g.xvp = g.xf0 = g.xf1 = 0.0;
  const double *ge = P.geom + 6 * b;
  const int sys = P.coords;
  const bool sph23 = (sys == ARTEMIS_SPHERICAL2D || sys == ARTEMIS_SPHERICAL3D);
  const bool sph = sph23 || sys == ARTEMIS_SPHERICAL1D;
  if (dir == 1) {
    const double f0 = ge[0] + i * ge[1], f1 = ge[0] + (i + 1) * ge[1];
    g.dx = 1.0 * (f1 - f0); // width1
  } else {
    const double x1v = P.plm_tab[(static_cast<long>(b) * 3 * PLM_TAB_ROWS + 8) * L + i];
    if (dir == 2) {
      const double f0 = ge[2] + j * ge[3], f1 = ge[2] + (j + 1) * ge[3];
      const double h = (sph || sys == ARTEMIS_CYLINDRICAL) ? x1v : 1.0; // width2
      g.dx = h * (f1 - f0);
    } else {
      const double f0 = ge[4] + k * ge[5], f1 = ge[4] + (k + 1) * ge[5];
      double h = 1.0; // width3
      if (sph23) h = x1v * (P.metric + b * metric_block_stride(P.nj, P.nk))[MT_SINV * (P.nj + 1) + j];
      else if (sys == ARTEMIS_AXISYMMETRIC) h = x1v;
      g.dx = h * (f1 - f0);
    }
  }
  g.rdx = recip(g.dx);
  return g;
}

// ---- per-workgroup geometry tables (the march kernels: kernels_curv.hip, kernels_diffusion.hip) --------------------
// A march along x3 visits the same (i, j) columns plane after plane, and in every system the reference supports the
// metric depends on (x1, x2) only.  Keeping a column's Coords-derived constants in registers costs ~100 doubles per
// thread (one wave per SIMD); recomputing them per plane costs a dozen IEEE divisions per zone.  These tables hold
// what is expensive -- per column index the x1 edges and every x1-only expression with a division in it, per row
// index the x2 edges, the tabulated trigonometry and the x2-only quotients -- once per workgroup in LDS; a thread
// rebuilds the Coords of any zone of its tile (own, halo, perimeter) from one column and one row entry and lets
// DCoordsT<true> do the remaining multiplications.  Filled with the CACHED = false functions: the same bits.
enum { GI_X1LO = 0, GI_X1HI, GI_X1V, GI_RCEN, GI_RFAC, GI_DH2, GI_DH3, GI_NF };
enum { GJ_X2LO = 0, GJ_X2HI, GJ_CF0, GJ_CF1, GJ_SF0, GJ_SF1, GJ_X2C, GJ_SV, GJ_SC, GJ_CV, GJ_DH32, GJ_RDC, GJ_NF };
template <int NX, int NY>
struct GeoTabs {
  double gi[GI_NF][NX]; // column x <-> zone index clamp(ibase + x, 0, ni - 1)
  double gj[GJ_NF][NY]; // row y    <-> zone index clamp(jbase + y, 0, nj - 1)
  double x3f0, dx3;     // the block's x3 edge table {x3f0, dx3}: a global load of them inside the march would be a
                        // vector load (stores are in flight, so no scalar load) with a vmcnt(0) wait behind it -- which
                        // also waits for every prefetch of the trip
};
// Threads [0, NX) fill the columns, threads [64, 64 + NY) the rows (NY <= 64: a second wave); the caller
// synchronises the workgroup before the first read.
template <int NX, int NY>
__device__ __forceinline__ void geotabs_fill(GeoTabs<NX, NY> &G, const PackView &P, int b, int ibase, int jbase, int t) {
  static_assert(NX <= 64 && NY <= 64, "one wave per table");
  if (t == 64 + NY) G.x3f0 = P.geom[6 * b + 4], G.dx3 = P.geom[6 * b + 5];
  if (t < NX) {
    const int ii = min(max(ibase + t, 0), P.ni - 1);
    const DCoords c = make_coords(P, b, 0, 0, ii); // (x1-only members: the x2 / x3 indices do not enter)
    G.gi[GI_X1LO][t] = c.x1[0], G.gi[GI_X1HI][t] = c.x1[1];
    G.gi[GI_X1V][t] = c.x1v(), G.gi[GI_RCEN][t] = c.rcen(), G.gi[GI_RFAC][t] = c.rfac();
    G.gi[GI_DH2][t] = c.dh2dx1(), G.gi[GI_DH3][t] = c.dh3dx1();
  } else if (t >= 64 && t < 64 + NY) {
    const int y = t - 64;
    const int jj = min(max(jbase + y, 0), P.nj - 1);
    const DCoords c = make_coords(P, b, 0, jj, 0);
    G.gj[GJ_X2LO][y] = c.x2[0], G.gj[GJ_X2HI][y] = c.x2[1];
    G.gj[GJ_CF0][y] = c.cf[0], G.gj[GJ_CF1][y] = c.cf[1], G.gj[GJ_SF0][y] = c.sf[0], G.gj[GJ_SF1][y] = c.sf[1];
    G.gj[GJ_X2C][y] = c.x2c, G.gj[GJ_SV][y] = c.sv, G.gj[GJ_SC][y] = c.sc, G.gj[GJ_CV][y] = c.cv;
    G.gj[GJ_DH32][y] = c.dh3dx2();
    G.gj[GJ_RDC][y] = c.sph23() ? recip(fabs(c.cf[0] - c.cf[1])).y : 0.0;
  }
}
// Coords of the zone at (column x, row y) of the tables on plane k (x3 edges from the block's edge table; c3 / s3 =
// cos / sin of the plane's x3 centre where the system has them, else 1 / 0)
template <int NX, int NY>
__device__ __forceinline__ DCoordsT<true> geotabs_coords(const GeoTabs<NX, NY> &G, int sys, int x, int y, int k, double c3,
                                                         double s3) {
  DCoordsT<true> c;
  c.sys = sys;
  c.x1[0] = G.gi[GI_X1LO][x], c.x1[1] = G.gi[GI_X1HI][x];
  c.k_x1v = G.gi[GI_X1V][x], c.k_rcen = G.gi[GI_RCEN][x], c.k_rfac = G.gi[GI_RFAC][x];
  c.k_dh2dx1 = G.gi[GI_DH2][x], c.k_dh3dx1 = G.gi[GI_DH3][x];
  c.x2[0] = G.gj[GJ_X2LO][y], c.x2[1] = G.gj[GJ_X2HI][y];
  c.cf[0] = G.gj[GJ_CF0][y], c.cf[1] = G.gj[GJ_CF1][y], c.sf[0] = G.gj[GJ_SF0][y], c.sf[1] = G.gj[GJ_SF1][y];
  c.x2c = G.gj[GJ_X2C][y], c.sv = G.gj[GJ_SV][y], c.sc = G.gj[GJ_SC][y], c.cv = G.gj[GJ_CV][y];
  c.k_dh3dx2 = G.gj[GJ_DH32][y], c.k_rdc = G.gj[GJ_RDC][y];
  const double f0 = G.x3f0, d3 = G.dx3;
  c.x3[0] = f0 + k * d3, c.x3[1] = f0 + (k + 1) * d3;
  c.c3 = c3, c.s3 = s3;
  return c;
}
} // namespace artemis
