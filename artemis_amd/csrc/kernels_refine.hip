// Mesh-refinement data-path operators of the reference (utils/refinement): RestrictAverage<GEOM>
// (restriction.hpp:42-114, volume-weighted mean of the 2^ndim fine zones of a coarse zone, pairwise
// summation order kept) and ProlongateSharedMinMod<GEOM> (prolongation.hpp:39-184, one minmod-limited
// gradient per direction from the coarse neighbours, evaluated at the fine centroids).  One thread per
// coarse zone; cell-centred fields; every coordinate system through the same DCoords as the hydro path.
#include "device_math.hpp"
#include "geometry_core.hpp"
#include "kernels.hpp"

namespace artemis {
namespace {

struct RefineView {
  artemis_refine_t r;
  long fN, cN;
};
ADEV DCoords fine_coords(const artemis_refine_t &r, int k, int j, int i) {
  return coords_of(r.coords, r.fgeom, r.fmetric, r.fnj, r.fnk, k, j, i);
}
ADEV DCoords coarse_coords(const artemis_refine_t &r, int k, int j, int i) {
  return coords_of(r.coords, r.cgeom, r.cmetric, r.cnj, r.cnk, k, j, i);
}
ADEV double centre_of(const DCoords &co, int d) { return d == 1 ? co.x1v() : (d == 2 ? co.x2v() : co.x3v()); }

#define COARSE_CELL                                                                         \
  const artemis_refine_t &r = R.r;                                                          \
  const int nci = r.cie - r.cis + 1, ncj = r.cje - r.cjs + 1, nck = r.cke - r.cks + 1;      \
  const long t = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;                  \
  if (t >= static_cast<long>(nci) * ncj * nck) return;                                      \
  const int ci = r.cis + static_cast<int>(t % nci), cj = r.cjs + static_cast<int>((t / nci) % ncj); \
  const int ck = r.cks + static_cast<int>(t / (static_cast<long>(nci) * ncj));              \
  const bool X1 = r.ndim > 0, X2 = r.ndim > 1, X3 = r.ndim > 2;                             \
  const int fi = X1 ? (ci - r.cib) * 2 + r.fib : r.fib;                                     \
  const int fj = X2 ? (cj - r.cjb) * 2 + r.fjb : r.fjb;                                     \
  const int fk = X3 ? (ck - r.ckb) * 2 + r.fkb : r.fkb;                                     \
  auto fidx = [&](int k, int j, int i) { return (static_cast<long>(k) * r.fnj + j) * r.fni + i; }; \
  auto cidx = [&](int k, int j, int i) { return (static_cast<long>(k) * r.cnj + j) * r.cni + i; };

__global__ __launch_bounds__(256) void restrict_kernel(const RefineView R) {
  COARSE_CELL
  double vol[2][2][2];
  for (int ok = 0; ok < 2; ++ok)
    for (int oj = 0; oj < 2; ++oj)
      for (int oi = 0; oi < 2; ++oi) vol[ok][oj][oi] = 0;
  for (int ok = 0; ok < 1 + X3; ++ok)
    for (int oj = 0; oj < 1 + X2; ++oj)
      for (int oi = 0; oi < 1 + X1; ++oi) vol[ok][oj][oi] = fine_coords(r, fk + ok, fj + oj, fi + oi).volume();
  const double tvol = ((vol[0][0][0] + vol[0][1][0]) + (vol[0][0][1] + vol[0][1][1])) +
                      ((vol[1][0][0] + vol[1][1][0]) + (vol[1][0][1] + vol[1][1][1]));
  for (int v = 0; v < r.nvar; ++v) {
    const double *q = r.fine[v];
    double terms[2][2][2];
    for (int ok = 0; ok < 2; ++ok)
      for (int oj = 0; oj < 2; ++oj)
        for (int oi = 0; oi < 2; ++oi) terms[ok][oj][oi] = 0;
    for (int ok = 0; ok < 1 + X3; ++ok)
      for (int oj = 0; oj < 1 + X2; ++oj)
        for (int oi = 0; oi < 1 + X1; ++oi)
          terms[ok][oj][oi] = vol[ok][oj][oi] * q[fidx(fk + ok, fj + oj, fi + oi)];
    r.coarse[v][cidx(ck, cj, ci)] =
        (((terms[0][0][0] + terms[0][1][0]) + (terms[0][0][1] + terms[0][1][1])) +
         ((terms[1][0][0] + terms[1][1][0]) + (terms[1][0][1] + terms[1][1][1]))) /
        tvol;
  }
}

__global__ __launch_bounds__(256) void prolongate_kernel(const RefineView R) {
  COARSE_CELL
  // GetGridSpacings<GEOM, d> (prolongation.hpp:39-68): geometry only, shared by the variables
  double dxm[3] = {1, 1, 1}, dxp[3] = {1, 1, 1}, dxfm[3] = {0, 0, 0}, dxfp[3] = {0, 0, 0};
  for (int d = 1; d <= r.ndim; ++d) {
    const int dk = (d == 3), dj = (d == 2), di = (d == 1);
    const double xm = centre_of(coarse_coords(r, ck - dk, cj - dj, ci - di), d);
    const double xc = centre_of(coarse_coords(r, ck, cj, ci), d);
    const double xp = centre_of(coarse_coords(r, ck + dk, cj + dj, ci + di), d);
    const double fxm = centre_of(fine_coords(r, fk, fj, fi), d);
    const double fxp = centre_of(fine_coords(r, fk + dk, fj + dj, fi + di), d);
    dxm[d - 1] = xc - xm, dxp[d - 1] = xp - xc, dxfm[d - 1] = xc - fxm, dxfp[d - 1] = fxp - xc;
  }
  for (int v = 0; v < r.nvar; ++v) {
    const double *q = r.coarse[v];
    const double fc = q[cidx(ck, cj, ci)];
    double g[3] = {0, 0, 0};
    for (int d = 1; d <= r.ndim; ++d) { // GradMinMod (:73-80); SIGN(a) = (a < 0) ? -1 : 1 (parthenon, upstream)
      const int dk = (d == 3), dj = (d == 2), di = (d == 1);
      const double gxm = (fc - q[cidx(ck - dk, cj - dj, ci - di)]) / dxm[d - 1];
      const double gxp = (q[cidx(ck + dk, cj + dj, ci + di)] - fc) / dxp[d - 1];
      const double sm = (gxm < 0.) ? -1. : 1., sp = (gxp < 0.) ? -1. : 1.;
      const double am = fabs(gxm), ap = fabs(gxp);
      g[d - 1] = 0.5 * (sm + sp) * ((ap < am) ? ap : am); // std::min(|gxm|, |gxp|)
    }
    const double gx1m = g[0], gx1p = g[0], gx2m = g[1], gx2p = g[1], gx3m = g[2], gx3p = g[2];
    const double dx1fm = dxfm[0], dx1fp = dxfp[0], dx2fm = dxfm[1], dx2fp = dxfp[1], dx3fm = dxfm[2], dx3fp = dxfp[2];
    double *o = r.fine[v];
    o[fidx(fk, fj, fi)] = fc - (gx1m * dx1fm + gx2m * dx2fm + gx3m * dx3fm);
    if (X1) o[fidx(fk, fj, fi + 1)] = fc + (gx1p * dx1fp - gx2m * dx2fm - gx3m * dx3fm);
    if (X2) o[fidx(fk, fj + 1, fi)] = fc - (gx1m * dx1fm - gx2p * dx2fp + gx3m * dx3fm);
    if (X2 && X1) o[fidx(fk, fj + 1, fi + 1)] = fc + (gx1p * dx1fp + gx2p * dx2fp - gx3m * dx3fm);
    if (X3) o[fidx(fk + 1, fj, fi)] = fc - (gx1m * dx1fm + gx2m * dx2fm - gx3p * dx3fp);
    if (X3 && X1) o[fidx(fk + 1, fj, fi + 1)] = fc + (gx1p * dx1fp - gx2m * dx2fm + gx3p * dx3fp);
    if (X3 && X2) o[fidx(fk + 1, fj + 1, fi)] = fc - (gx1m * dx1fm - gx2p * dx2fp - gx3p * dx3fp);
    if (X3 && X2 && X1) o[fidx(fk + 1, fj + 1, fi + 1)] = fc + (gx1p * dx1fp + gx2p * dx2fp + gx3p * dx3fp);
  }
}

// ---- refinement criteria (amr_criteria.hpp:28-168): one thread per zone, wave maximum, one atomic ----
// std::max(l, eps) = (l < eps) ? eps : l drops a NaN eps; so does the comparison below.  The values are
// non-negative (or dropped), so the IEEE bit pattern orders like an unsigned integer.
struct CriterionView {
  artemis_amr_criterion_t a;
};
ADEV void block_max(double m, double *out) {
  for (int o = 32; o > 0; o >>= 1) {
    const double other = __shfl_down(m, o, 64);
    m = (m < other) ? other : m;
  }
  if ((threadIdx.x & 63) == 0 && m > 0.0)
    atomicMax(reinterpret_cast<unsigned long long *>(out), static_cast<unsigned long long>(__double_as_longlong(m)));
}
// sie != null: the field is the gas density and the criterion runs on the pressure max(0, gm1 rho sie) (fill_derived.cpp:247)
ADEV void first_derivative_body(const artemis_amr_criterion_t &a, const double *sie = nullptr, double gm1 = 0.0) {
  const bool X3 = a.ndim > 2;
  const int i0 = a.is - 1, j0 = a.js - 1, k0 = X3 ? a.ks - 1 : a.ks;
  const int ni = a.ie - a.is + 3, nj = a.je - a.js + 3, nk = X3 ? a.ke - a.ks + 3 : 1;
  const long t = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  double eps = 0.0;
  if (t < static_cast<long>(ni) * nj * nk) {
    const int i = i0 + static_cast<int>(t % ni), j = j0 + static_cast<int>((t / ni) % nj);
    const int k = k0 + static_cast<int>(t / (static_cast<long>(ni) * nj));
    auto co = [&](int kk, int jj, int ii) { return coords_of(a.coords, a.geom, a.metric, a.nj, a.nk, kk, jj, ii); };
    auto v = [&](int kk, int jj, int ii) {
      const long c_ = (static_cast<long>(kk) * a.nj + jj) * a.ni + ii;
      return sie ? amax(0.0, gm1 * a.field[c_] * sie[c_]) : a.field[c_];
    };
    const double sdx1 = co(k, j, i + 1).x1v() - co(k, j, i - 1).x1v();
    const double sdx2 = co(k, j + 1, i).x2v() - co(k, j - 1, i).x2v();
    const DCoords c = co(k, j, i);
    const double hx1 = 1.0, hx2 = c.hx2v();
    const double g1 = (v(k, j, i + 1) - v(k, j, i - 1)) / sdx1 / hx1;
    const double g2 = (v(k, j + 1, i) - v(k, j - 1, i)) / sdx2 / hx2;
    double e;
    if (X3) {
      const double sdx3 = co(k + 1, j, i).x3v() - co(k - 1, j, i).x3v();
      const double hx3 = c.hx3v();
      const double g3 = (v(k + 1, j, i) - v(k - 1, j, i)) / sdx3 / hx3;
      e = sqrt(g1 * g1 + g2 * g2 + g3 * g3);
      const double w1 = sdx1 * hx1, w2 = sdx2 * hx2, w3 = sdx3 * hx3;
      e /= (v(k, j, i) / sqrt(w1 * w1 + w2 * w2 + w3 * w3));
    } else {
      e = sqrt(g1 * g1 + g2 * g2);
      const double w1 = sdx1 * hx1, w2 = sdx2 * hx2;
      e /= (v(k, j, i) / sqrt(w1 * w1 + w2 * w2));
    }
    eps = (eps < e) ? e : eps;
  }
  block_max(eps, a.scratch);
}
ADEV void magnitude_body(const artemis_amr_criterion_t &a, const double *sie = nullptr, double gm1 = 0.0) {
  const int ni = a.ie - a.is + 1, nj = a.je - a.js + 1, nk = a.ke - a.ks + 1;
  const long t = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  double m = 0.0;
  if (t < static_cast<long>(ni) * nj * nk) {
    const int i = a.is + static_cast<int>(t % ni), j = a.js + static_cast<int>((t / ni) % nj);
    const int k = a.ks + static_cast<int>(t / (static_cast<long>(ni) * nj));
    const long c_ = (static_cast<long>(k) * a.nj + j) * a.ni + i;
    const double q = sie ? amax(0.0, gm1 * a.field[c_] * sie[c_]) : a.field[c_];
    m = (m < q) ? q : m;
  }
  block_max(m, a.scratch);
}
__global__ __launch_bounds__(256) void first_derivative_kernel(const CriterionView C) { first_derivative_body(C.a); }
__global__ __launch_bounds__(256) void magnitude_kernel(const CriterionView C) { magnitude_body(C.a); }
// every block of a pack in one launch (blockIdx.y = mesh block): the criterion of variable `var` of the gas
// primitives, maxima[b] receives the block maximum
struct PackCriterion {
  artemis_amr_criterion_t a; // geom / metric / field / scratch are filled per block in the kernel
  double *const *prim;
  int nvar, var, var_sie; // var_sie >= 0: pressure recomputed from density (var) and sie
  double gm1;
  const double *geom, *metric;
  long metric_stride;
  double *maxima;
};
template <bool MAGNITUDE>
__global__ __launch_bounds__(256) void pack_criterion_kernel(const PackCriterion C) {
  artemis_amr_criterion_t a = C.a;
  const int b = blockIdx.y;
  a.geom = C.geom + 6 * b, a.metric = C.metric ? C.metric + b * C.metric_stride : nullptr;
  a.field = C.prim[b * C.nvar + C.var], a.scratch = C.maxima + b;
  const double *sie = (C.var_sie >= 0) ? C.prim[b * C.nvar + C.var_sie] : nullptr;
  if constexpr (MAGNITUDE) magnitude_body(a, sie, C.gm1);
  else first_derivative_body(a, sie, C.gm1);
}
} // namespace

void launch_pack_criterion(const PackView &P, int var, int var_sie, int magnitude, double *maxima, hipStream_t s) {
  PackCriterion C;
  C.a.coords = P.coords, C.a.ndim = P.ndim, C.a.ni = P.ni, C.a.nj = P.nj, C.a.nk = P.nk;
  C.a.is = P.is, C.a.ie = P.ie, C.a.js = P.js, C.a.je = P.je, C.a.ks = P.ks, C.a.ke = P.ke;
  C.a.refine_thr = C.a.deref_thr = 0.0;
  C.prim = P.gas.prim, C.nvar = 6 * P.gas.ns, C.var = var, C.var_sie = var_sie, C.gm1 = P.gm1;
  C.geom = P.geom, C.metric = P.metric, C.metric_stride = metric_block_stride(P.nj, P.nk);
  C.maxima = maxima;
  (void)hipMemsetAsync(maxima, 0, sizeof(double) * P.nb, s);
  const bool X3 = P.ndim > 2;
  const long n = magnitude ? static_cast<long>(P.ie - P.is + 1) * (P.je - P.js + 1) * (P.ke - P.ks + 1)
                           : static_cast<long>(P.ie - P.is + 3) * (P.je - P.js + 3) * (X3 ? P.ke - P.ks + 3 : 1);
  if (n <= 0 || P.nb <= 0) return;
  const dim3 grid((n + 255) / 256, P.nb);
  if (magnitude) hipLaunchKernelGGL(pack_criterion_kernel<true>, grid, dim3(256), 0, s, C);
  else hipLaunchKernelGGL(pack_criterion_kernel<false>, grid, dim3(256), 0, s, C);
}

void launch_refine(const artemis_refine_t &r, int prolongate, hipStream_t s) {
  RefineView R;
  R.r = r;
  R.fN = static_cast<long>(r.fni) * r.fnj * r.fnk, R.cN = static_cast<long>(r.cni) * r.cnj * r.cnk;
  const long n = static_cast<long>(r.cie - r.cis + 1) * (r.cje - r.cjs + 1) * (r.cke - r.cks + 1);
  if (n <= 0) return;
  if (prolongate) hipLaunchKernelGGL(prolongate_kernel, dim3((n + 255) / 256), dim3(256), 0, s, R);
  else hipLaunchKernelGGL(restrict_kernel, dim3((n + 255) / 256), dim3(256), 0, s, R);
}

// The block maximum lands in a.scratch (zeroed on the stream first); the caller copies it back.
void launch_amr_criterion(const artemis_amr_criterion_t &a, int magnitude, hipStream_t s) {
  CriterionView C;
  C.a = a;
  (void)hipMemsetAsync(a.scratch, 0, sizeof(double), s);
  const bool X3 = a.ndim > 2;
  const long n = magnitude ? static_cast<long>(a.ie - a.is + 1) * (a.je - a.js + 1) * (a.ke - a.ks + 1)
                           : static_cast<long>(a.ie - a.is + 3) * (a.je - a.js + 3) * (X3 ? a.ke - a.ks + 3 : 1);
  if (n <= 0) return;
  if (magnitude) hipLaunchKernelGGL(magnitude_kernel, dim3((n + 255) / 256), dim3(256), 0, s, C);
  else hipLaunchKernelGGL(first_derivative_kernel, dim3((n + 255) / 256), dim3(256), 0, s, C);
}

} // namespace artemis
