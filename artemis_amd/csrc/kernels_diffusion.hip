// Gas diffusion tasks (artemis_driver.cpp:189-193, :218-221), constant coefficients, every coordinate
// system (CURV instantiations read the metric from geometry.hpp; Coords::Distance uses the
// tabulated trigonometry of the cell centres): ZeroDiffusionFlux, ViscousFlux (momentum_diffusion.hpp), ThermalFlux
// (thermal_diffusion.hpp), DiffusionUpdate and the diffusive timestep (diffusion.hpp).
//
// The reference walks pencils with scratch rows for the strain tensor, div(u) and the
// coefficient; every face value is a pure function of the primitives around the face, so here
// one thread owns one face (thread x walks i, coalesced) and evaluates its neighbourhood
// directly -- the cells involved are shared through L1/L2 with the neighbouring threads.
// In the Cartesian instantiations scale factors are 1 and the connection coefficients 0; the terms
// are kept (multiplications by 1.0 / additions of 0.0 * v) so that NaN/Inf propagate as in the
// reference's arithmetic.
#include <cfloat>
#include <cstring>
#include <cstdlib>

#include "device_math.hpp"
#include "geometry.hpp"
#include "kernels.hpp"
#include "options.hpp"
#include "pack_view.hpp"
#include "task_device.hpp"
#include "diffusion_device.hpp"
#include "sources_device.hpp"
#include "fused_device.hpp"
#include <type_traits>

// -DVS_PROF (development builds only: ARTEMIS_HIPFLAGS_KERNELS_DIFFUSION=-DVS_PROF): per-phase shader-clock sums of the
// viscous-source march; artemis_hip_debug_vs_prof reads them
#ifdef VS_PROF
__device__ unsigned long long g_vs_prof[16];
#define VPROF_DECL unsigned long long prof_t = __builtin_readcyclecounter(), prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define VPROF(slot)                                               \
  do {                                                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    prof_acc[slot] += now_ - prof_t;                              \
    prof_t = now_;                                                \
  } while (0)
#else
#define VPROF_DECL
#define VPROF(slot)
#endif

namespace artemis {
namespace {
constexpr int TX = 64, TY = 4;

struct Box {
  int il, iu, jl, ju, kl, ku;
};
// Thread shape of the one-thread-per-zone kernels: 64 x 4 by default; for narrow mesh blocks (refined meshes
// run 16^3 blocks) the x1 extent of the workgroup shrinks to the next power of two >= nx and the rows it
// frees fold along x2, so that a wave's 64 lanes stay on real zones (a 16-zone row filled a quarter of them).
inline dim3 tile_threads(int nx) {
  int tx = TX;
  while (tx > 8 && tx / 2 >= nx) tx >>= 1;
  return dim3(tx, TX * TY / tx);
}
// (ranges the tiles cover badly are walked flat, 256 consecutive zones per workgroup: see kernels_unfused.hip)
inline bool flat_range(const Box &r) {
  const int nx = r.iu - r.il + 1, ny = r.ju - r.jl + 1;
  const dim3 t = tile_threads(nx);
  const long covered = static_cast<long>((nx + t.x - 1) / t.x) * t.x * ((ny + t.y - 1) / t.y) * t.y;
  return static_cast<long>(nx) * ny * 10 < covered * 6 && !opt(OPT_NO_FLAT_RANGES);
}
inline dim3 threads_of(const Box &r) { return flat_range(r) ? dim3(TX * TY, 1, 1) : tile_threads(r.iu - r.il + 1); }
inline dim3 grid_of(const Box &r, int nb) {
  const int nx = r.iu - r.il + 1, ny = r.ju - r.jl + 1, nz = r.ku - r.kl + 1;
  if (flat_range(r)) return dim3(static_cast<unsigned>((static_cast<long>(nx) * ny * nz + TX * TY - 1) / (TX * TY)), 1, nb);
  const dim3 t = tile_threads(nx);
  return dim3((nx + t.x - 1) / t.x, (ny + t.y - 1) / t.y, nz * nb);
}
struct CellIdx {
  int i, j, k, b;
  bool ok;
};
__device__ __forceinline__ CellIdx cell_of_box(const Box &r) {
  CellIdx q;
  if (blockDim.y == 1) { // flat walk
    const int nx = r.iu - r.il + 1, ny = r.ju - r.jl + 1;
    const int p = blockIdx.x * blockDim.x + threadIdx.x, row = p / nx;
    q.i = r.il + (p - row * nx), q.j = r.jl + row % ny, q.k = r.kl + row / ny, q.b = blockIdx.z;
    q.ok = q.k <= r.ku;
  } else {
    const int nkr = r.ku - r.kl + 1;
    q.i = r.il + blockIdx.x * blockDim.x + threadIdx.x, q.j = r.jl + blockIdx.y * blockDim.y + threadIdx.y;
    q.b = blockIdx.z / nkr, q.k = r.kl + blockIdx.z % nkr;
    q.ok = q.i <= r.iu && q.j <= r.ju;
  }
  return q;
}
#define BOX_CELL(r)                                                                         \
  const CellIdx ci_ = cell_of_box(r);                                                      \
  if (!ci_.ok) return;                                                                     \
  const int i = ci_.i, j = ci_.j, k = ci_.k, b = ci_.b;                                    \
  const long c = (static_cast<long>(k) * P.nj + j) * P.ni + i;

// Geometry of one block as the diffusion tasks need it: cell centres, Coords::Distance between two
// cell centres (geometry.hpp:407-412), volume-averaged scale factors, connection coefficients.
template <bool CURV>
struct Geo {
  const PackView &P;
  int b;
  // optional: the Cartesian images of this block's cell centres (ConvertCoordsToCart, written once per stage by
  // viscous_cell_kernel), so that a Distance is six cached loads instead of two Coords evaluations
  const double *xc0 = nullptr, *xc1 = nullptr, *xc2 = nullptr;
  // optional: the table of artemis_hip_viscous_distance_fill (static geometry, host-owned): [6][nb][N] doubles,
  // q = 0..2 Distance(cell, its lower x1 / x2 / x3 neighbour), q = 3..5 Distance(lower, upper neighbour)
  const double *dtab = nullptr;
  ADEV double dtab_at(int q, long c) const {
    const long N = static_cast<long>(P.ni) * P.nj * P.nk;
    return dtab[(static_cast<long>(q) * P.nb + b) * N + c];
  }
  ADEV const double *g() const { return P.geom + 6 * b; }
  ADEV double x1v(int i) const { return 0.5 * ((g()[0] + i * g()[1]) + (g()[0] + (i + 1) * g()[1])); }
  ADEV double x2v(int j) const { return 0.5 * ((g()[2] + j * g()[3]) + (g()[2] + (j + 1) * g()[3])); }
  ADEV double x3v(int k) const { return 0.5 * ((g()[4] + k * g()[5]) + (g()[4] + (k + 1) * g()[5])); }
  // Distance(cell c, lower neighbour along DIR) / Distance(lower, upper neighbour of c along DIR)
  template <int DIR>
  ADEV double dist_lower(int k, int j, int i) const {
    constexpr int dk = (DIR == 3), dj = (DIR == 2), di = (DIR == 1);
    if (dtab) return dtab_at(DIR - 1, (static_cast<long>(k) * P.nj + j) * P.ni + i);
    return dist(k, j, i, k - dk, j - dj, i - di);
  }
  template <int DIR>
  ADEV double dist_across(int k, int j, int i) const {
    constexpr int dk = (DIR == 3), dj = (DIR == 2), di = (DIR == 1);
    if (dtab) return dtab_at(2 + DIR, (static_cast<long>(k) * P.nj + j) * P.ni + i);
    return dist(k - dk, j - dj, i - di, k + dk, j + dj, i + di);
  }
  ADEV double dist(int k1, int j1, int i1, int k2, int j2, int i2) const {
    if constexpr (CURV) {
      if (xc0) {
        const long c1 = (static_cast<long>(k1) * P.nj + j1) * P.ni + i1, c2 = (static_cast<long>(k2) * P.nj + j2) * P.ni + i2;
        return sqrt(sqr(xc0[c1] - xc0[c2]) + sqr(xc1[c1] - xc1[c2]) + sqr(xc2[c1] - xc2[c2]));
      }
      double a[3], c[3];
      make_coords(P, b, k1, j1, i1).centre_to_cart(a);
      make_coords(P, b, k2, j2, i2).centre_to_cart(c);
      return sqrt(sqr(a[0] - c[0]) + sqr(a[1] - c[1]) + sqr(a[2] - c[2]));
    } else {
      return sqrt(sqr(x1v(i1) - x1v(i2)) + sqr(x2v(j1) - x2v(j2)) + sqr(x3v(k1) - x3v(k2)));
    }
  }
  ADEV void hx(int k, int j, int i, double h[3]) const { scale_factors<CURV>(P, b, k, j, i, h); }
  ADEV DCoords coords(int k, int j, int i) const { return make_coords(P, b, k, j, i); }
  // {dh2dx1, dh3dx1, dh3dx2}: the only non-zero connection coefficients (geometry.hpp:236-246)
  ADEV void conn(int k, int j, int i, double &d21, double &d31, double &d32) const {
    d21 = d31 = d32 = 0.0;
    if constexpr (CURV) {
      const DCoords co = make_coords(P, b, k, j, i);
      d21 = co.dh2dx1(), d31 = co.dh3dx1(), d32 = co.dh3dx2();
    }
  }
};

ADEV double face_average(int avg, double mu1, double mu2) { // diffusion_coeff.hpp:139-150
  return (avg == 0) * (0.5 * (mu1 + mu2)) + (avg == 1) * (2.0 * mu1 * mu2 / (mu1 + mu2));
}

__global__ __launch_bounds__(TX *TY) void zero_dflux_kernel(const PackView P, const Box r) {
  BOX_CELL(r)
  const int nv = 4 * P.gas.ns;
  for (int d = 0; d < P.ndim; ++d)
    for (int n = 0; n < nv; ++n) P.gas.dflux[d][b * nv + n][c] = 0.0;
}

// VelocityDivergence (momentum_diffusion.hpp:562-591) of cell (k,j,i), species n: the numerator; the caller
// divides by vol2 = 2 * cell volume
template <bool CURV>
ADEV double velocity_divergence(const PackView &P, double *const *prim, int b, int n, int k, int j, int i, double &vol2) {
  const int ns = P.gas.ns, nv = 6 * ns;
  const int multid = (P.ndim >= 2), threed = (P.ndim == 3);
  const CellMetric m = cell_metric<CURV>(P, b, k, j, i);
  const double a1[2] = {m.ax1[0], m.ax1[1]};
  const double a2[2] = {multid ? m.ax2[0] : 0.0, multid ? m.ax2[1] : 0.0};
  const double a3[2] = {threed ? m.ax3[0] : 0.0, threed ? m.ax3[1] : 0.0};
  const double *v1 = prim[b * nv + ns + 3 * n + 0], *v2 = prim[b * nv + ns + 3 * n + 1];
  const double *v3 = prim[b * nv + ns + 3 * n + 2];
  const long c = (static_cast<long>(k) * P.nj + j) * P.ni + i;
  const long sj = multid * P.sj, sk = threed * P.sk;
  const double divv = a1[1] * (v1[c] + v1[c + 1]) - a1[0] * (v1[c] + v1[c - 1]) +
                      multid * a2[1] * (v2[c] + v2[c + sj]) - multid * a2[0] * (v2[c] + v2[c - sj]) +
                      threed * a3[1] * (v3[c] + v3[c + sk]) - threed * a3[0] * (v3[c] + v3[c - sk]);
  vol2 = 2.0 * m.vol;
  return divv;
}

// Per-cell quantities the face kernels share, written once per stage by viscous_cell_kernel (the
// reference keeps them in scratch rows per pencil, momentum_diffusion.hpp:620-700): the contravariant
// velocities v^d = v_d / h_d, VelocityDivergence and the dynamic viscosity.  Arrays of
// [nb * ns][nk * nj * ni] doubles in a library-owned device buffer.
struct ViscScratch {
  double *sv[3], *divu, *mu;
  double *xc[3]; // [nb][nk * nj * ni] Cartesian images of the cell centres (curvilinear blocks)
};
template <bool CURV>
__global__ __launch_bounds__(TX *TY) void viscous_cell_kernel(const PackView P, const Box r,
                                                              const artemis_diffusion_t D, const ViscScratch w) {
  BOX_CELL(r)
  const FluidView &f = P.gas;
  const int ns = f.ns, nv = 6 * ns;
  const long N = static_cast<long>(P.ni) * P.nj * P.nk;
  double hx[3];
  scale_factors<CURV>(P, b, k, j, i, hx);
  if (CURV && !D.dist) { // geometry.hpp:248 of the cell centre, for Coords::Distance in the face kernels
    double xc[3];
    make_coords(P, b, k, j, i).centre_to_cart(xc);
    for (int d = 0; d < 3; ++d) w.xc[d][static_cast<long>(b) * N + c] = xc[d];
  }
  for (int n = 0; n < ns; ++n) {
    const long q = (static_cast<long>(b) * ns + n) * N + c;
    const double v[3] = {f.prim[b * nv + ns + 3 * n + 0][c], f.prim[b * nv + ns + 3 * n + 1][c], f.prim[b * nv + ns + 3 * n + 2][c]};
    double vol2;
    const double divv = velocity_divergence<CURV>(P, f.prim, b, n, k, j, i, vol2);
    // four quotients by geometry of order one: reciprocal form unless a numerator of this wave is tiny (see QuotF)
    const bool odd = tiny_nonzero(v[0]) || tiny_nonzero(v[1]) || tiny_nonzero(v[2]) || tiny_nonzero(divv);
    if (__any(odd)) {
      for (int d = 0; d < 3; ++d) w.sv[d][q] = v[d] / hx[d];
      w.divu[q] = divv / vol2;
    } else {
      for (int d = 0; d < 3; ++d) w.sv[d][q] = CURV ? div(v[d], hx[d]) : v[d];
      w.divu[q] = div(divv, vol2);
    }
    w.mu[q] = coeff_of(D.visc, D.cv, P.gm1, f.prim[b * nv + n][c], f.prim[b * nv + 5 * ns + n][c], b, c);
  }
}

// MomentumFluxImpl (momentum_diffusion.hpp:597-755): StrainTensorFace<XDIR> (:28-377) and
// StressTensorFaceX? (:379-560) of the lower `dir` face of cell (k,j,i)
// OVERWRITE: store 0.0 + flux instead of adding to the array (ZeroDiffusionFlux folded in: same bits)
//
// The strain rows of the three directions have one shape.  With d = DIR - 1 the face-normal component and
// t the two others in ascending order,
//   flx[d] = (2 dv_d) / dxa + 0.5 (src + src_m)
//   flx[t] = M_t 0.5 (dvt / dx_t + dvt_m / dx_t_m) + ((h_t / h_d)^2 dv_t) / dxa
// where dv_* are differences of the contravariant velocities across the face, dvt / dvt_m the differences of the
// normal velocity across the two cells along t, and dx_* the Coords::Distance between the centres involved.
// Nine quotients per face: QuotF divides with one refined reciprocal per denominator (device_math.hpp: the
// bits of an IEEE division for numerators that are zero or not tiny, which the caller checks per wave); QuotI is
// the plain division.
struct QuotF {
  Recip r;
  ADEV explicit QuotF(double b) : r(recip(b)) {}
  ADEV double operator()(double a) const { return div(a, r); }
};
struct QuotI {
  double b;
  ADEV explicit QuotI(double b_) : b(b_) {}
  ADEV double operator()(double a) const { return a / b; }
};
struct StrainIn {
  double n_a[3];        // numerators over dxa: 2 dv_d and (h_t / h_d)^2 dv_t, by component
  double n_t[2], n_tm[2]; // dvt, dvt_m of the two transverse components
  double half_src;      // 0.5 (src + src_m)
  double mfac[2];       // M_t * 0.5
  double mu1, mu2;
};
template <int DIR, class Q>
ADEV void strain_rows(const StrainIn &in, double dxa, const double dxt[2], const double dxtm[2], int avg, double flx[3],
                      double &mus) {
  constexpr int d = DIR - 1, t0 = (DIR == 1) ? 1 : 0, t1 = (DIR == 3) ? 1 : 2;
  const Q qa(dxa), qb(dxt[0]), qbm(dxtm[0]), qc(dxt[1]), qcm(dxtm[1]);
  flx[d] = qa(in.n_a[d]) + in.half_src;
  flx[t0] = in.mfac[0] * (qb(in.n_t[0]) + qbm(in.n_tm[0])) + qa(in.n_a[t0]);
  flx[t1] = in.mfac[1] * (qc(in.n_t[1]) + qcm(in.n_tm[1])) + qa(in.n_a[t1]);
  const Q qm(in.mu1 + in.mu2); // face_average (diffusion_coeff.hpp:139-150)
  mus = (avg == 0) * (0.5 * (in.mu1 + in.mu2)) + (avg == 1) * qm(2.0 * in.mu1 * in.mu2);
}
// Geometry of one face as the stress rows use it (shared by the species): scale factors at the face centroid, the
// connection coefficients of the two cells that share the face, the Coords::Distance values, (h_t / h_d)^2.
struct FaceGeo {
  double hxf[3];
  double dh0, dh1, dh0_m, dh1_m;
  double dxa, dxt[2], dxtm[2];
  double ratio[2];
  double mfac[2];
};
// ... its metric part (everything but the distances)
template <int DIR, bool CURV, class GEO>
ADEV FaceGeo face_metric(const PackView &P, const GEO &ge, const int b, const int k, const int j, const int i) {
  const int multid = (P.ndim >= 2), threed = (P.ndim == 3);
  constexpr int dk = (DIR == 3), dj = (DIR == 2), di = (DIR == 1);
  constexpr int d = DIR - 1, t0 = (DIR == 1) ? 1 : 0, t1 = (DIR == 3) ? 1 : 2;
  FaceGeo f;
  f.hxf[0] = f.hxf[1] = f.hxf[2] = 1.0;
  if constexpr (CURV) ge.coords(k, j, i).face_scale(DIR, f.hxf); // h_d at the face centroid
  // dh_a/dx_k of the two cells sharing the face, a = DIR - 1; only dh2dx1, dh3dx1, dh3dx2 can be non-zero
  f.dh0 = f.dh1 = f.dh0_m = f.dh1_m = 0.0;
  if constexpr (CURV && DIR != 1) {
    double d21, d31, d32;
    ge.conn(k, j, i, d21, d31, d32);
    f.dh0 = (DIR == 2) ? d21 : d31, f.dh1 = (DIR == 3) ? d32 : 0.0;
    ge.conn(k - dk, j - dj, i - di, d21, d31, d32);
    f.dh0_m = (DIR == 2) ? d21 : d31, f.dh1_m = (DIR == 3) ? d32 : 0.0;
  }
  if constexpr (DIR == 1) f.mfac[0] = multid * 0.5, f.mfac[1] = threed * 0.5;
  else if constexpr (DIR == 2) f.mfac[0] = 0.5, f.mfac[1] = threed * 0.5;
  else f.mfac[0] = 0.5, f.mfac[1] = 0.5;
  // (h_t / h_d)^2 at the face centroid: geometry of order one, always the reciprocal form
  f.ratio[0] = f.ratio[1] = 1.0;
  if constexpr (CURV) {
    const QuotF qh(f.hxf[d]);
    f.ratio[0] = sqr(qh(f.hxf[t0])), f.ratio[1] = sqr(qh(f.hxf[t1]));
  }
  f.dxa = f.dxt[0] = f.dxt[1] = f.dxtm[0] = f.dxtm[1] = 1.0; // (the caller's part)
  return f;
}
template <int DIR, bool CURV, class GEO>
ADEV FaceGeo face_geometry(const PackView &P, const GEO &ge, const int b, const int k, const int j, const int i) {
  const int multid = (P.ndim >= 2), threed = (P.ndim == 3);
  const double fuzz = 1e-99; // Fuzz<Real>()
  FaceGeo f = face_metric<DIR, CURV>(P, ge, b, k, j, i);
  // Coords::Distance between cell centres: geometry only, shared by the species.  dxt / dxtm: across the face's
  // own cell and across its lower neighbour, along the two transverse directions
  f.dxa = ge.template dist_lower<DIR>(k, j, i);
  if constexpr (DIR == 1) {
    f.dxt[0] = multid ? ge.template dist_across<2>(k, j, i) : fuzz;
    f.dxtm[0] = multid ? ge.template dist_across<2>(k, j, i - 1) : fuzz;
    f.dxt[1] = threed ? ge.template dist_across<3>(k, j, i) : fuzz;
    f.dxtm[1] = threed ? ge.template dist_across<3>(k, j, i - 1) : fuzz;
  } else if constexpr (DIR == 2) {
    f.dxt[0] = ge.template dist_across<1>(k, j, i);
    f.dxtm[0] = ge.template dist_across<1>(k, j - 1, i);
    f.dxt[1] = threed ? ge.template dist_across<3>(k, j, i) : fuzz;
    f.dxtm[1] = threed ? ge.template dist_across<3>(k, j - 1, i) : fuzz;
  } else {
    f.dxt[0] = ge.template dist_across<1>(k, j, i);
    f.dxtm[0] = ge.template dist_across<1>(k - 1, j, i);
    f.dxt[1] = ge.template dist_across<2>(k, j, i);
    f.dxtm[1] = ge.template dist_across<2>(k - 1, j, i);
  }
  return f;
}
// What one species contributes at the face: contravariant velocities of the two cells sharing it (s_c: the cell
// whose lower face it is, s_m: its lower neighbour), the differences of the NORMAL contravariant velocity across
// each of the two cells along the two transverse directions, their dynamic viscosities and velocity divergences.
struct FaceIn {
  double s_c[3], s_m[3];
  double n_t[2], n_tm[2];
  double mu1, mu2, divu, divu_m;
};
// The stress rows of the face: fl[0..2] = momentum fluxes, fe = energy flux (momentum_diffusion.hpp:379-560).
// `valid`: lanes that do not stand for a face (ragged tiles of the marching kernel) must not steer the wave.
template <int DIR>
ADEV void viscous_face_core(const FaceGeo &g, const FaceIn &q, const int avg, const double eta, double fl[3], double &fe,
                            const bool valid = true) {
  constexpr int d = DIR - 1, t0 = (DIR == 1) ? 1 : 0, t1 = (DIR == 3) ? 1 : 2;
  // v^k dh_a/dx_k / h_a of a cell (the third connection row is zero for every system)
  const double src = q.s_c[0] * g.dh0 + q.s_c[1] * g.dh1 + q.s_c[2] * 0.0;
  const double src_m = q.s_m[0] * g.dh0_m + q.s_m[1] * g.dh1_m + q.s_m[2] * 0.0;
  StrainIn in;
  in.n_a[d] = 2 * (q.s_c[d] - q.s_m[d]);
  in.n_a[t0] = g.ratio[0] * (q.s_c[t0] - q.s_m[t0]);
  in.n_a[t1] = g.ratio[1] * (q.s_c[t1] - q.s_m[t1]);
  for (int t = 0; t < 2; ++t) in.n_t[t] = q.n_t[t], in.n_tm[t] = q.n_tm[t];
  in.half_src = 0.5 * (src + src_m);
  in.mfac[0] = g.mfac[0], in.mfac[1] = g.mfac[1];
  in.mu1 = q.mu1, in.mu2 = q.mu2;
  double flx[3], mus;
  const bool odd = valid && (tiny_nonzero(in.n_a[0]) || tiny_nonzero(in.n_a[1]) || tiny_nonzero(in.n_a[2]) || tiny_nonzero(in.n_t[0]) ||
                             tiny_nonzero(in.n_tm[0]) || tiny_nonzero(in.n_t[1]) || tiny_nonzero(in.n_tm[1]) ||
                             tiny_nonzero(in.mu1 * in.mu2) || !(in.mu1 + in.mu2 > 0x1p-200));
  if (__any(odd)) strain_rows<DIR, QuotI>(in, g.dxa, g.dxt, g.dxtm, avg, flx, mus);
  else strain_rows<DIR, QuotF>(in, g.dxa, g.dxt, g.dxtm, avg, flx, mus);
  const double hf = g.hxf[DIR - 1];
  for (int qq = 0; qq < 3; ++qq) fl[qq] = hf * mus * flx[qq];
  fl[DIR - 1] = hf * mus * (flx[DIR - 1] - 1. / 3 * (1. - eta) * (q.divu + q.divu_m));
  fe = 0.5 * (q.s_c[0] + q.s_m[0]) * fl[0] + 0.5 * (q.s_c[1] + q.s_m[1]) * fl[1] + 0.5 * (q.s_c[2] + q.s_m[2]) * fl[2];
}
template <int DIR, bool CURV, bool OVERWRITE>
ADEV void viscous_face(const PackView &P, const artemis_diffusion_t &D, const ViscScratch &w, const int b, const int k,
                       const int j, const int i, const long c) {
  const FluidView &f = P.gas;
  const int ns = f.ns, nq = 4 * ns;
  const int multid = (P.ndim >= 2), threed = (P.ndim == 3);
  const long N = static_cast<long>(P.ni) * P.nj * P.nk;
  Geo<CURV> ge{P, b};
  ge.dtab = D.dist;
  if constexpr (CURV) ge.xc0 = w.xc[0] + b * N, ge.xc1 = w.xc[1] + b * N, ge.xc2 = w.xc[2] + b * N;
  const artemis_diffcoeff_t &dp = D.visc;
  constexpr int d = DIR - 1;
  const long sd = (DIR == 1) ? 1 : ((DIR == 2) ? P.sj : P.sk);
  const long cm = c - sd;
  const FaceGeo fg = face_geometry<DIR, CURV>(P, ge, b, k, j, i);
  long st[2]; // strides of the transverse directions (0 where the direction is not active)
  if constexpr (DIR == 1) st[0] = multid * P.sj, st[1] = threed * P.sk;
  else if constexpr (DIR == 2) st[0] = 1, st[1] = threed * P.sk;
  else st[0] = 1, st[1] = P.sj;
  for (int n = 0; n < ns; ++n) {
    const long base = (static_cast<long>(b) * ns + n) * N;
    const double *sv[3] = {w.sv[0] + base, w.sv[1] + base, w.sv[2] + base};
    const double *sn = sv[d];
    FaceIn q;
    for (int m = 0; m < 3; ++m) q.s_c[m] = sv[m][c], q.s_m[m] = sv[m][cm];
    for (int t = 0; t < 2; ++t) {
      q.n_t[t] = sn[c + st[t]] - sn[c - st[t]];
      q.n_tm[t] = sn[cm + st[t]] - sn[cm - st[t]];
    }
    q.mu1 = w.mu[base + c], q.mu2 = w.mu[base + cm];
    q.divu = w.divu[base + c], q.divu_m = w.divu[base + cm];
    double fl[3], fe;
    viscous_face_core<DIR>(fg, q, dp.avg, dp.eta, fl, fe);
    double *const *qf = f.dflux[DIR - 1];
    if constexpr (OVERWRITE) {
      for (int qq = 0; qq < 3; ++qq) qf[b * nq + 3 * n + qq][c] = 0.0 + fl[qq];
      qf[b * nq + 3 * ns + n][c] = 0.0 + fe;
    } else {
      for (int qq = 0; qq < 3; ++qq) qf[b * nq + 3 * n + qq][c] += fl[qq];
      qf[b * nq + 3 * ns + n][c] += fe;
    }
  }
}
// The three directions in one pass over the cells of the active region grown by one zone at the upper ends: a
// thread computes the lower x1 / x2 / x3 faces its cell owns (the shared per-cell scratch is read once instead
// of three times, nothing is read-modify-written); OVERWRITE also replaces the zeroing pass.
template <bool CURV, bool OVERWRITE>
__global__ __launch_bounds__(TX *TY, 4) void viscous_flux3_kernel(const PackView P, const Box r,
                                                               const artemis_diffusion_t D, const ViscScratch w) {
  // (a k-march form -- one thread per (i, j) column of 8 planes, the (x1, x2)-only metric hoisted out of the plane
  // loop -- was measured at 470 us against 306 us for this one on the 256 x 128^2 spherical disk: 249 registers)
  BOX_CELL(r)
  const bool in1 = i <= P.ie, in2 = j <= P.je || P.ndim < 2, in3 = k <= P.ke || P.ndim < 3;
  if (in2 && in3) viscous_face<1, CURV, OVERWRITE>(P, D, w, b, k, j, i, c);
  if (P.ndim > 1 && in1 && in3) viscous_face<2, CURV, OVERWRITE>(P, D, w, b, k, j, i, c);
  if (P.ndim > 2 && in1 && in2) viscous_face<3, CURV, OVERWRITE>(P, D, w, b, k, j, i, c);
}



// ---- viscous source: ZeroDiffusionFlux + ViscousFlux + the viscous part of DiffusionUpdate as ONE tile march ---------
// The three tasks above move a lot of memory for what they compute: viscous_cell_kernel writes five doubles per zone,
// viscous_flux3_kernel reads them back through ~50 cached loads per thread and stores twelve face fluxes per zone, and
// the stage kernel reads 24 of those per zone to form the five numbers DiffusionUpdate subtracts from the zone's momenta
// and energies (diffusion.hpp:110-241) -- on the spherical disk deck two thirds of the stage's HBM traffic.  This kernel
// produces those five numbers directly: a 256-thread workgroup owns a VTX x VTY column of zones and marches along x3,
//   * the primitive velocities of the arriving plane staged once in LDS (halo 2),
//   * the contravariant velocities v^d = v_d / h_d of three planes in an LDS ring (halo 1, corners included: the
//     cross derivatives of momentum_diffusion.hpp:95-141 read them),
//   * VelocityDivergence and the dynamic viscosity of the tile and of the ring of its face neighbours for the plane
//     the faces are formed on (the own column's values of the planes below and above live in registers),
//   * every face inside the tile computed once (the lower x1 / x2 faces of each zone go to the neighbour through LDS,
//     the x3 face is carried along the march), the faces on the tile's upper perimeter by one wave whose turn
//     rotates with k,
//   * every global load of a trip -- the primitives and the viscosity law's radial factor two planes ahead, the zone's
//     six entries of the distance table one plane ahead (neighbours read them from LDS) -- issued at the top of the
//     trip and consumed after the plane's LDS phases: the first version fetched distances and radial factors where
//     it needed them, seven exposed memory latencies per plane, and ran at a fifth of the instruction rate,
// with the device functions of the three tasks (viscous_face_core, viscosity_of, diffusion_update_core): the same bits
// as ZeroDiffusionFlux -> ViscousFlux -> DiffusionUpdate, without a diffusion-flux array.  HBM per zone: five
// primitives, the radial factor of the viscosity law and the distance table in, five doubles out.
// 3-D blocks, one gas species, viscosity only (conduction adds to the same energy flux and keeps the flux arrays).
struct VsArgs {
  artemis_diffusion_t D;
  double dt;
  const double *dt_ptr;
  double *const *out; // [nblocks * 5]: what DiffusionUpdate subtracts from M1, M2, M3, E and e_int
  int nti, ntj, nchunk, kchunk;
};
template <int VTX>
struct VsTile {
  static constexpr int VTY = 256 / VTX, QX = VTX + 4, QY = VTY + 4, SX = VTX + 2, SY = VTY + 2;
  double V[3][QY][QX];                  // primitive velocities of the plane being staged
  double S[3][3][SY][SX];               // [plane mod 3][component]: contravariant velocities
  double DV[2][SY][SX], MU[2][SY][SX];  // [plane mod 2]: tile + the ring of its face neighbours
  double FX[4][VTY][VTX + 1], FY[4][VTY + 1][VTX]; // lower x1 / x2 faces: 3 momentum fluxes + the energy flux
  double DA[3][VTY][VTX];               // Distance across the zone along x1, x2, x3 (plane k), for the neighbours
};
struct Vel6 {
  double d, v1, v2, v3, e, rad;
};
// What face_metric / face_geometry ask of a block's geometry, answered from the workgroup's LDS tables (geometry.hpp
// GeoTabs: the staged rectangle's columns and rows) instead of per-thread registers (the distances the march reads
// from the host's table itself, a plane ahead)
template <int SYS, int NX, int NY>
struct TileGeo {
  static constexpr bool CURV = (SYS != ARTEMIS_CARTESIAN);
  const PackView &P;
  const int b, ibase, jbase;
  const GeoTabs<NX, NY> &G;
  ADEV int col(int i) const { return min(max(i - ibase, 0), NX - 1); }
  ADEV int row(int j) const { return min(max(j - jbase, 0), NY - 1); }
  ADEV DCoordsT<true> coords(int k, int j, int i) const {
    // SYS is a compile-time constant: every switch on the coordinate system inside Coords folds away
    return geotabs_coords(G, SYS, col(i), row(j), k, 1.0, 0.0); // (nothing here reads c3 / s3)
  }
  ADEV void conn(int, int j, int i, double &d21, double &d31, double &d32) const {
    d21 = d31 = d32 = 0.0;
    if constexpr (CURV) d21 = G.gi[GI_DH2][col(i)], d31 = G.gi[GI_DH3][col(i)], d32 = G.gj[GJ_DH32][row(j)];
  }
};
// The block's five primitive arrays, read from the pack's pointer table ONCE, before the march: inside it the kernel has
// stores in flight, a table entry would come through a vector load (no scalar load next to stores) with a vmcnt(0) wait
// behind it -- which also waits for every prefetch of the trip.
struct Prim5Ptr {
  const double *d, *v1, *v2, *v3, *e;
};
// (radial: the viscosity law's radial factor, or -- no such law -- any array of the block: the caller ignores the value)
// WITH_V3 = false: the march fetches a plane's x3 velocity one trip ahead of the rest of the zone (it closes the
// divergence of the plane below) and puts it into the record itself -- no second load of the same value.  The march
// waits about 200 cycles per vector-memory instruction it issues (32 a trip): every one that can go, goes.
template <bool WITH_V3 = true>
ADEV Vel6 load6(const Prim5Ptr &p, const double *radial, unsigned c) {
  Vel6 q;
  q.d = fused::gld(p.d, c), q.v1 = fused::gld(p.v1, c), q.v2 = fused::gld(p.v2, c);
  if constexpr (WITH_V3) q.v3 = fused::gld(p.v3, c);
  else q.v3 = 0.0;
  q.e = fused::gld(p.e, c);
  q.rad = fused::gld(radial, c);
  return q;
}
#ifndef VS_OCC
#define VS_OCC 2
#endif
// SYS: the coordinate system as a template constant -- the march is instruction-bound, and with a run-time system the
// Coords members compile to a six-way branch each (the first build of this kernel: 14 700 instructions, half of them
// scalar branches); the per-task kernels keep the run-time switch, they wait for memory anyway.
template <int SYS, int VTX>
__global__ __launch_bounds__(256, VS_OCC) void viscous_source_kernel(const PackView P, const VsArgs a) {
  constexpr bool CURV = (SYS != ARTEMIS_CARTESIAN);
  using T = VsTile<VTX>;
  constexpr int VTY = T::VTY, QX = T::QX, QY = T::QY;
  __shared__ T L;
  __shared__ GeoTabs<QX, QY> GT;
  const int t = threadIdx.x, tx = t % VTX, ty = t / VTX;
  int id = blockIdx.x;
  { // workgroup ids are dealt round-robin over the 8 XCDs: give each XCD's L2 one contiguous run of tiles (the halo
    // columns and the 128-byte lines a row segment straddles are then fetched from HBM once, not once per XCD)
    const int n = static_cast<int>(gridDim.x), q = n >> 3, rem = n & 7, xcd = id & 7;
    id = xcd * q + min(xcd, rem) + (id >> 3);
  }
  const int ti = id % a.nti;
  id /= a.nti;
  const int tj = id % a.ntj;
  id /= a.ntj;
  const int chunk = id % a.nchunk, b = id / a.nchunk;
  const int i0 = P.is + ti * VTX, j0 = P.js + tj * VTY;
  const int k0 = P.ks + chunk * a.kchunk, k1 = min(P.ke, k0 + a.kchunk - 1);
  const int i = i0 + tx, j = j0 + ty;
  const bool active = (i <= P.ie) && (j <= P.je);
  const bool facev = (i <= P.ie + 1) && (j <= P.je + 1); // the lane's lower faces exist (ragged tiles)
  const int il = min(i, P.ni - 1), jl = min(j, P.nj - 1);
  const unsigned sj = static_cast<unsigned>(P.sj), sk = static_cast<unsigned>(P.sk);
  const unsigned col = static_cast<unsigned>(jl) * sj + static_cast<unsigned>(il);
  const Prim5Ptr prim{P.gas.prim[b * 6 + 0], P.gas.prim[b * 6 + 1], P.gas.prim[b * 6 + 2], P.gas.prim[b * 6 + 3], P.gas.prim[b * 6 + 5]};
  double *const o_m1 = a.out[b * 5 + 0], *const o_m2 = a.out[b * 5 + 1], *const o_m3 = a.out[b * 5 + 2];
  double *const o_e = a.out[b * 5 + 3], *const o_eg = a.out[b * 5 + 4];
  const artemis_diffcoeff_t &dp = a.D.visc;
  const bool has_radial = dp.radial != nullptr;
  const double *radial = has_radial ? dp.radial[b] : prim.d; // (load6 fetches unconditionally)
  const double dt = a.dt_ptr ? *a.dt_ptr : a.dt;
  const TileGeo<SYS, QX, QY> ge{P, b, i0 - 2, j0 - 2, GT};
  geotabs_fill(GT, P, b, i0 - 2, j0 - 2, t);
  __syncthreads();
  // the distance table: [6][nb][N]; q = 0..2 to the lower neighbour along x1, x2, x3, q = 3..5 across the zone
  const long NN = static_cast<long>(P.ni) * P.nj * P.nk;
  const double *dtab = a.D.dist + static_cast<long>(b) * NN; // (required: launch_viscous_source's caller checks)
  const long dq = static_cast<long>(P.nb) * NN;
  // the halo zone this thread stages (every plane of the march), its place in the staged rectangle and its duties
  // numbering: the ring of the tile's face neighbours first (2 VTX + 2 VTY zones: they also carry a divergence and a
  // viscosity), the four corners of the one-zone frame, then the outer frame -- so the duties end on a wave boundary
  // early (32 x 8: ring in waves 0 and 1; 16 x 16: wave 0)
  constexpr int NH = QX * QY - 256, NRING = 2 * VTX + 2 * VTY, NS1 = NRING + 4;
  int hr = -1, hc = -1;
  if (t < 2 * VTX) hr = (t < VTX) ? 1 : QY - 2, hc = 2 + t % VTX;
  else if (t < NRING) hc = ((t - 2 * VTX) < VTY) ? 1 : QX - 2, hr = 2 + (t - 2 * VTX) % VTY;
  else if (t < NS1) hr = ((t - NRING) & 1) ? QY - 2 : 1, hc = ((t - NRING) & 2) ? QX - 2 : 1;
  else if (t < NH) {
    const int u = t - NS1; // outer frame: rows 0 and QY - 1 (QX zones each), columns 0 and QX - 1 (rows 1 .. QY - 2)
    if (u < 2 * QX) hr = (u < QX) ? 0 : QY - 1, hc = u % QX;
    else hc = ((u - 2 * QX) < QY - 2) ? 0 : QX - 1, hr = 1 + (u - 2 * QX) % (QY - 2);
  }
  const bool h_any = hr >= 0;
  const int gi = min(max(i0 - 2 + hc, 0), P.ni - 1), gj = min(max(j0 - 2 + hr, 0), P.nj - 1);
  const unsigned hcol = h_any ? static_cast<unsigned>(gj) * sj + static_cast<unsigned>(gi) : col;
  const bool h_s = h_any && hr >= 1 && hr <= QY - 2 && hc >= 1 && hc <= QX - 2; // within one zone of the tile
  const bool h_ring = h_any && (((hr == 1 || hr == QY - 2) && hc >= 2 && hc <= QX - 3) ||
                                ((hc == 1 || hc == QX - 2) && hr >= 2 && hr <= QY - 3)); // a face neighbour of the tile
  auto contravariant = [&](const double v[3], int jj, int ii, double s[3]) { // viscous_cell_kernel's quotients
    if constexpr (!CURV) {
      s[0] = v[0], s[1] = v[1], s[2] = v[2];
    } else {
      double h[3]; // (rebuilt from the tables every plane: six doubles less to carry through the march)
      scale_factors_of(ge.coords(k0, jj, ii), h);
      const bool odd = tiny_nonzero(v[0]) || tiny_nonzero(v[1]) || tiny_nonzero(v[2]);
      if (__any(odd)) {
        for (int d = 0; d < 3; ++d) s[d] = v[d] / h[d];
      } else {
        for (int d = 0; d < 3; ++d) s[d] = div(v[d], h[d]);
      }
    }
  };
  // VelocityDivergence (momentum_diffusion.hpp:562-591) of the zone at (row, column) of the staged rectangle, plane kk
  auto divergence = [&](int kk, int gjj, int gii, int qr, int qc, double v3m, double v3c, double v3p) {
    const CellMetric m = cell_metric_of(ge.coords(kk, gjj, gii));
    const double divv = m.ax1[1] * (L.V[0][qr][qc] + L.V[0][qr][qc + 1]) - m.ax1[0] * (L.V[0][qr][qc] + L.V[0][qr][qc - 1]) +
                        1 * m.ax2[1] * (L.V[1][qr][qc] + L.V[1][qr + 1][qc]) - 1 * m.ax2[0] * (L.V[1][qr][qc] + L.V[1][qr - 1][qc]) +
                        1 * m.ax3[1] * (v3c + v3p) - 1 * m.ax3[0] * (v3c + v3m);
    const double vol2 = 2.0 * m.vol;
    return __any(tiny_nonzero(divv)) ? divv / vol2 : div(divv, vol2);
  };
  auto viscosity = [&](const Vel6 &q) { // (alpha law: one quotient by the radial factor; tiny numerators take `/`)
    const bool odd = dp.type == ARTEMIS_VISCOSITY_ALPHA && tiny_nonzero(q.d * q.e);
    return viscosity_of(dp, P.gm1, q.d, q.e, has_radial ? q.rad : 1.0, !__any(odd));
  };
  // Rolling state.  Nothing a trip loads is consumed in that trip: the own / halo zone's primitives arrive one plane
  // before they are staged (the x3 velocity two planes before: it closes the divergence of the plane below), the
  // distances one trip before the faces that use them.
  const int kmax = P.nk - 1;
  auto plane = [&](int kk) { return static_cast<unsigned>(min(max(kk, 0), kmax)) * sk; };
  Vel6 rn = load6(prim, radial, col + plane(k0 - 1)), hn = rn;
  double vc[3] = {0.0, 0.0, fused::gld(prim.v3, col + plane(k0 - 2))}, h3c = 0.0;
  double v3n2 = fused::gld(prim.v3, col + plane(k0)), h3n2 = 0.0; // x3 velocity of plane k + 2
  if (h_any) {
    hn = load6(prim, radial, hcol + plane(k0 - 1));
    h3c = fused::gld(prim.v3, hcol + plane(k0 - 2)), h3n2 = fused::gld(prim.v3, hcol + plane(k0));
  }
  double dv_c = 0.0, mu_c = 0.0;
  double f3lo[4] = {0.0, 0.0, 0.0, 0.0};
  // the own zone's distances: plane k (to the lower x1 / x2 neighbour, across the zone along x1, x2, x3) and what the x3
  // face above it needs of plane k + 1 (to the lower x3 neighbour, across along x1 and x2)
  double dl1 = 1.0, dl2 = 1.0, da1 = 1.0, da2 = 1.0, da3 = 1.0;
  double ul3 = 1.0, ua1 = 1.0, ua2 = 1.0;
  {
    const unsigned c1 = col + plane(k0 - 1);
    ul3 = fused::gld(dtab + 2 * dq, c1), ua1 = fused::gld(dtab + 3 * dq, c1), ua2 = fused::gld(dtab + 4 * dq, c1);
  }
  double nb0 = 1.0, nb1 = 1.0, nb2 = 1.0, nb3 = 1.0; // across the lower neighbours just outside the tile (edge lanes), plane k
  double dd0 = 1.0, dd1 = 1.0, dd2 = 1.0;            // this trip's perimeter duty: to the lower neighbour, across x2|x1, across x3
  double pend[5] = {0.0, 0.0, 0.0, 0.0, 0.0}; // the sums of the zone finished in the previous trip
  unsigned pend_c = 0;
  bool pend_on = false;
  auto flush = [&]() {
    if (pend_on) {
      fused::gst(o_m1, pend_c, pend[0]), fused::gst(o_m2, pend_c, pend[1]), fused::gst(o_m3, pend_c, pend[2]);
      fused::gst(o_e, pend_c, pend[3]), fused::gst(o_eg, pend_c, pend[4]);
    }
    pend_on = false;
  };
  VPROF_DECL;
  for (int k = k0 - 2; k <= k1; ++k) {
    const bool live = k >= k0; // (wave-uniform) faces of plane k are formed
#ifdef VS_PROF
    __builtin_amdgcn_s_waitcnt(0x0F70); // (profiling builds: the wait for the previous trip's loads gets its own slot)
#endif
    VPROF(0);
    flush();
    // ---- this trip's global loads, all of them, first -------------------------------------------------------------
    // Every load of the own / halo zone is UNCONDITIONAL (threads without a halo zone fetch their own zone again: hcol ==
    // col for them): behind a branch, the compiler's s_waitcnt for an older load has to assume the younger ones were not
    // issued, i.e. it waits for all of them.  And nothing may read a value in the trip that loads it -- not even a copy.
    const Vel6 rnn = load6<false>(prim, radial, col + plane(k + 2)); // (its v3 slot: the value fetched a trip ago)
    const double v3n3 = fused::gld(prim.v3, col + plane(k + 3));
    const Vel6 hnn = load6<false>(prim, radial, hcol + plane(k + 2));
    const double h3n3 = fused::gld(prim.v3, hcol + plane(k + 3));
    const unsigned cn1 = col + static_cast<unsigned>(k + 1) * sk, cn2 = col + plane(k + 2);
    double nl1 = 1.0, nl2 = 1.0, na3 = 1.0;  // plane k + 1: for the next trip's x1 / x2 faces
    double wl3 = 1.0, wa1 = 1.0, wa2 = 1.0;  // plane k + 2: for the next trip's x3 face
    // ... and what the NEXT trip's faces need from outside the tile, so that no load is waited for in the trip that
    // issues it: across the lower x1 (tx == 0) / x2 (ty == 0) neighbour, and the perimeter duty's own three distances
    double nnb0 = 1.0, nnb1 = 1.0, nnb2 = 1.0, nnb3 = 1.0, ndd0 = 1.0, ndd1 = 1.0, ndd2 = 1.0;
    const int duty = (t + 64 * (k & 3)) & 255, dutyn = (t + 64 * ((k + 1) & 3)) & 255;
    const bool duty1 = live && duty < VTY, duty2 = live && duty >= 64 && duty < 64 + VTX;
    const int dj1 = min(j0 + duty, P.nj - 1), di1 = min(i0 + VTX, P.ni - 1);        // the x1 face of zone (j0 + duty, i0 + VTX)
    const int dj2 = min(j0 + VTY, P.nj - 1), di2 = min(i0 + (duty - 64), P.ni - 1); // the x2 face of zone (j0 + VTY, i0 + duty - 64)
    {
      nl1 = fused::gld(dtab, cn1), nl2 = fused::gld(dtab + dq, cn1), na3 = fused::gld(dtab + 5 * dq, cn1);
      wl3 = fused::gld(dtab + 2 * dq, cn2), wa1 = fused::gld(dtab + 3 * dq, cn2), wa2 = fused::gld(dtab + 4 * dq, cn2);
      if (k + 1 >= k0 && k < k1) { // (trip k + 1 forms faces)
        if (tx == 0) nnb0 = fused::gld(dtab + 4 * dq, cn1 - 1), nnb1 = fused::gld(dtab + 5 * dq, cn1 - 1);
        if (ty == 0) nnb2 = fused::gld(dtab + 3 * dq, cn1 - sj), nnb3 = fused::gld(dtab + 5 * dq, cn1 - sj);
        if (dutyn < VTY) {
          const unsigned cd = static_cast<unsigned>(((k + 1) * P.nj + min(j0 + dutyn, P.nj - 1)) * P.ni + di1);
          ndd0 = fused::gld(dtab, cd), ndd1 = fused::gld(dtab + 4 * dq, cd), ndd2 = fused::gld(dtab + 5 * dq, cd);
        } else if (dutyn >= 64 && dutyn < 64 + VTX) {
          const unsigned cd = static_cast<unsigned>(((k + 1) * P.nj + dj2) * P.ni + min(i0 + (dutyn - 64), P.ni - 1));
          ndd0 = fused::gld(dtab + dq, cd), ndd1 = fused::gld(dtab + 3 * dq, cd), ndd2 = fused::gld(dtab + 5 * dq, cd);
        }
      }
    }
    const int pn = (k + 1 + 3) % 3, pc = (k + 3) % 3, pm = (k + 2) % 3; // ring slots of planes k + 1, k, k - 1
    const int dn = (k + 1) & 1, dc = k & 1;
    VPROF(1);
    // ---- (a) stage plane k + 1: primitive and contravariant velocities; the zone's distances of plane k ------------
    {
      const double v[3] = {rn.v1, rn.v2, rn.v3};
      double s[3];
      contravariant(v, jl, il, s);
      for (int m = 0; m < 3; ++m) L.V[m][ty + 2][tx + 2] = v[m], L.S[pn][m][ty + 1][tx + 1] = s[m];
      L.DA[0][ty][tx] = da1, L.DA[1][ty][tx] = da2, L.DA[2][ty][tx] = da3;
      if (h_any) {
        const double w[3] = {hn.v1, hn.v2, hn.v3};
        for (int m = 0; m < 3; ++m) L.V[m][hr][hc] = w[m];
      }
      if (t < ((NS1 + 63) & ~63)) { // (whole waves: the division choice is wave-uniform)
        const double w[3] = {hn.v1, hn.v2, hn.v3};
        double sh[3];
        contravariant(w, gj, gi, sh);
        if (h_s)
          for (int m = 0; m < 3; ++m) L.S[pn][m][hr - 1][hc - 1] = sh[m];
      }
    }
    VPROF(2);
    __syncthreads();
    VPROF(3);
    // ---- (c) faces of plane k ----------------------------------------------------------------------------------------
    // the stress rows of the lower DIR face of the zone at (row sy, column sx) of the S rectangle, block indices (k, jj, ii);
    // d5 = {to the lower neighbour, across the zone / across the lower neighbour along the first transverse direction,
    // the same along x3}
    auto face12 = [&](auto DIRTAG, int sy, int sx, int jj, int ii, bool valid, const double d5[5], double fl[3], double &fe) {
      constexpr int DIR = decltype(DIRTAG)::value;
      constexpr int dy = (DIR == 2), dx = (DIR == 1), nc = DIR - 1;
      FaceGeo fg = face_metric<DIR, CURV>(P, ge, b, k, jj, ii);
      fg.dxa = d5[0], fg.dxt[0] = d5[1], fg.dxtm[0] = d5[2], fg.dxt[1] = d5[3], fg.dxtm[1] = d5[4];
      FaceIn q;
      for (int m = 0; m < 3; ++m) q.s_c[m] = L.S[pc][m][sy][sx], q.s_m[m] = L.S[pc][m][sy - dy][sx - dx];
      // transverse directions: (x2, x3) for an x1 face, (x1, x3) for an x2 face
      q.n_t[0] = L.S[pc][nc][sy + dx][sx + dy] - L.S[pc][nc][sy - dx][sx - dy];
      q.n_tm[0] = L.S[pc][nc][sy - dy + dx][sx - dx + dy] - L.S[pc][nc][sy - dy - dx][sx - dx - dy];
      q.n_t[1] = L.S[pn][nc][sy][sx] - L.S[pm][nc][sy][sx];
      q.n_tm[1] = L.S[pn][nc][sy - dy][sx - dx] - L.S[pm][nc][sy - dy][sx - dx];
      q.mu1 = L.MU[dc][sy][sx], q.mu2 = L.MU[dc][sy - dy][sx - dx];
      q.divu = L.DV[dc][sy][sx], q.divu_m = L.DV[dc][sy - dy][sx - dx];
      viscous_face_core<DIR>(fg, q, dp.avg, dp.eta, fl, fe, valid);
    };
    if (live) {
      {
        const double d5[5] = {dl1, da2, (tx == 0) ? nb0 : L.DA[1][ty][max(tx - 1, 0)], da3, (tx == 0) ? nb1 : L.DA[2][ty][max(tx - 1, 0)]};
        double fl[4];
        face12(std::integral_constant<int, 1>{}, ty + 1, tx + 1, jl, il, facev, d5, fl, fl[3]);
        for (int m = 0; m < 4; ++m) L.FX[m][ty][tx] = fl[m];
      }
      {
        const double d5[5] = {dl2, da1, (ty == 0) ? nb2 : L.DA[0][max(ty - 1, 0)][tx], da3, (ty == 0) ? nb3 : L.DA[2][max(ty - 1, 0)][tx]};
        double fl[4];
        face12(std::integral_constant<int, 2>{}, ty + 1, tx + 1, jl, il, facev, d5, fl, fl[3]);
        for (int m = 0; m < 4; ++m) L.FY[m][ty][tx] = fl[m];
      }
      // the tile's upper perimeter: the x1 faces of column i0 + VTX on one wave, the x2 faces of row j0 + VTY on the
      // next; the turn rotates with k
      if (duty1) {
        const double d5[5] = {dd0, dd1, L.DA[1][duty][VTX - 1], dd2, L.DA[2][duty][VTX - 1]};
        double fl[4];
        face12(std::integral_constant<int, 1>{}, duty + 1, VTX + 1, dj1, di1, (j0 + duty <= P.je) && (i0 + VTX <= P.ie + 1), d5, fl, fl[3]);
        for (int m = 0; m < 4; ++m) L.FX[m][duty][VTX] = fl[m];
      } else if (duty2) {
        const int u = duty - 64;
        const double d5[5] = {dd0, dd1, L.DA[0][VTY - 1][u], dd2, L.DA[2][VTY - 1][u]};
        double fl[4];
        face12(std::integral_constant<int, 2>{}, VTY + 1, u + 1, dj2, di2, (i0 + u <= P.ie) && (j0 + VTY <= P.je + 1), d5, fl, fl[3]);
        for (int m = 0; m < 4; ++m) L.FY[m][VTY][u] = fl[m];
      }
    }
    VPROF(4);
    // ---- (b) VelocityDivergence and viscosity of plane k + 1: own zone, ring zones ------------------------------
    const double dv_n = divergence(k + 1, jl, il, ty + 2, tx + 2, vc[2], rn.v3, v3n2);
    const double mu_n = viscosity(rn);
    L.DV[dn][ty + 1][tx + 1] = dv_n, L.MU[dn][ty + 1][tx + 1] = mu_n;
    if (t < ((NRING + 63) & ~63)) {
      const int qr = h_ring ? hr : 2, qc = h_ring ? hc : 2;
      const double dvh = divergence(k + 1, gj, gi, qr, qc, h3c, hn.v3, h3n2);
      const double muh = viscosity(hn);
      if (h_ring) L.DV[dn][hr - 1][hc - 1] = dvh, L.MU[dn][hr - 1][hc - 1] = muh;
    }
    VPROF(5);
    // ---- the x3 face between planes k and k + 1 (the lower face of zone k + 1): registers + the ring ----------------
    double f3hi[4] = {0.0, 0.0, 0.0, 0.0};
    if (k >= k0 - 1) {
      FaceGeo fg = face_metric<3, CURV>(P, ge, b, k + 1, jl, il);
      fg.dxa = ul3, fg.dxt[0] = ua1, fg.dxtm[0] = da1, fg.dxt[1] = ua2, fg.dxtm[1] = da2;
      FaceIn q;
      const int sy = ty + 1, sx = tx + 1;
      for (int m = 0; m < 3; ++m) q.s_c[m] = L.S[pn][m][sy][sx], q.s_m[m] = L.S[pc][m][sy][sx];
      q.n_t[0] = L.S[pn][2][sy][sx + 1] - L.S[pn][2][sy][sx - 1];
      q.n_tm[0] = L.S[pc][2][sy][sx + 1] - L.S[pc][2][sy][sx - 1];
      q.n_t[1] = L.S[pn][2][sy + 1][sx] - L.S[pn][2][sy - 1][sx];
      q.n_tm[1] = L.S[pc][2][sy + 1][sx] - L.S[pc][2][sy - 1][sx];
      q.mu1 = mu_n, q.mu2 = mu_c, q.divu = dv_n, q.divu_m = dv_c;
      viscous_face_core<3>(fg, q, dp.avg, dp.eta, f3hi, f3hi[3], active);
    }
    VPROF(6);
    __syncthreads();
    VPROF(7);
    // ---- (d) DiffusionUpdate's sums of zone k (diffusion.hpp:110-241) -----------------------------------------------
    if (live && active) {
      double F[3][2][4];
      for (int m = 0; m < 4; ++m) {
        F[0][0][m] = 0.0 + L.FX[m][ty][tx], F[0][1][m] = 0.0 + L.FX[m][ty][tx + 1];
        F[1][0][m] = 0.0 + L.FY[m][ty][tx], F[1][1][m] = 0.0 + L.FY[m][ty + 1][tx];
        F[2][0][m] = 0.0 + f3lo[m], F[2][1][m] = 0.0 + f3hi[m];
      }
      const auto co = ge.coords(k, j, i);
      double hx[3];
      scale_factors_of(co, hx);
      const DiffCell g = diffusion_cell_of(co, cell_metric_of(co), hx, 3);
      double dm[3], de, deg;
      auto FF = [&](int d, int var, int u) { return F[d][u][var]; };
      diffusion_update_core<true>(g, FF, 0, 1, 1, dt, vc, dm, de, deg);
      // (stored at the top of the next trip, in front of its loads: as the youngest memory operations of this trip the
      // stores would be waited for by the next trip's first read of a prefetched value)
      pend[0] = dm[0], pend[1] = dm[1], pend[2] = dm[2], pend[3] = de, pend[4] = deg;
      pend_c = col + static_cast<unsigned>(k) * sk, pend_on = true;
    }
    VPROF(8);
    vc[0] = rn.v1, vc[1] = rn.v2, vc[2] = rn.v3;
    rn = rnn, rn.v3 = v3n2, v3n2 = v3n3;
    h3c = hn.v3, hn = hnn, hn.v3 = h3n2, h3n2 = h3n3;
    dv_c = dv_n, mu_c = mu_n;
    for (int m = 0; m < 4; ++m) f3lo[m] = f3hi[m];
    dl1 = nl1, dl2 = nl2, da1 = ua1, da2 = ua2, da3 = na3;
    ul3 = wl3, ua1 = wa1, ua2 = wa2;
    nb0 = nnb0, nb1 = nnb1, nb2 = nnb2, nb3 = nnb3, dd0 = ndd0, dd1 = ndd1, dd2 = ndd2;
  }
  flush();
#ifdef VS_PROF
  VPROF(9);
  if ((t & 63) == 0)
    for (int q = 0; q < 10; ++q) atomicAdd(&g_vs_prof[q], prof_acc[q]);
#endif
}

// ---- viscous fluxes of LISTED faces (refined meshes on the one-kernel stages) ------------------------------------
// With artemis_hip_viscous_source no diffusion-flux array is filled for the pack, but the flux correction of a refined
// mesh needs the viscous fluxes themselves on two small sets of faces: the fine side of every coarse-fine boundary (to
// restrict) and the six faces of every coarse zone the fix-up redoes.  These kernels evaluate exactly those, each face
// from the primitives around it: the per-zone quantities viscous_cell_kernel would have stored (v / h,
// VelocityDivergence / (2 V), the dynamic viscosity) formed on the spot with the same expressions, then
// viscous_face_core -- the bits ZeroDiffusionFlux + ViscousFlux leave in those entries.  One gas species.
template <int DIR, bool CURV>
ADEV void viscous_face_direct(const PackView &P, const artemis_diffusion_t &D, const int b, const int k, const int j, const int i) {
  const FluidView &f = P.gas;
  const int multid = (P.ndim >= 2), threed = (P.ndim == 3);
  constexpr int dk = (DIR == 3), dj = (DIR == 2), di = (DIR == 1), d = DIR - 1;
  Geo<CURV> ge{P, b};
  ge.dtab = D.dist;
  const FaceGeo fg = face_geometry<DIR, CURV>(P, ge, b, k, j, i);
  auto at = [&](int kk, int jj, int ii) { return (static_cast<long>(kk) * P.nj + jj) * P.ni + ii; };
  auto sv = [&](int comp, int kk, int jj, int ii) { // contravariant velocity component of a zone (IEEE quotient)
    double hx[3];
    scale_factors<CURV>(P, b, kk, jj, ii, hx);
    return f.prim[b * 6 + 1 + comp][at(kk, jj, ii)] / hx[comp];
  };
  auto divu = [&](int kk, int jj, int ii) {
    double vol2;
    const double divv = velocity_divergence<CURV>(P, f.prim, b, 0, kk, jj, ii, vol2);
    return divv / vol2;
  };
  auto mu = [&](int kk, int jj, int ii) {
    const long c = at(kk, jj, ii);
    return coeff_of(D.visc, D.cv, P.gm1, f.prim[b * 6 + 0][c], f.prim[b * 6 + 5][c], b, c);
  };
  // the two transverse directions (as viscous_face's strides: inactive directions repeat the zone itself)
  int tk[2], tj[2], ti[2];
  if constexpr (DIR == 1) tk[0] = 0, tj[0] = multid, ti[0] = 0, tk[1] = threed, tj[1] = 0, ti[1] = 0;
  else if constexpr (DIR == 2) tk[0] = 0, tj[0] = 0, ti[0] = 1, tk[1] = threed, tj[1] = 0, ti[1] = 0;
  else tk[0] = 0, tj[0] = 0, ti[0] = 1, tk[1] = 0, tj[1] = 1, ti[1] = 0;
  FaceIn q;
  for (int m = 0; m < 3; ++m) q.s_c[m] = sv(m, k, j, i), q.s_m[m] = sv(m, k - dk, j - dj, i - di);
  for (int t = 0; t < 2; ++t) {
    q.n_t[t] = sv(d, k + tk[t], j + tj[t], i + ti[t]) - sv(d, k - tk[t], j - tj[t], i - ti[t]);
    q.n_tm[t] = sv(d, k - dk + tk[t], j - dj + tj[t], i - di + ti[t]) - sv(d, k - dk - tk[t], j - dj - tj[t], i - di - ti[t]);
  }
  q.mu1 = mu(k, j, i), q.mu2 = mu(k - dk, j - dj, i - di);
  q.divu = divu(k, j, i), q.divu_m = divu(k - dk, j - dj, i - di);
  double fl[3], fe;
  viscous_face_core<DIR>(fg, q, D.visc.avg, D.visc.eta, fl, fe);
  const long c = at(k, j, i);
  double *const *qf = f.dflux[DIR - 1];
  for (int qq = 0; qq < 3; ++qq) qf[b * 4 + qq][c] = 0.0 + fl[qq];
  qf[b * 4 + 3][c] = 0.0 + fe;
}
template <bool CURV>
ADEV void viscous_face_any(const PackView &P, const artemis_diffusion_t &D, int dir, int b, int k, int j, int i) {
  if (dir == 0) viscous_face_direct<1, CURV>(P, D, b, k, j, i);
  else if (dir == 1) viscous_face_direct<2, CURV>(P, D, b, k, j, i);
  else viscous_face_direct<3, CURV>(P, D, b, k, j, i);
}
// one workgroup per face box (the faces of direction bx.dir stored at the zones of the box)
template <bool CURV>
__global__ __launch_bounds__(256) void viscous_box_faces_kernel(const PackView P, const artemis_diffusion_t D,
                                                                const artemis_ml_face_box_t *__restrict__ boxes) {
  const artemis_ml_face_box_t bx = boxes[blockIdx.x];
  const long ncell = static_cast<long>(bx.n[0]) * bx.n[1] * bx.n[2];
  for (long t = threadIdx.x; t < ncell; t += blockDim.x) {
    const int i = bx.lo[0] + static_cast<int>(t % bx.n[0]), j = bx.lo[1] + static_cast<int>((t / bx.n[0]) % bx.n[1]);
    const int k = bx.lo[2] + static_cast<int>(t / (static_cast<long>(bx.n[0]) * bx.n[1]));
    viscous_face_any<CURV>(P, D, bx.dir, bx.block, k, j, i);
  }
}
// the 2 ndim faces of every listed zone: thread (zone, face slot 2 d + side)
template <bool CURV>
__global__ __launch_bounds__(256) void viscous_cell_faces_kernel(const PackView P, const artemis_diffusion_t D,
                                                                 const artemis_ml_fix_cell_t *__restrict__ cells, int ncells) {
  const long q = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= 6L * ncells) return;
  const artemis_ml_fix_cell_t z = cells[q / 6];
  const int slot = static_cast<int>(q % 6), d = slot >> 1, side = slot & 1;
  if (d >= P.ndim) return;
  viscous_face_any<CURV>(P, D, d, z.block, z.k + ((d == 2) ? side : 0), z.j + ((d == 1) ? side : 0), z.i + ((d == 0) ? side : 0));
}

// artemis_hip_viscous_distance_fill: the six Coords::Distance values (geometry.hpp:407-412) a cell contributes
// to the face kernels, evaluated exactly as Geo::dist does without its cache.  Static geometry: the host fills
// the table once per mesh.  Entries whose neighbours fall outside the block's arrays are left alone (never read).
template <bool CURV>
__global__ __launch_bounds__(TX *TY) void distance_fill_kernel(const PackView P, const Box r, double *tab) {
  BOX_CELL(r)
  const Geo<CURV> ge{P, b};
  const long N = static_cast<long>(P.ni) * P.nj * P.nk;
  const int hi[3] = {P.ni - 1, P.nj - 1, P.nk - 1}, at[3] = {i, j, k};
  for (int dir = 0; dir < P.ndim; ++dir) {
    const int di = (dir == 0), dj = (dir == 1), dk = (dir == 2);
    if (at[dir] >= 1) tab[(static_cast<long>(dir) * P.nb + b) * N + c] = ge.dist(k, j, i, k - dk, j - dj, i - di);
    if (at[dir] >= 1 && at[dir] < hi[dir])
      tab[(static_cast<long>(3 + dir) * P.nb + b) * N + c] = ge.dist(k - dk, j - dj, i - di, k + dk, j + dj, i + di);
  }
}

// ThermalFluxImpl (thermal_diffusion.hpp:30-222)
template <int DIR, bool CURV>
__global__ __launch_bounds__(TX *TY) void thermal_flux_kernel(const PackView P, const Box r,
                                                              const artemis_diffusion_t D) {
  BOX_CELL(r)
  const FluidView &f = P.gas;
  const int ns = f.ns, nv = 6 * ns, nq = 4 * ns;
  Geo<CURV> ge{P, b};
  ge.dtab = D.dist;
  const artemis_diffcoeff_t &dp = D.cond;
  const long cm = c - ((DIR == 1) ? 1 : ((DIR == 2) ? P.sj : P.sk));
  const double dx = ge.template dist_lower<DIR>(k, j, i);
  for (int n = 0; n < ns; ++n) {
    const double *rho = f.prim[b * nv + n], *se = f.prim[b * nv + 5 * ns + n];
    const double T = amax(0.0, se[c] / D.cv);   // IdealGas TemperatureFromDensityInternalEnergy
    const double Tm = amax(0.0, se[cm] / D.cv);
    const double kcond = face_average(dp.avg, coeff_of(dp, D.cv, P.gm1, rho[c], se[c], b, c),
                                      coeff_of(dp, D.cv, P.gm1, rho[cm], se[cm], b, cm));
    f.dflux[DIR - 1][b * nq + 3 * ns + n][c] += kcond * (T - Tm) / dx;
  }
}

// DiffusionUpdateImpl (diffusion.hpp:110-241); the cell body lives in diffusion_device.hpp, shared
// with the general fused stage
template <bool CURV>
__global__ __launch_bounds__(TX *TY) void diffusion_update_kernel(const PackView P, const Box r,
                                                                  int do_viscosity, double dt) {
  BOX_CELL(r)
  const FluidView &f = P.gas;
  const int ns = f.ns, nv = 6 * ns;
  const DiffCell dc = diffusion_cell<CURV>(P, b, k, j, i);
  for (int n = 0; n < ns; ++n) {
    const double v[3] = {f.prim[b * nv + ns + 3 * n + 0][c], f.prim[b * nv + ns + 3 * n + 1][c],
                         f.prim[b * nv + ns + 3 * n + 2][c]};
    double dm[3], de, deg;
    diffusion_update_cell(P, dc, b, n, c, do_viscosity, dt, v, dm, de, deg);
    f.cons0[b * nv + ns + 3 * n + 0][c] -= dm[0];
    f.cons0[b * nv + ns + 3 * n + 1][c] -= dm[1];
    f.cons0[b * nv + ns + 3 * n + 2][c] -= dm[2];
    f.cons0[b * nv + 4 * ns + n][c] -= de;
    f.cons0[b * nv + 5 * ns + n][c] -= deg;
  }
}

// Diffusion::EstimateTimestep (diffusion.hpp:66-108) for one coefficient; grid-stride reduction
template <bool CURV>
__global__ __launch_bounds__(TX *TY) void diffusion_dt_kernel(const PackView P, const Box r,
                                                              const artemis_diffcoeff_t dp, double cv,
                                                              double cfl, unsigned long long *dt_bits) {
  const int gx = (r.iu - r.il + TX) / TX, gy = (r.ju - r.jl + TY) / TY;
  const int nkr = r.ku - r.kl + 1;
  const long ntile = static_cast<long>(gx) * gy * nkr * P.nb;
  double ldt = DBL_MAX;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int i = r.il + static_cast<int>(tile % gx) * TX + threadIdx.x;
    const int j = r.jl + static_cast<int>((tile / gx) % gy) * TY + threadIdx.y;
    const int bz = static_cast<int>(tile / (static_cast<long>(gx) * gy));
    const int b = bz / nkr, k = r.kl + bz % nkr;
    if (i > r.iu || j > r.ju) continue;
    const long c = (static_cast<long>(k) * P.nj + j) * P.ni + i;
    double dx[3]; // GetCellWidths (geometry.hpp:352-361)
    if constexpr (CURV) {
      const DCoords co = make_coords(P, b, k, j, i);
      dx[0] = co.width1(), dx[1] = co.width2(), dx[2] = co.width3();
    } else {
      const CellGeom g = cell_geom(P.geom + 6 * b, k, j, i);
      dx[0] = 1.0 * g.dx1, dx[1] = 1.0 * g.dx2, dx[2] = 1.0 * g.dx3;
    }
    double min_dx = DBL_MAX;
    for (int d = 0; d < P.ndim; d++) min_dx = amin(min_dx, dx[d]);
    const int ns = P.gas.ns, nv = 6 * ns;
    for (int n = 0; n < ns; ++n) {
      const double dens = P.gas.prim[b * nv + n][c];
      double mu = coeff_of(dp, cv, P.gm1, dens, P.gas.prim[b * nv + 5 * ns + n][c], b, c);
      if (dp.type == ARTEMIS_CONDUCTIVITY_PLAW) mu /= (dens * cv);
      else if (dp.type == ARTEMIS_VISCOSITY_PLAW || dp.type == ARTEMIS_VISCOSITY_ALPHA) mu *= (1.0 + (dp.eta > 1.0) * (dp.eta - 1.0)) / dens;
      ldt = amin(ldt, sqr(min_dx) / (mu + 1e-99));
    }
  }
  for (int off = 32; off > 0; off >>= 1) ldt = fmin(ldt, __shfl_down(ldt, off, 64));
  __shared__ double wmin[TY];
  if (threadIdx.x == 0) wmin[threadIdx.y] = ldt;
  __syncthreads();
  if (threadIdx.y == 0 && threadIdx.x == 0) {
    double m = wmin[0];
    for (int w = 1; w < TY; ++w) m = fmin(m, wmin[w]);
    atomicMin(dt_bits, static_cast<unsigned long long>(__double_as_longlong(cfl * (m / (2.0 * P.ndim)))));
  }
}

// Gas::EstimateTimestepMesh (gas.cpp:411-467: hydrodynamic, viscous and conductive limits) and
// Dust::EstimateTimestepMesh (dust.cpp:256-272) of one state in ONE pass: every limit is a minimum over the zones and
// min is exact, so combining them in one kernel gives the bits of the separate tasks with one read of the primitives.
template <bool CURV>
__global__ __launch_bounds__(TX *TY) void timestep_all_kernel(const PackView P, const Box r, const artemis_diffcoeff_t visc,
                                                              const artemis_diffcoeff_t cond, double cv, double cfl_gas,
                                                              double cfl_dust, unsigned long long *dt_bits) {
  // zones in a flat order inside a block (x1 fastest): every lane has a zone whatever the block's width -- a 64-wide
  // thread row on the 16^3 blocks of a refined mesh keeps a quarter of the lanes busy.  A workgroup's 256 zones belong
  // to ONE block: the block index is wave-uniform, so the table pointers and the edge table are scalar loads and a
  // zone's nine values are one level of vector loads (with the block index per lane every array was pointer load ->
  // data load, and the kernel waited 84 % of its cycles: 1.18 ms per pass over the configs[4] mesh)
  const unsigned nx = r.iu - r.il + 1, ny = r.ju - r.jl + 1, nz = r.ku - r.kl + 1;
  const unsigned per_block = nx * ny * nz, tpb = (per_block + TX * TY - 1) / (TX * TY);
  const unsigned long ntile = static_cast<unsigned long>(tpb) * P.nb;
  const unsigned tid = threadIdx.y * TX + threadIdx.x;
  double lg = DBL_MAX, ld = DBL_MAX, lv = DBL_MAX, lc = DBL_MAX; // gas hydro, dust, viscous, conductive
  for (unsigned long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int b = static_cast<int>(tile / tpb);
    const unsigned rr = static_cast<unsigned>(tile - static_cast<unsigned long>(b) * tpb) * (TX * TY) + tid;
    if (rr >= per_block) continue;
    const unsigned row = rr / nx;
    const int i = r.il + static_cast<int>(rr - row * nx), j = r.jl + static_cast<int>(row % ny), k = r.kl + static_cast<int>(row / ny);
    const long c = (static_cast<long>(k) * P.nj + j) * P.ni + i;
    double dx[3]; // GetCellWidths (geometry.hpp:352-361)
    if constexpr (CURV) {
      const DCoords co = make_coords(P, b, k, j, i);
      dx[0] = co.width1(), dx[1] = co.width2(), dx[2] = co.width3();
    } else {
      const CellGeom g = cell_geom(P.geom + 6 * b, k, j, i);
      dx[0] = 1.0 * g.dx1, dx[1] = 1.0 * g.dx2, dx[2] = 1.0 * g.dx3;
    }
    double min_dx = DBL_MAX;
    for (int d = 0; d < P.ndim; d++) min_dx = amin(min_dx, dx[d]);
    {
      const int ns = P.gas.ns, nv = 6 * ns;
      for (int n = 0; n < ns; ++n) {
        const double dens = P.gas.prim[b * nv + n][c];
        const double sie = P.gas.prim[b * nv + 5 * ns + n][c];
        const double bulk = (P.gm1 + 1.0) * P.gm1 * dens * sie; // IdealGas bulk modulus
        const double cs = sqrt(bulk / dens);
        double denom = 0.0;
        for (int d = 0; d < P.ndim; d++) {
          const double ss = fabs(P.gas.prim[b * nv + ns + 3 * n + d][c]) + cs;
          denom += ss / dx[d];
        }
        lg = amin(lg, 1.0 / denom);
        if (visc.type != ARTEMIS_DIFF_OFF) { // Diffusion::EstimateTimestep (diffusion.hpp:66-108)
          double mu = coeff_of(visc, cv, P.gm1, dens, sie, b, c);
          if (visc.type == ARTEMIS_VISCOSITY_PLAW || visc.type == ARTEMIS_VISCOSITY_ALPHA) mu *= (1.0 + (visc.eta > 1.0) * (visc.eta - 1.0)) / dens;
          lv = amin(lv, sqr(min_dx) / (mu + 1e-99));
        }
        if (cond.type != ARTEMIS_DIFF_OFF) {
          double mu = coeff_of(cond, cv, P.gm1, dens, sie, b, c);
          if (cond.type == ARTEMIS_CONDUCTIVITY_PLAW) mu /= (dens * cv);
          lc = amin(lc, sqr(min_dx) / (mu + 1e-99));
        }
      }
    }
    {
      const int ns = P.dust.ns, nv = 4 * ns;
      for (int n = 0; n < ns; ++n) {
        double denom = 0.0;
        for (int d = 0; d < P.ndim; d++) denom += fabs(P.dust.prim[b * nv + ns + 3 * n + d][c]) / dx[d];
        ld = amin(ld, 1.0 / denom);
      }
    }
  }
  // the four candidates as the tasks would min-combine them
  double m = DBL_MAX;
  if (P.gas.ns) {
    m = fmin(m, cfl_gas * lg);
    if (visc.type != ARTEMIS_DIFF_OFF) m = fmin(m, cfl_gas * (lv / (2.0 * P.ndim)));
    if (cond.type != ARTEMIS_DIFF_OFF) m = fmin(m, cfl_gas * (lc / (2.0 * P.ndim)));
  }
  if (P.dust.ns) m = fmin(m, cfl_dust * ld);
  for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_down(m, off, 64));
  __shared__ double wmin[TY];
  if (threadIdx.x == 0) wmin[threadIdx.y] = m;
  __syncthreads();
  if (threadIdx.y == 0 && threadIdx.x == 0) {
    double q = wmin[0];
    for (int w = 1; w < TY; ++w) q = fmin(q, wmin[w]);
    atomicMin(dt_bits, static_cast<unsigned long long>(__double_as_longlong(q)));
  }
}

inline Box interior(const PackView &P) { return Box{P.is, P.ie, P.js, P.je, P.ks, P.ke}; }
inline Box faces(const PackView &P, int dir) {
  Box r = interior(P);
  if (dir == 1) r.iu = P.ie + 1;
  if (dir == 2) r.ju = P.je + 1;
  if (dir == 3) r.ku = P.ke + 1;
  return r;
}
} // namespace

void launch_zero_diffusion_flux(const PackView &P, hipStream_t s) {
  const Box r{0, P.ni - 1, 0, P.nj - 1, 0, P.nk - 1};
  hipLaunchKernelGGL(zero_dflux_kernel, grid_of(r, P.nb), threads_of(r), 0, s, P, r);
}
#define LAUNCH_DIR(kern, DIR)                                                                    \
  do {                                                                                           \
    const Box fr = faces(P, DIR);                                                                \
    if (P.coords == ARTEMIS_CARTESIAN)                                                           \
      hipLaunchKernelGGL((kern<DIR, false>), grid_of(fr, P.nb), threads_of(fr), 0, s, P, fr, D);    \
    else                                                                                         \
      hipLaunchKernelGGL((kern<DIR, true>), grid_of(fr, P.nb), threads_of(fr), 0, s, P, fr, D);     \
  } while (0)
// library-owned scratch of the viscous tasks, grown on demand (one stream per caller thread)
thread_local struct {
  double *p = nullptr;
  size_t n = 0;
} g_visc;
size_t viscous_distance_count(const PackView &P) { return 6 * static_cast<size_t>(P.nb) * P.ni * P.nj * P.nk; }
void launch_viscous_distance_fill(const PackView &P, double *tab, hipStream_t s) {
  const Box r{0, P.ni - 1, 0, P.nj - 1, 0, P.nk - 1};
  if (P.coords != ARTEMIS_CARTESIAN) hipLaunchKernelGGL(distance_fill_kernel<true>, grid_of(r, P.nb), threads_of(r), 0, s, P, r, tab);
  else hipLaunchKernelGGL(distance_fill_kernel<false>, grid_of(r, P.nb), threads_of(r), 0, s, P, r, tab);
}
int launch_viscous_flux(const PackView &P, const artemis_diffusion_t &D, hipStream_t s, bool overwrite) {
  const size_t N = static_cast<size_t>(P.ni) * P.nj * P.nk, per = static_cast<size_t>(P.nb) * P.gas.ns * N;
  const size_t geo = 3 * static_cast<size_t>(P.nb) * N, need = 5 * per + geo;
  if (g_visc.n < need) {
    if (g_visc.p) (void)hipFree(g_visc.p);
    g_visc.p = nullptr, g_visc.n = 0;
    if (hipMalloc(reinterpret_cast<void **>(&g_visc.p), need * sizeof(double)) != hipSuccess) return 1;
    g_visc.n = need;
  }
  ViscScratch w;
  for (int d = 0; d < 3; ++d) w.sv[d] = g_visc.p + d * per;
  w.divu = g_visc.p + 3 * per, w.mu = g_visc.p + 4 * per;
  for (int d = 0; d < 3; ++d) w.xc[d] = g_visc.p + 5 * per + d * static_cast<size_t>(P.nb) * N;
  // cells the face kernels read: the active region grown by one zone in every active direction
  Box rc = interior(P);
  rc.il -= 1, rc.iu += 1;
  if (P.ndim > 1) rc.jl -= 1, rc.ju += 1;
  if (P.ndim > 2) rc.kl -= 1, rc.ku += 1;
  const bool curv = P.coords != ARTEMIS_CARTESIAN;
  if (curv) hipLaunchKernelGGL(viscous_cell_kernel<true>, grid_of(rc, P.nb), threads_of(rc), 0, s, P, rc, D, w);
  else hipLaunchKernelGGL(viscous_cell_kernel<false>, grid_of(rc, P.nb), threads_of(rc), 0, s, P, rc, D, w);
  // one pass for the three directions
  Box fr = interior(P);
  fr.iu = P.ie + 1;
  if (P.ndim > 1) fr.ju = P.je + 1;
  if (P.ndim > 2) fr.ku = P.ke + 1;
  if (curv) {
    if (overwrite) hipLaunchKernelGGL((viscous_flux3_kernel<true, true>), grid_of(fr, P.nb), threads_of(fr), 0, s, P, fr, D, w);
    else hipLaunchKernelGGL((viscous_flux3_kernel<true, false>), grid_of(fr, P.nb), threads_of(fr), 0, s, P, fr, D, w);
  } else {
    if (overwrite) hipLaunchKernelGGL((viscous_flux3_kernel<false, true>), grid_of(fr, P.nb), threads_of(fr), 0, s, P, fr, D, w);
    else hipLaunchKernelGGL((viscous_flux3_kernel<false, false>), grid_of(fr, P.nb), threads_of(fr), 0, s, P, fr, D, w);
  }
  return 0;
}
void launch_viscous_listed_faces(const PackView &P, const artemis_diffusion_t &D, const artemis_ml_face_box_t *boxes, int nboxes,
                                 const artemis_ml_fix_cell_t *cells, int ncells, hipStream_t s) {
  const bool curv = P.coords != ARTEMIS_CARTESIAN;
  if (nboxes > 0) {
    if (curv) hipLaunchKernelGGL(viscous_box_faces_kernel<true>, dim3(nboxes), dim3(256), 0, s, P, D, boxes);
    else hipLaunchKernelGGL(viscous_box_faces_kernel<false>, dim3(nboxes), dim3(256), 0, s, P, D, boxes);
  }
  if (ncells > 0) {
    const dim3 grid(static_cast<unsigned>((6L * ncells + 255) / 256));
    if (curv) hipLaunchKernelGGL(viscous_cell_faces_kernel<true>, grid, dim3(256), 0, s, P, D, cells, ncells);
    else hipLaunchKernelGGL(viscous_cell_faces_kernel<false>, grid, dim3(256), 0, s, P, D, cells, ncells);
  }
}
// Does the viscous-source march cover this pack?  (3-D blocks of one gas species, 32-bit zone offsets.)
bool viscous_source_covers(const PackView &P) {
  if (opt(OPT_NO_VISC_SOURCE)) return false;
  if (P.ndim != 3 || P.gas.ns != 1 || P.ng < 2) return false;
  if (P.coords == ARTEMIS_SPHERICAL1D || P.coords == ARTEMIS_SPHERICAL2D) return false; // (never 3-D blocks)
  if (static_cast<long>(P.nk) * P.nj * P.ni >= (1L << 29)) return false;
  return (P.ie - P.is + 1) >= 8 && (P.je - P.js + 1) >= 8;
}
void launch_viscous_source(const PackView &P, const artemis_diffusion_t &D, double dt, const double *dt_dev, double *const *out,
                           hipStream_t s) {
  VsArgs a;
  a.D = D, a.dt = dt, a.dt_ptr = dt_dev, a.out = out;
  const int nx = P.ie - P.is + 1, ny = P.je - P.js + 1, nz = P.ke - P.ks + 1;
  // tile shape: 32 x 8, or 16 x 16 where a 32-zone row would leave half the lanes without a zone (16-zone blocks)
  const bool narrow = (nx % 32 != 0) && (nx % 16 == 0 || nx < 32);
  const int vtx = narrow ? 16 : 32, vty = 256 / vtx;
  a.nti = (nx + vtx - 1) / vtx, a.ntj = (ny + vty - 1) / vty;
  // chunks along x3: two priming trips each, so long ones -- but enough workgroups for two rounds of the chip's slots
  const long tiles = static_cast<long>(a.nti) * a.ntj * P.nb;
  int kch = 32;
  if (opt(OPT_VISC_KCHUNK) > 0) kch = static_cast<int>(opt(OPT_VISC_KCHUNK));
  else { // full rounds of the chip's 512 slots (two workgroups per CU), two priming trips per chunk: kernels.hpp
    const int n = pick_march_chunks(nz, tiles, 512, 64, 1.0);
    kch = (nz + n - 1) / n;
  }
  a.nchunk = (nz + kch - 1) / kch, a.kchunk = (nz + a.nchunk - 1) / a.nchunk;
  a.nchunk = (nz + a.kchunk - 1) / a.kchunk;
  const dim3 grid(static_cast<unsigned>(tiles * a.nchunk)), block(256);
#define VS_GO(SYS)                                                                                 \
  case SYS:                                                                                       \
    if (narrow) hipLaunchKernelGGL((viscous_source_kernel<SYS, 16>), grid, block, 0, s, P, a);    \
    else hipLaunchKernelGGL((viscous_source_kernel<SYS, 32>), grid, block, 0, s, P, a);           \
    break;
  switch (P.coords) {
    VS_GO(ARTEMIS_CARTESIAN)
    VS_GO(ARTEMIS_CYLINDRICAL)
    VS_GO(ARTEMIS_SPHERICAL3D)
    VS_GO(ARTEMIS_AXISYMMETRIC)
  default: break;
  }
#undef VS_GO
}
void launch_thermal_flux(const PackView &P, const artemis_diffusion_t &D, hipStream_t s) {
  LAUNCH_DIR(thermal_flux_kernel, 1);
  if (P.ndim > 1) LAUNCH_DIR(thermal_flux_kernel, 2);
  if (P.ndim > 2) LAUNCH_DIR(thermal_flux_kernel, 3);
}
void launch_diffusion_update(const PackView &P, const artemis_diffusion_t &D, double dt, hipStream_t s) {
  const Box r = interior(P);
  const int visc = D.visc.type != ARTEMIS_DIFF_OFF ? 1 : 0;
  if (P.coords == ARTEMIS_CARTESIAN)
    hipLaunchKernelGGL(diffusion_update_kernel<false>, grid_of(r, P.nb), threads_of(r), 0, s, P, r, visc, dt);
  else
    hipLaunchKernelGGL(diffusion_update_kernel<true>, grid_of(r, P.nb), threads_of(r), 0, s, P, r, visc, dt);
}
void launch_timestep_all(const PackView &P, const artemis_diffusion_t *D, double cfl_gas, double cfl_dust, double *dt_dev,
                         hipStream_t s) {
  const Box r = interior(P);
  const long per_block = static_cast<long>(r.iu - r.il + 1) * (r.ju - r.jl + 1) * (r.ku - r.kl + 1);
  const long ntile = (per_block + TX * TY - 1) / (TX * TY) * P.nb;
  const dim3 g(static_cast<unsigned>(ntile < 2048 ? std::max<long>(ntile, 1) : 2048));
  auto *bits = reinterpret_cast<unsigned long long *>(dt_dev);
  artemis_diffcoeff_t off;
  std::memset(&off, 0, sizeof off);
  off.type = ARTEMIS_DIFF_OFF;
  const artemis_diffcoeff_t visc = D ? D->visc : off, cond = D ? D->cond : off;
  const double cv = D ? D->cv : 0.0;
  if (P.coords != ARTEMIS_CARTESIAN)
    hipLaunchKernelGGL(timestep_all_kernel<true>, g, dim3(TX, TY), 0, s, P, r, visc, cond, cv, cfl_gas, cfl_dust, bits);
  else
    hipLaunchKernelGGL(timestep_all_kernel<false>, g, dim3(TX, TY), 0, s, P, r, visc, cond, cv, cfl_gas, cfl_dust, bits);
}
void launch_diffusion_dt(const PackView &P, const artemis_diffusion_t &D, double cfl, double *dt_dev,
                         hipStream_t s) {
  const Box r = interior(P);
  const dim3 g3 = grid_of(r, P.nb);
  const long ntile = static_cast<long>(g3.x) * g3.y * g3.z;
  const dim3 g(static_cast<unsigned>(ntile < 1024 ? ntile : 1024)); // (one atomicMin per workgroup on ONE address: 512 / 1024 / 4096 / 16384 workgroups: 77 / 52 / 67 / 194 us)
  auto *bits = reinterpret_cast<unsigned long long *>(dt_dev);
  const bool curv = P.coords != ARTEMIS_CARTESIAN;
  for (const artemis_diffcoeff_t *c : {&D.visc, &D.cond}) {
    if (c->type == ARTEMIS_DIFF_OFF) continue;
    if (curv) hipLaunchKernelGGL(diffusion_dt_kernel<true>, g, dim3(TX, TY), 0, s, P, r, *c, D.cv, cfl, bits);
    else hipLaunchKernelGGL(diffusion_dt_kernel<false>, g, dim3(TX, TY), 0, s, P, r, *c, D.cv, cfl, bits);
  }
}

#ifdef VS_PROF
extern "C" int artemis_hip_debug_vs_prof(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vs_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_vs_prof), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif
} // namespace artemis
