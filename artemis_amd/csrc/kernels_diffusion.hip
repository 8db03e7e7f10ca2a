// Gas diffusion tasks (artemis_driver.cpp:189-193, :218-221), Cartesian coordinates, constant
// coefficients: ZeroDiffusionFlux, ViscousFlux (momentum_diffusion.hpp), ThermalFlux
// (thermal_diffusion.hpp), DiffusionUpdate and the diffusive timestep (diffusion.hpp).
//
// The reference walks pencils with scratch rows for the strain tensor, div(u) and the
// coefficient; every face value is a pure function of the primitives around the face, so here
// one thread owns one face (thread x walks i, coalesced) and evaluates its neighbourhood
// directly -- the cells involved are shared through L1/L2 with the neighbouring threads.
// Scale factors are 1 and the connection coefficients 0 in Cartesian coordinates; the terms are
// kept (multiplications by 1.0 / additions of 0.0 * v) so that NaN/Inf propagate as in the
// reference's arithmetic.
#include <cfloat>

#include "device_math.hpp"
#include "kernels.hpp"
#include "pack_view.hpp"

namespace artemis {
namespace {
constexpr int TX = 64, TY = 4;

struct Box {
  int il, iu, jl, ju, kl, ku;
};
inline dim3 grid_of(const Box &r, int nb) {
  return dim3((r.iu - r.il + TX) / TX, (r.ju - r.jl + TY) / TY, (r.ku - r.kl + 1) * nb);
}
#define BOX_CELL(r)                                                                         \
  const int i = (r).il + blockIdx.x * TX + threadIdx.x;                                    \
  const int j = (r).jl + blockIdx.y * TY + threadIdx.y;                                    \
  const int nkr = (r).ku - (r).kl + 1;                                                     \
  const int b = blockIdx.z / nkr;                                                          \
  const int k = (r).kl + blockIdx.z % nkr;                                                 \
  if (i > (r).iu || j > (r).ju) return;                                                    \
  const long c = (static_cast<long>(k) * P.nj + j) * P.ni + i;

struct Cart { // cell centres and widths of one block (geometry.hpp:163-166, :199-225)
  const double *g;
  ADEV double x1v(int i) const { return 0.5 * ((g[0] + i * g[1]) + (g[0] + (i + 1) * g[1])); }
  ADEV double x2v(int j) const { return 0.5 * ((g[2] + j * g[3]) + (g[2] + (j + 1) * g[3])); }
  ADEV double x3v(int k) const { return 0.5 * ((g[4] + k * g[5]) + (g[4] + (k + 1) * g[5])); }
  ADEV double dx1(int i) const { return (g[0] + (i + 1) * g[1]) - (g[0] + i * g[1]); }
  ADEV double dx2(int j) const { return (g[2] + (j + 1) * g[3]) - (g[2] + j * g[3]); }
  ADEV double dx3(int k) const { return (g[4] + (k + 1) * g[5]) - (g[4] + k * g[5]); }
  // Coords::Distance (geometry.hpp:407-412) between the centres of two cells
  ADEV double dist(int k1, int j1, int i1, int k2, int j2, int i2) const {
    return sqrt(sqr(x1v(i1) - x1v(i2)) + sqr(x2v(j1) - x2v(j2)) + sqr(x3v(k1) - x3v(k2)));
  }
};

// DiffusionCoeff<DIFF>::Get with zero exponents: std::pow(x, 0.0) == 1.0 for every x
ADEV double coeff_of(const artemis_diffcoeff_t &dp, double cv, double dens) {
  switch (dp.type) {
  case ARTEMIS_VISCOSITY_PLAW: return dp.coeff * dens * 1.0;       // diffusion_coeff.hpp:222-224
  case ARTEMIS_CONDUCTIVITY_PLAW: return dp.coeff * 1.0 * 1.0;     // :312-316
  default: return dp.coeff * 1.0 * 1.0 * dens * cv;                // thermaldiff_plaw :353-359
  }
}
ADEV double face_average(int avg, double mu1, double mu2) { // diffusion_coeff.hpp:139-150
  return (avg == 0) * (0.5 * (mu1 + mu2)) + (avg == 1) * (2.0 * mu1 * mu2 / (mu1 + mu2));
}

__global__ __launch_bounds__(TX *TY) void zero_dflux_kernel(const PackView P, const Box r) {
  BOX_CELL(r)
  const int nv = 4 * P.gas.ns;
  for (int d = 0; d < P.ndim; ++d)
    for (int n = 0; n < nv; ++n) P.gas.dflux[d][b * nv + n][c] = 0.0;
}

// VelocityDivergence (momentum_diffusion.hpp:562-591) of cell (k,j,i), species n
ADEV double velocity_divergence(const PackView &P, const Cart &ge, double *const *prim, int b, int n,
                                int k, int j, int i) {
  const int ns = P.gas.ns, nv = 6 * ns;
  const int multid = (P.ndim >= 2), threed = (P.ndim == 3);
  const double d1 = ge.dx1(i), d2 = ge.dx2(j), d3 = ge.dx3(k);
  const double vol = d1 * d2 * d3;
  const double a1 = d2 * d3, a2 = multid ? d1 * d3 : 0.0, a3 = threed ? d1 * d2 : 0.0;
  const double *v1 = prim[b * nv + ns + 3 * n + 0], *v2 = prim[b * nv + ns + 3 * n + 1];
  const double *v3 = prim[b * nv + ns + 3 * n + 2];
  const long c = (static_cast<long>(k) * P.nj + j) * P.ni + i;
  const long sj = multid * P.sj, sk = threed * P.sk;
  const double divv = a1 * (v1[c] + v1[c + 1]) - a1 * (v1[c] + v1[c - 1]) +
                      multid * a2 * (v2[c] + v2[c + sj]) - multid * a2 * (v2[c] + v2[c - sj]) +
                      threed * a3 * (v3[c] + v3[c + sk]) - threed * a3 * (v3[c] + v3[c - sk]);
  return divv / (2.0 * vol);
}

// MomentumFluxImpl (momentum_diffusion.hpp:597-755): StrainTensorFace<XDIR> (:28-377) and
// StressTensorFaceX? (:379-560) of the lower `dir` face of cell (k,j,i)
template <int DIR>
__global__ __launch_bounds__(TX *TY) void viscous_flux_kernel(const PackView P, const Box r,
                                                              const artemis_diffusion_t D) {
  BOX_CELL(r)
  const FluidView &f = P.gas;
  const int ns = f.ns, nv = 6 * ns, nq = 4 * ns;
  const int multid = (P.ndim >= 2), threed = (P.ndim == 3);
  const Cart ge{P.geom + 6 * b};
  const artemis_diffcoeff_t &dp = D.visc;
  constexpr int dk = (DIR == 3), dj = (DIR == 2), di = (DIR == 1);
  const long cm = c - ((DIR == 1) ? 1 : ((DIR == 2) ? P.sj : P.sk));
  const double fuzz = 1e-99; // Fuzz<Real>()
  for (int n = 0; n < ns; ++n) {
    const double *q[3] = {f.prim[b * nv + ns + 3 * n + 0], f.prim[b * nv + ns + 3 * n + 1],
                          f.prim[b * nv + ns + 3 * n + 2]};
    auto sv = [&](int comp, int kk, int jj, int ii) { // v / hx of the cell, hx = 1
      return q[comp][(static_cast<long>(kk) * P.nj + jj) * P.ni + ii] / 1.0;
    };
    auto src_of = [&](int kk, int jj, int ii) { // v^k dh_a/dx_k / h_a: every dh is zero
      return sv(0, kk, jj, ii) * 0.0 + sv(1, kk, jj, ii) * 0.0 + sv(2, kk, jj, ii) * 0.0;
    };
    const double v[3] = {sv(0, k, j, i), sv(1, k, j, i), sv(2, k, j, i)};
    double flx[3];
    if constexpr (DIR == 1) {
      const double dx1 = ge.dist(k, j, i, k, j, i - 1);
      const double dx2 = multid ? ge.dist(k, j - multid, i, k, j + multid, i) : fuzz;
      const double dx2_xm = multid ? ge.dist(k, j - multid, i - 1, k, j + multid, i - 1) : fuzz;
      const double dx3 = threed ? ge.dist(k - threed, j, i, k + threed, j, i) : fuzz;
      const double dx3_xm = threed ? ge.dist(k - threed, j, i - 1, k + threed, j, i - 1) : fuzz;
      const double dv1 = v[0] - sv(0, k, j, i - 1);
      flx[0] = 2 * dv1 / dx1 + 0.5 * (src_of(k, j, i) + src_of(k, j, i - 1));
      const double dv2 = v[1] - sv(1, k, j, i - 1);
      const double dv12 = sv(0, k, j + multid, i) - sv(0, k, j - multid, i);
      const double dv12_xm = sv(0, k, j + multid, i - 1) - sv(0, k, j - multid, i - 1);
      flx[1] = multid * 0.5 * (dv12 / dx2 + dv12_xm / dx2_xm) + sqr(1.0 / 1.0) * dv2 / dx1;
      const double dv3 = v[2] - sv(2, k, j, i - 1);
      const double dv13 = sv(0, k + threed, j, i) - sv(0, k - threed, j, i);
      const double dv13_xm = sv(0, k + threed, j, i - 1) - sv(0, k - threed, j, i - 1);
      flx[2] = threed * 0.5 * (dv13 / dx3 + dv13_xm / dx3_xm) + sqr(1.0 / 1.0) * dv3 / dx1;
    } else if constexpr (DIR == 2) {
      const double dx1 = ge.dist(k, j, i - 1, k, j, i + 1);
      const double dx1_ym = ge.dist(k, j - 1, i - 1, k, j - 1, i + 1);
      const double dx2 = ge.dist(k, j, i, k, j - 1, i);
      const double dx3 = threed ? ge.dist(k - threed, j, i, k + threed, j, i) : fuzz;
      const double dx3_ym = threed ? ge.dist(k - threed, j - 1, i, k + threed, j - 1, i) : fuzz;
      const double dv1 = v[0] - sv(0, k, j - 1, i);
      const double dv21 = sv(1, k, j, i + 1) - sv(1, k, j, i - 1);
      const double dv21_ym = sv(1, k, j - 1, i + 1) - sv(1, k, j - 1, i - 1);
      flx[0] = 0.5 * (dv21 / dx1 + dv21_ym / dx1_ym) + sqr(1.0 / 1.0) * dv1 / dx2;
      const double dv2 = v[1] - sv(1, k, j - 1, i);
      flx[1] = 2 * dv2 / dx2 + 0.5 * (src_of(k, j, i) + src_of(k, j - 1, i));
      const double dv3 = v[2] - sv(2, k, j - 1, i);
      const double dv23 = sv(1, k + threed, j, i) - sv(1, k - threed, j, i);
      const double dv23_ym = sv(1, k + threed, j - 1, i) - sv(1, k - threed, j - 1, i);
      flx[2] = threed * 0.5 * (dv23 / dx3 + dv23_ym / dx3_ym) + sqr(1.0 / 1.0) * dv3 / dx2;
    } else {
      const double dx1 = ge.dist(k, j, i - 1, k, j, i + 1);
      const double dx1_zm = ge.dist(k - 1, j, i - 1, k - 1, j, i + 1);
      const double dx2 = ge.dist(k, j - 1, i, k, j + 1, i);
      const double dx2_zm = ge.dist(k - 1, j - 1, i, k - 1, j + 1, i);
      const double dx3 = ge.dist(k, j, i, k - 1, j, i);
      const double dv1 = v[0] - sv(0, k - 1, j, i);
      const double dv31 = sv(2, k, j, i + 1) - sv(2, k, j, i - 1);
      const double dv31_zm = sv(2, k - 1, j, i + 1) - sv(2, k - 1, j, i - 1);
      flx[0] = 0.5 * (dv31 / dx1 + dv31_zm / dx1_zm) + sqr(1.0 / 1.0) * dv1 / dx3;
      const double dv2 = v[1] - sv(1, k - 1, j, i);
      const double dv32 = sv(2, k, j + 1, i) - sv(2, k, j - 1, i);
      const double dv32_zm = sv(2, k - 1, j + 1, i) - sv(2, k - 1, j - 1, i);
      flx[1] = 0.5 * (dv32 / dx2 + dv32_zm / dx2_zm) + sqr(1.0 / 1.0) * dv2 / dx3;
      const double dv3 = v[2] - sv(2, k - 1, j, i);
      flx[2] = 2 * dv3 / dx3 + 0.5 * (src_of(k, j, i) + src_of(k - 1, j, i));
    }
    const double *rho = f.prim[b * nv + n];
    const double mu = coeff_of(dp, D.cv, rho[c]), mu_m = coeff_of(dp, D.cv, rho[cm]);
    const double mus = face_average(dp.avg, mu, mu_m);
    const double divu = velocity_divergence(P, ge, f.prim, b, n, k, j, i);
    const double divu_m = velocity_divergence(P, ge, f.prim, b, n, k - dk, j - dj, i - di);
    const double hf = 1.0;
    double fl[3];
    for (int qq = 0; qq < 3; ++qq) fl[qq] = hf * mus * flx[qq];
    fl[DIR - 1] = hf * mus * (flx[DIR - 1] - 1. / 3 * (1. - dp.eta) * (divu + divu_m));
    double *const *qf = f.dflux[DIR - 1];
    for (int qq = 0; qq < 3; ++qq) qf[b * nq + 3 * n + qq][c] += fl[qq];
    qf[b * nq + 3 * ns + n][c] += 0.5 * (q[0][c] / 1.0 + q[0][cm] / 1.0) * fl[0] +
                                  0.5 * (q[1][c] / 1.0 + q[1][cm] / 1.0) * fl[1] +
                                  0.5 * (q[2][c] / 1.0 + q[2][cm] / 1.0) * fl[2];
  }
}

// ThermalFluxImpl (thermal_diffusion.hpp:30-222)
template <int DIR>
__global__ __launch_bounds__(TX *TY) void thermal_flux_kernel(const PackView P, const Box r,
                                                              const artemis_diffusion_t D) {
  BOX_CELL(r)
  const FluidView &f = P.gas;
  const int ns = f.ns, nv = 6 * ns, nq = 4 * ns;
  const Cart ge{P.geom + 6 * b};
  const artemis_diffcoeff_t &dp = D.cond;
  constexpr int dk = (DIR == 3), dj = (DIR == 2), di = (DIR == 1);
  const long cm = c - ((DIR == 1) ? 1 : ((DIR == 2) ? P.sj : P.sk));
  const double dx = ge.dist(k, j, i, k - dk, j - dj, i - di);
  for (int n = 0; n < ns; ++n) {
    const double *rho = f.prim[b * nv + n], *se = f.prim[b * nv + 5 * ns + n];
    const double T = amax(0.0, se[c] / D.cv);   // IdealGas TemperatureFromDensityInternalEnergy
    const double Tm = amax(0.0, se[cm] / D.cv);
    const double kcond = face_average(dp.avg, coeff_of(dp, D.cv, rho[c]), coeff_of(dp, D.cv, rho[cm]));
    f.dflux[DIR - 1][b * nq + 3 * ns + n][c] += kcond * (T - Tm) / dx;
  }
}

// DiffusionUpdateImpl (diffusion.hpp:110-241), Cartesian: no metric sources
__global__ __launch_bounds__(TX *TY) void diffusion_update_kernel(const PackView P, const Box r,
                                                                  int do_viscosity, double dt) {
  BOX_CELL(r)
  const FluidView &f = P.gas;
  const int ns = f.ns, nv = 6 * ns, nq = 4 * ns;
  const int multi_d = (P.ndim > 1), three_d = (P.ndim > 2);
  const Cart ge{P.geom + 6 * b};
  const double d1 = ge.dx1(i), d2 = ge.dx2(j), d3 = ge.dx3(k);
  const double ax1 = d2 * d3, ax2 = multi_d ? d1 * d3 : 0.0, ax3 = three_d ? d1 * d2 : 0.0;
  const double vol = d1 * d2 * d3;
  const long c2 = c + multi_d * P.sj, c3 = c + three_d * P.sk;
  for (int n = 0; n < ns; ++n) {
    auto F = [&](int d, int var, long cc) { return f.dflux[d][b * nq + var][cc]; };
    auto divergence = [&](int var) {
      return (ax1 * F(0, var, c) - ax1 * F(0, var, c + 1)) +
             multi_d * (ax2 * F(multi_d ? 1 : 0, var, c) - ax2 * F(multi_d ? 1 : 0, var, c2)) +
             three_d * (ax3 * F(three_d ? 2 : 0, var, c) - ax3 * F(three_d ? 2 : 0, var, c3));
    };
    const int imx1 = 3 * n + 0, imx2 = 3 * n + 1, imx3 = 3 * n + 2, ien = 3 * ns + n;
    double divfxm = 0., divfym = 0., divfzm = 0.;
    if (do_viscosity) {
      divfxm = divergence(imx1);
      divfxm /= vol;
      divfxm += 0 * 0.0; // x1dep * src (false in Cartesian coordinates)
      divfym = divergence(imx2);
      divfym /= vol;
      divfym += 0 * 0.0;
      divfzm = divergence(imx3);
      divfzm /= vol;
      divfzm += 0 * 0.0;
    }
    double divfe = divergence(ien);
    divfe /= vol;
    f.cons0[b * nv + ns + 3 * n + 0][c] -= dt * divfxm;
    f.cons0[b * nv + ns + 3 * n + 1][c] -= dt * divfym;
    f.cons0[b * nv + ns + 3 * n + 2][c] -= dt * divfzm;
    f.cons0[b * nv + 4 * ns + n][c] -= dt * divfe;
    f.cons0[b * nv + 5 * ns + n][c] -=
        dt * divfe - dt * (divfxm * f.prim[b * nv + ns + 3 * n + 0][c] / 1.0 +
                           divfym * f.prim[b * nv + ns + 3 * n + 1][c] / 1.0 +
                           divfzm * f.prim[b * nv + ns + 3 * n + 2][c] / 1.0);
  }
}

// Diffusion::EstimateTimestep (diffusion.hpp:66-108) for one coefficient; grid-stride reduction
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
    const Cart ge{P.geom + 6 * b};
    const double dx[3] = {1.0 * ge.dx1(i), 1.0 * ge.dx2(j), 1.0 * ge.dx3(k)};
    double min_dx = DBL_MAX;
    for (int d = 0; d < P.ndim; d++) min_dx = amin(min_dx, dx[d]);
    const int ns = P.gas.ns, nv = 6 * ns;
    for (int n = 0; n < ns; ++n) {
      const double dens = P.gas.prim[b * nv + n][c];
      double mu = coeff_of(dp, cv, dens);
      if (dp.type == ARTEMIS_CONDUCTIVITY_PLAW) mu /= (dens * cv);
      else if (dp.type == ARTEMIS_VISCOSITY_PLAW) mu *= (1.0 + (dp.eta > 1.0) * (dp.eta - 1.0)) / dens;
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
  hipLaunchKernelGGL(zero_dflux_kernel, grid_of(r, P.nb), dim3(TX, TY), 0, s, P, r);
}
void launch_viscous_flux(const PackView &P, const artemis_diffusion_t &D, hipStream_t s) {
  const dim3 t(TX, TY);
  hipLaunchKernelGGL(viscous_flux_kernel<1>, grid_of(faces(P, 1), P.nb), t, 0, s, P, faces(P, 1), D);
  if (P.ndim > 1) hipLaunchKernelGGL(viscous_flux_kernel<2>, grid_of(faces(P, 2), P.nb), t, 0, s, P, faces(P, 2), D);
  if (P.ndim > 2) hipLaunchKernelGGL(viscous_flux_kernel<3>, grid_of(faces(P, 3), P.nb), t, 0, s, P, faces(P, 3), D);
}
void launch_thermal_flux(const PackView &P, const artemis_diffusion_t &D, hipStream_t s) {
  const dim3 t(TX, TY);
  hipLaunchKernelGGL(thermal_flux_kernel<1>, grid_of(faces(P, 1), P.nb), t, 0, s, P, faces(P, 1), D);
  if (P.ndim > 1) hipLaunchKernelGGL(thermal_flux_kernel<2>, grid_of(faces(P, 2), P.nb), t, 0, s, P, faces(P, 2), D);
  if (P.ndim > 2) hipLaunchKernelGGL(thermal_flux_kernel<3>, grid_of(faces(P, 3), P.nb), t, 0, s, P, faces(P, 3), D);
}
void launch_diffusion_update(const PackView &P, const artemis_diffusion_t &D, double dt, hipStream_t s) {
  const Box r = interior(P);
  hipLaunchKernelGGL(diffusion_update_kernel, grid_of(r, P.nb), dim3(TX, TY), 0, s, P, r,
                     D.visc.type != ARTEMIS_DIFF_OFF ? 1 : 0, dt);
}
void launch_diffusion_dt(const PackView &P, const artemis_diffusion_t &D, double cfl, double *dt_dev,
                         hipStream_t s) {
  const Box r = interior(P);
  const dim3 g3 = grid_of(r, P.nb);
  const long ntile = static_cast<long>(g3.x) * g3.y * g3.z;
  const dim3 g(static_cast<unsigned>(ntile < 4096 ? ntile : 4096));
  auto *bits = reinterpret_cast<unsigned long long *>(dt_dev);
  if (D.visc.type != ARTEMIS_DIFF_OFF)
    hipLaunchKernelGGL(diffusion_dt_kernel, g, dim3(TX, TY), 0, s, P, r, D.visc, D.cv, cfl, bits);
  if (D.cond.type != ARTEMIS_DIFF_OFF)
    hipLaunchKernelGGL(diffusion_dt_kernel, g, dim3(TX, TY), 0, s, P, r, D.cond, D.cv, cfl, bits);
}

} // namespace artemis
