// Cell body of Gas::DiffusionUpdate (DiffusionUpdateImpl, utils/diffusion/diffusion.hpp:110-241), shared by
// the per-task kernel (kernels_diffusion.hip) and the general fused stage (kernels_stage_cell.hip).
#pragma once
#include "device_math.hpp"
#include "geometry.hpp"
#include "pack_view.hpp"
#include "task_device.hpp"

namespace artemis {

// DiffusionCoeff<DIFF>::Get of cell c of block b (also the drag package's damp_to_visc, drag.hpp:240,393); the radial factors come from the host-filled table.
// Dynamic viscosity with the cell's radial factor already in hand (`rad`: dp.radial[b][c], or 1 where the law has no
// table): the march kernels fetch it with the cell's primitives, a plane ahead of its use.  `fast`: the caller has
// checked (wave-wide) that no numerator is tiny, so the refined-reciprocal division returns the bits of `/`.
ADEV double viscosity_of(const artemis_diffcoeff_t &dp, double gm1, double dens, double sie, double rad, bool fast = false) {
  if (dp.type == ARTEMIS_VISCOSITY_PLAW) return dp.coeff * dens * rad; // diffusion_coeff.hpp:222-224
  const double blk = (gm1 + 1.0) * gm1 * dens * sie; // :262-268: alpha B / Omega_K, B = gamma gm1 rho sie (IdealGas)
  return fast ? div(dp.coeff * blk, rad) : dp.coeff * blk / rad;
}
ADEV double coeff_of(const artemis_diffcoeff_t &dp, double cv, double gm1, double dens, double sie, int b,
                     long c) {
  switch (dp.type) {
  case ARTEMIS_VISCOSITY_PLAW:
    return viscosity_of(dp, gm1, dens, sie, dp.radial ? dp.radial[b][c] : 1.0);
  case ARTEMIS_VISCOSITY_ALPHA:
    return viscosity_of(dp, gm1, dens, sie, dp.radial[b][c]);
  default: { // conductivity_plaw :312-316, thermaldiff_plaw :353-359
    // zero exponents (every shipped deck): std::pow(x, 0.0) == 1.0, bit-exact.  Otherwise the power laws of
    // the STATE run on the device's pow(): agreement with a host libm is to rounding, not bitwise.
    double ft = 1.0, fr = 1.0;
    if (dp.temp_exp != 0.0) ft = pow(amax(0.0, sie / cv) / dp.T_ref, dp.temp_exp);
    if (dp.rho_exp != 0.0) fr = pow(dens / dp.rho_ref, dp.rho_exp);
    if (dp.type == ARTEMIS_CONDUCTIVITY_PLAW) return dp.coeff * ft * fr;
    return dp.coeff * ft * fr * dens * cv;
  }
  }
}

struct DiffCell { // geometry of one cell as the update uses it
  double ax1[2], ax2[2], ax3[2], vol, hx[3], dhdx1[3], dhdx2[3];
  int x1dep, x2dep, multi_d, three_d;
};
// the same record from a cell's Coords (curvilinear callers that already hold them)
template <class CO>
__device__ __forceinline__ DiffCell diffusion_cell_of(const CO &co, const CellMetric &m, const double hx[3], int ndim) {
  DiffCell d;
  d.multi_d = (ndim > 1), d.three_d = (ndim > 2);
  d.ax1[0] = m.ax1[0], d.ax1[1] = m.ax1[1];
  d.ax2[0] = d.multi_d ? m.ax2[0] : 0.0, d.ax2[1] = d.multi_d ? m.ax2[1] : 0.0;
  d.ax3[0] = d.three_d ? m.ax3[0] : 0.0, d.ax3[1] = d.three_d ? m.ax3[1] : 0.0;
  d.vol = m.vol;
  d.hx[0] = hx[0], d.hx[1] = hx[1], d.hx[2] = hx[2];
  for (int q = 0; q < 3; ++q) d.dhdx1[q] = 0.0, d.dhdx2[q] = 0.0;
  d.x1dep = co.x1dep(), d.x2dep = co.x2dep() && d.multi_d;
  if (d.x1dep) d.dhdx1[1] = co.dh2dx1(), d.dhdx1[2] = co.dh3dx1();
  if (d.x2dep) d.dhdx2[2] = co.dh3dx2();
  return d;
}
template <bool CURV>
__device__ __forceinline__ DiffCell diffusion_cell(const PackView &P, int b, int k, int j, int i) {
  DiffCell d;
  d.multi_d = (P.ndim > 1), d.three_d = (P.ndim > 2);
  const CellMetric m = cell_metric<CURV>(P, b, k, j, i);
  d.ax1[0] = m.ax1[0], d.ax1[1] = m.ax1[1];
  d.ax2[0] = d.multi_d ? m.ax2[0] : 0.0, d.ax2[1] = d.multi_d ? m.ax2[1] : 0.0;
  d.ax3[0] = d.three_d ? m.ax3[0] : 0.0, d.ax3[1] = d.three_d ? m.ax3[1] : 0.0;
  d.vol = m.vol;
  scale_factors<CURV>(P, b, k, j, i, d.hx);
  // GetConnX1 = {0, dh2dx1, dh3dx1}, GetConnX2 = {0, 0, dh3dx2}, GetConnX3 = 0 (geometry.hpp:407-418)
  for (int q = 0; q < 3; ++q) d.dhdx1[q] = 0.0, d.dhdx2[q] = 0.0;
  d.x1dep = 0, d.x2dep = 0;
  if constexpr (CURV) {
    const DCoords co = make_coords(P, b, k, j, i);
    d.x1dep = co.x1dep(), d.x2dep = co.x2dep() && d.multi_d;
    if (d.x1dep) d.dhdx1[1] = co.dh2dx1(), d.dhdx1[2] = co.dh3dx1();
    if (d.x2dep) d.dhdx2[2] = co.dh3dx2();
  }
  return d;
}
// what the update subtracts from the momenta (dm), total energy (de) and internal energy (deg) of
// species n of a cell; v = the stage-input primitive velocity of the cell.  F(d, var, up) = the diffusion flux
// `var` (3 n + component, or 3 ns + n for the energy) through the cell's lower (up = 0) or upper (up = 1) face
// of direction d; for an inactive direction the caller returns a finite stand-in (it is multiplied by 0).
// FAST (the march kernels): the seven quotients by geometry -- four by the cell volume, three by the scale factors --
// through shared refined reciprocals (device_math.hpp: the bits of `/` while no numerator is tiny-but-nonzero); the
// wave looks at its numerators first and takes the plain divisions if any lane holds such a value.
template <bool FAST = false, class FX>
__device__ __forceinline__ void diffusion_update_core(const DiffCell &g, const FX &F, int n, int ns, int do_viscosity,
                                                      double dt, const double v[3], double dm[3], double &de,
                                                      double &deg) {
  const int multi_d = g.multi_d, three_d = g.three_d;
  auto divergence = [&](int var) {
    return (g.ax1[0] * F(0, var, 0) - g.ax1[1] * F(0, var, 1)) +
           multi_d * (g.ax2[0] * F(1, var, 0) - g.ax2[1] * F(1, var, 1)) +
           three_d * (g.ax3[0] * F(2, var, 0) - g.ax3[1] * F(2, var, 1));
  };
  const int imx1 = 3 * n + 0, imx2 = 3 * n + 1, imx3 = 3 * n + 2, ien = 3 * ns + n;
  auto metric_src = [&](const double dh[3]) {
    return dh[0] * 0.5 * (F(0, imx1, 0) + F(0, imx1, 1)) +
           multi_d * dh[1] * 0.5 * (F(1, imx2, 0) + F(1, imx2, 1)) +
           three_d * dh[2] * 0.5 * (F(2, imx3, 0) + F(2, imx3, 1));
  };
  auto small = [](double x) { return x != 0.0 && fabs(x) < 0x1p-200; }; // (geometry.hpp tiny_nonzero)
  double divfxm = 0., divfym = 0., divfzm = 0.;
  if (do_viscosity) divfxm = divergence(imx1), divfym = divergence(imx2), divfzm = divergence(imx3);
  double divfe = divergence(ien);
  bool fastv = false;
  Recip rvol{};
  if constexpr (FAST) {
    fastv = !__any(small(divfxm) || small(divfym) || small(divfzm) || small(divfe));
    if (fastv) rvol = recip(g.vol);
  }
  auto by_vol = [&](double x) { return (FAST && fastv) ? div(x, rvol) : x / g.vol; };
  if (do_viscosity) {
    const double zero3[3] = {0.0, 0.0, 0.0};
    divfxm = by_vol(divfxm);
    divfxm += g.x1dep * metric_src(g.dhdx1);
    divfym = by_vol(divfym);
    divfym += g.x2dep * metric_src(g.dhdx2);
    divfzm = by_vol(divfzm);
    divfzm += 0 * metric_src(zero3); // x3dep is false for every system (geometry.hpp:107-110)
  }
  divfe = by_vol(divfe);
  dm[0] = dt * divfxm, dm[1] = dt * divfym, dm[2] = dt * divfzm;
  de = dt * divfe;
  const double w1 = divfxm * v[0], w2 = divfym * v[1], w3 = divfzm * v[2];
  bool fasth = false;
  if constexpr (FAST) fasth = !__any(small(w1) || small(w2) || small(w3));
  auto by_h = [&](double x, double h) { return (FAST && fasth) ? div(x, h) : x / h; };
  deg = dt * divfe - dt * (by_h(w1, g.hx[0]) + by_h(w2, g.hx[1]) + by_h(w3, g.hx[2]));
}
// ... with the fluxes read from the pack's diffusion-flux arrays at cell c of block b
__device__ __forceinline__ void diffusion_update_cell(const PackView &P, const DiffCell &g, int b, int n, long c,
                                                      int do_viscosity, double dt, const double v[3],
                                                      double dm[3], double &de, double &deg) {
  const FluidView &f = P.gas;
  const int ns = f.ns, nq = 4 * ns;
  const long up[3] = {c + 1, c + g.multi_d * P.sj, c + g.three_d * P.sk};
  const int dd[3] = {0, g.multi_d ? 1 : 0, g.three_d ? 2 : 0}; // inactive directions have no flux table
  auto F = [&](int d, int var, int u) { return f.dflux[dd[d]][b * nq + var][u ? up[d] : c]; };
  diffusion_update_core(g, F, n, ns, do_viscosity, dt, v, dm, de, deg);
}

} // namespace artemis
