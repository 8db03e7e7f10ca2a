// Device-side view of an artemis_pack_t (include/artemis_hip.h): plain struct passed BY VALUE
// as a kernel argument, so kernels read the pointer tables with scalar loads.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/artemis_hip.h"

namespace artemis {

struct FluidView {
  int ns;
  double dfloor, siefloor, de_switch;
  double *const *prim;
  double *const *cons0;
  double *const *cons1;
  double *const *flux[3];
  double *const *pflux[3];
  double *const *vface[3];
  double *const *dflux[3]; // diffusion fluxes (gas)
};

struct PackView {
  int nb, ndim, ng;
  int ni, nj, nk;             // array extents incl. ghosts
  int is, ie, js, je, ks, ke; // interior bounds (inclusive)
  long sj, sk;                // strides of j and k
  double gm1;
  const double *geom; // [nb][6]
  int coords;           // enum artemis_coords
  const double *metric; // [nb][6][nj+1] x2 trig tables (spherical2D/3D), else null
  double omf;           // frame frequency for FluxSource's coordinate sources
  const double *plm_tab; // PLM_G weights per (block, direction, index) (artemis_hip_plm_table_fill), or null
  int plm_len;           // row length of that table: max(ni, nj, nk)
  FluidView gas, dust;
};
constexpr int PLM_TAB_ROWS = 9; // cr, cl, up, lo, ra.b, ra.y, rb.b, rb.y, x1v (direction 1 rows only)

inline FluidView make_fluid_view(const artemis_fluid_pack_t &f) {
  FluidView v;
  v.ns = f.nspecies;
  v.dfloor = f.dfloor, v.siefloor = f.siefloor, v.de_switch = f.de_switch;
  v.prim = f.prim, v.cons0 = f.cons0, v.cons1 = f.cons1;
  for (int d = 0; d < 3; ++d)
    v.flux[d] = f.flux[d], v.pflux[d] = f.pflux[d], v.vface[d] = f.vface[d], v.dflux[d] = f.diff_flux[d];
  return v;
}

inline PackView make_pack_view(const artemis_pack_t &p) {
  PackView v;
  v.nb = p.nblocks;
  v.ndim = (p.nx3 > 1) ? 3 : ((p.nx2 > 1) ? 2 : 1);
  v.ng = p.nghost;
  const int g1 = p.nghost, g2 = (p.nx2 > 1) ? p.nghost : 0, g3 = (p.nx3 > 1) ? p.nghost : 0;
  v.ni = p.nx1 + 2 * g1, v.nj = p.nx2 + 2 * g2, v.nk = p.nx3 + 2 * g3;
  v.is = g1, v.ie = g1 + p.nx1 - 1;
  v.js = g2, v.je = g2 + p.nx2 - 1;
  v.ks = g3, v.ke = g3 + p.nx3 - 1;
  v.sj = v.ni, v.sk = static_cast<long>(v.ni) * v.nj;
  v.gm1 = p.gm1;
  v.geom = p.geom;
  v.coords = p.coords;
  v.metric = p.metric;
  v.omf = p.omega_frame;
  v.plm_tab = p.plm_table;
  v.plm_len = (v.ni > v.nj) ? ((v.ni > v.nk) ? v.ni : v.nk) : ((v.nj > v.nk) ? v.nj : v.nk);
  v.gas = make_fluid_view(p.gas);
  v.dust = make_fluid_view(p.dust);
  return v;
}

} // namespace artemis
