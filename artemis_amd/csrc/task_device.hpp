// Device helpers shared by the per-task kernels (kernels_unfused.hip) and the cell-centred
// fused stage (kernels_stage_cell.hip): both must evaluate the same expression trees.
#pragma once
#include "device_math.hpp"
#include "geometry.hpp"
#include "pack_view.hpp"

namespace artemis {

// Face states of one variable at the face below cell c: L from the cell below, R from cell c.
// CURV && PLM uses PLM_G with the Mignone weights of each of the two cells (plm.hpp:90-103).
template <int RECON, bool CURV>
__device__ __forceinline__ void face_states(const double *q, long st, const PlmGeo &gl,
                                            const PlmGeo &gr, double &L, double &R) {
  double unused;
  if constexpr (CURV && RECON == 1) {
    plm_g_shared(q[-2 * st], q[-st], q[0], L, unused, gl);
    plm_g_shared(q[-st], q[0], q[st], unused, R, gr);
  } else {
    recon_cell<RECON>(q - st, st, L, unused), recon_cell<RECON>(q, st, unused, R);
  }
}


// Face areas, volume and coordinate widths of one cell (GetFaceAreaX?, Volume, geometry.hpp:199-225
// and the curvilinear overrides).
struct CellMetric {
  double ax1[2], ax2[2], ax3[2], vol; // GetFaceAreaX?, Volume
  double dx[3];                       // coordinate widths bnds.x?[1] - bnds.x?[0]
};
template <class CO>
__device__ __forceinline__ CellMetric cell_metric_of(const CO &co) {
  CellMetric m;
  m.ax1[0] = co.area1(0), m.ax1[1] = co.area1(1);
  m.ax2[0] = co.area2(0), m.ax2[1] = co.area2(1);
  m.ax3[0] = co.area3(0), m.ax3[1] = co.area3(1);
  m.vol = co.volume();
  m.dx[0] = co.x1[1] - co.x1[0], m.dx[1] = co.x2[1] - co.x2[0], m.dx[2] = co.x3[1] - co.x3[0];
  return m;
}
template <bool CURV>
__device__ __forceinline__ CellMetric cell_metric(const PackView &P, int b, int k, int j, int i) {
  CellMetric m;
  if constexpr (CURV) {
    m = cell_metric_of(make_coords(P, b, k, j, i));
  } else {
    const CellGeom g = cell_geom(P.geom + 6 * b, k, j, i);
    m.ax1[0] = m.ax1[1] = g.dx2 * g.dx3; // geometry.hpp:199-204
    m.ax2[0] = m.ax2[1] = g.dx1 * g.dx3; // :205-210
    m.ax3[0] = m.ax3[1] = g.dx1 * g.dx2; // :211-216
    m.vol = g.dx1 * g.dx2 * g.dx3;       // :219-225
    m.dx[0] = g.dx1, m.dx[1] = g.dx2, m.dx[2] = g.dx3;
  }
  return m;
}


template <bool CURV>
__device__ __forceinline__ void scale_factors(const PackView &P, int b, int k, int j, int i,
                                              double hx[3]) {
  hx[0] = 1.0, hx[1] = 1.0, hx[2] = 1.0; // GetScaleFactors (geometry.hpp:384-388)
  if constexpr (CURV) {
    const DCoords co = make_coords(P, b, k, j, i);
    hx[1] = co.hx2v(), hx[2] = co.hx3v();
  }
}

} // namespace artemis
