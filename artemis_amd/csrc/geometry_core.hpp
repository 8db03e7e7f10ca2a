// Device-side geometry::Coords<GEOM> (reference geometry/geometry.hpp:145-420 for the Cartesian
// defaults, cylindrical.hpp, spherical.hpp, axisymmetric.hpp for the overrides).
//
// One struct serves the six coordinate systems through a wave-uniform runtime switch: these
// per-task kernels are bandwidth-bound and the metric arithmetic is a few dozen flops, so the
// 6x template fan-out of the reference buys nothing here.  Cell edges come from the same
// Xf(idx) = xf0 + idx*dx formula the Cartesian kernels use.  Every transcendental the
// reference evaluates per cell (cos/sin of the x2 faces, of the x2 centroid and of the x2
// midpoint) depends on j only and is read from a per-block table that the host fills with
// libm (artemis_hip_metric_fill), so results do not depend on a device sin/cos
// implementation; everything else is the reference's expression tree evaluated in place.
//
// This header holds the system-independent core (usable from the g++-compiled host driver for
// problem generators and history integrals as well as from kernels); geometry.hpp adds the
// PackView glue for device code.
#pragma once
#include <cmath>

#include "../../include/artemis_hip.h"

#ifdef __HIPCC__
#define GDEV __host__ __device__ __forceinline__
#else
#define GDEV inline
#endif

namespace artemis {

// Metric tables of one block: MT_ROWS rows of nj + 1 doubles indexed by j (x2 faces / cells),
// then MT3_ROWS rows of nk + 1 doubles indexed by k (cos / sin of the x3 cell centre: the azimuth of
// spherical3D and axisymmetric coordinates, used by ConvertCoordsToCart).
enum { MT_COSF = 0, MT_SINF = 1, MT_X2V = 2, MT_SINV = 3, MT_SINC = 4, MT_COSV = 5, MT_ROWS = 6 };
enum { MT3_COS = 0, MT3_SIN = 1, MT3_ROWS = 2 };
GDEV long metric_block_stride(int nj, int nk) {
  return static_cast<long>(MT_ROWS) * (nj + 1) + static_cast<long>(MT3_ROWS) * (nk + 1);
}

// CACHED = true (device kernels that rebuild a cell's Coords every plane of a march from small LDS tables instead
// of keeping ~100 doubles of geometry per thread in registers, kernels_curv.hip / kernels_diffusion.hip): the few
// member functions that contain a DIVISION -- x1v, rcen, rsq3 / 3, the connection coefficients, and the quotient
// inside hx3v -- return values that were computed once per workgroup by the CACHED = false functions of the same
// struct (the same expression trees, hence the same bits) and travel in the k_* fields; everything else is the
// multiplications and additions below, evaluated where they are needed.  k_rdc is the refined reciprocal of
// |cos(x2f0) - cos(x2f1)| (device_math.hpp `recip`: `div` through it returns the bits of `/` for these
// order-one operands, as everywhere in the march kernels).
template <bool CACHED>
struct DCoordsT {
  int sys;
  double x1[2], x2[2], x3[2];
  double cf[2], sf[2]; // cos / sin of the two x2 faces         (spherical2D/3D only)
  double x2c, sv, sc;  // x2 centroid, sin(centroid), sin(0.5*(x2[0]+x2[1]))
  double cv;           // cos(centroid) (basis vectors of ConvertVecToCyl, spherical.hpp:198-205)
  double c3, s3;       // cos / sin of the x3 cell centre (spherical3D, axisymmetric)
  double k_x1v, k_rcen, k_rfac, k_dh2dx1, k_dh3dx1, k_dh3dx2, k_rdc; // CACHED only (see above)

  GDEV bool sph23() const { return sys == ARTEMIS_SPHERICAL2D || sys == ARTEMIS_SPHERICAL3D; }
  GDEV bool sph() const { return sys == ARTEMIS_SPHERICAL1D || sph23(); }
  // geometry.hpp:98-110
  GDEV bool x1dep() const { return sys != ARTEMIS_CARTESIAN; }
  GDEV bool x2dep() const { return sph23(); }

  // sum of squares-and-product that every curvilinear radius formula shares
  GDEV double rsq3() const { return x1[0] * x1[0] + x1[0] * x1[1] + x1[1] * x1[1]; }

  GDEV double x1v() const {
    if constexpr (CACHED) return k_x1v;
    if (sph()) { // spherical.hpp:57-60
      const double dr2 = x1[0] * x1[0] + x1[1] * x1[1];
      return 0.75 * (x1[0] + x1[1]) * dr2 / (dr2 + x1[0] * x1[1]);
    }
    if (sys == ARTEMIS_CYLINDRICAL || sys == ARTEMIS_AXISYMMETRIC) // cylindrical.hpp:41-45
      return 2.0 / 3.0 * rsq3() / (x1[0] + x1[1]);
    return 0.5 * (x1[0] + x1[1]);
  }
  GDEV double x2v() const { return sph23() ? x2c : 0.5 * (x2[0] + x2[1]); }
  GDEV double x3v() const { return 0.5 * (x3[0] + x3[1]); }
  GDEV double rcen() const {
    if constexpr (CACHED) return k_rcen;
    return 2.0 / 3.0 * rsq3() / (x1[0] + x1[1]);
  }
  GDEV double rfac() const { // the radial factor of the spherical volume (spherical.hpp:126)
    if constexpr (CACHED) return k_rfac;
    return rsq3() / 3.0;
  }

  // volume-averaged scale factors (GetScaleFactors, geometry.hpp:384-388)
  GDEV double hx2v() const { // spherical.hpp:70, cylindrical.hpp:51; spherical1D keeps 1
    return (sph23() || sys == ARTEMIS_CYLINDRICAL) ? x1v() : 1.0;
  }
  GDEV double hx3v() const {
    if (sph23()) { // spherical.hpp:71-82
      const double dsc = sf[1] * cf[1] - sf[0] * cf[0];
      const double dx2 = x2[1] - x2[0];
#ifdef __HIPCC__
      if constexpr (CACHED) {
        const double num = x1v() * 0.5 * (dx2 - dsc), den = fabs(cf[0] - cf[1]);
        const double q = num * k_rdc; // device_math.hpp div(): the refined-reciprocal form of num / den
        return __builtin_fma(__builtin_fma(-den, q, num), k_rdc, q);
      }
#endif
      return x1v() * 0.5 * (dx2 - dsc) / fabs(cf[0] - cf[1]);
    }
    if (sys == ARTEMIS_AXISYMMETRIC) return x1v(); // axisymmetric.hpp:46
    return 1.0;
  }
  // GetCellWidths (geometry.hpp:352-361): h_d(x1v, x2v, x3v) * coordinate width
  GDEV double width1() const { return 1.0 * (x1[1] - x1[0]); }
  GDEV double width2() const {
    const double h = (sph() || sys == ARTEMIS_CYLINDRICAL) ? x1v() : 1.0;
    return h * (x2[1] - x2[0]);
  }
  GDEV double width3() const {
    double h = 1.0;
    if (sph23()) h = x1v() * sv;                        // spherical.hpp:53-55
    else if (sys == ARTEMIS_AXISYMMETRIC) h = x1v();    // axisymmetric.hpp:43-45
    return h * (x3[1] - x3[0]);
  }

  // face areas; f = 0 lower, 1 upper (GetFaceAreaX?, geometry.hpp:390-405)
  GDEV double area1(int f) const {
    const double x1f = x1[f];
    const double dx2 = x2[1] - x2[0], dx3 = x3[1] - x3[0];
    switch (sys) {
    case ARTEMIS_SPHERICAL3D: return x1f * x1f * fabs(cf[0] - cf[1]) * dx3; // spherical.hpp:106-109
    case ARTEMIS_SPHERICAL2D: return x1f * x1f * fabs(cf[0] - cf[1]);
    case ARTEMIS_SPHERICAL1D: return x1f * x1f;
    case ARTEMIS_CYLINDRICAL:
    case ARTEMIS_AXISYMMETRIC: return x1f * dx2 * dx3; // cylindrical.hpp:62-66
    default: return dx2 * dx3;
    }
  }
  GDEV double area2(int f) const {
    const double dx1 = x1[1] - x1[0], dx3 = x3[1] - x3[0];
    switch (sys) {
    case ARTEMIS_SPHERICAL3D: return 0.5 * (x1[1] + x1[0]) * sf[f] * dx1 * dx3; // spherical.hpp:110-114
    case ARTEMIS_SPHERICAL2D: return 0.5 * (x1[1] + x1[0]) * sf[f] * dx1;
    case ARTEMIS_SPHERICAL1D: return 0.5 * (x1[1] + x1[0]) * dx1;
    case ARTEMIS_AXISYMMETRIC: return (x1[0] + x1[1]) * 0.5 * dx1 * dx3; // axisymmetric.hpp:60-64
    default: return dx1 * dx3;
    }
  }
  GDEV double area3(int) const {
    const double dx1 = x1[1] - x1[0], dx2 = x2[1] - x2[0];
    switch (sys) {
    case ARTEMIS_SPHERICAL3D:
    case ARTEMIS_SPHERICAL2D:
    case ARTEMIS_CYLINDRICAL: return 0.5 * (x1[0] + x1[1]) * dx1 * dx2; // spherical.hpp:115-119
    case ARTEMIS_SPHERICAL1D: return 0.5 * (x1[0] + x1[1]) * dx1;
    default: return dx1 * dx2;
    }
  }
  GDEV double volume() const {
    const double dx1 = x1[1] - x1[0], dx2 = x2[1] - x2[0], dx3 = x3[1] - x3[0];
    if (sph()) { // spherical.hpp:124-133
      const double rfac = this->rfac();
      if (sys == ARTEMIS_SPHERICAL1D) return rfac * dx1;
      const double dc = fabs(cf[0] - cf[1]);
      if (sys == ARTEMIS_SPHERICAL2D) return rfac * dx1 * dc;
      return rfac * dx1 * dc * dx3;
    }
    if (sys == ARTEMIS_CYLINDRICAL || sys == ARTEMIS_AXISYMMETRIC) // cylindrical.hpp:73-78
      return (x1[0] + x1[1]) * 0.5 * dx1 * dx2 * dx3;
    return dx1 * dx2 * dx3;
  }
  // connection coefficients (GetConnX1/X2, geometry.hpp:407-418)
  GDEV double dh2dx1() const {
    if constexpr (CACHED) return k_dh2dx1;
    if (sph()) return 3.0 / 2.0 * (x1[0] + x1[1]) / rsq3(); // spherical.hpp:135-138
    if (sys == ARTEMIS_CYLINDRICAL) return 1.0 / (0.5 * (x1[0] + x1[1])); // cylindrical.hpp:80
    return 0.0;
  }
  GDEV double dh3dx1() const {
    if constexpr (CACHED) return k_dh3dx1;
    if (sph()) return 3.0 / 2.0 * (x1[0] + x1[1]) / rsq3(); // spherical.hpp:139-142
    if (sys == ARTEMIS_AXISYMMETRIC) return 1.0 / (0.5 * (x1[0] + x1[1])); // axisymmetric.hpp:71
    return 0.0;
  }
  GDEV double dh3dx2() const { // spherical.hpp:143-146
    if constexpr (CACHED) return k_dh3dx2;
    return sph23() ? (sf[1] - sf[0]) / fabs(cf[0] - cf[1]) : 0.0;
  }
  // frames of the cell centre with the tabulated trigonometry
  GDEV void centre(double xv[3]) const { xv[0] = x1v(), xv[1] = x2v(), xv[2] = x3v(); }
  // RFWeights (cylindrical.hpp:82-87, axisymmetric.hpp:73-78, spherical.hpp:148-169 / :349-370 /
  // :515-526; zero for Cartesian): +-(<R^2>_face - <R^2>)
  GDEV void rf_weights(double bx1[2], double bx2[2], double bx3[2]) const {
    bx1[0] = bx1[1] = bx2[0] = bx2[1] = bx3[0] = bx3[1] = 0.0;
    if (sys == ARTEMIS_CYLINDRICAL || sys == ARTEMIS_AXISYMMETRIC) {
      const double ans = 0.5 * (x1[0] + x1[1]) * (x1[1] - x1[0]);
      bx1[0] = bx1[1] = ans;
    } else if (sph23()) {
      const double rv = x1v();
      const double stv = sv;
      const double rf = rcen();
      const double r2cyl = (rv * stv) * (rv * stv);
      bx1[0] = r2cyl - (x1[0] * stv) * (x1[0] * stv), bx1[1] = (x1[1] * stv) * (x1[1] * stv) - r2cyl;
      bx2[0] = r2cyl - (rf * sf[0]) * (rf * sf[0]), bx2[1] = (rf * sf[1]) * (rf * sf[1]) - r2cyl;
    } else if (sys == ARTEMIS_SPHERICAL1D) {
      const double rv = x1v();
      const double r2cyl = rv * rv;
      bx1[0] = r2cyl - x1[0] * x1[0], bx1[1] = x1[1] * x1[1] - r2cyl;
    }
  }
  // ConvertCoordsToCart of this cell's centre (geometry.hpp:248, cylindrical.hpp:88-92,
  // spherical.hpp:166-173 / :355-362 / :528-534, axisymmetric.hpp:77-82); trigonometry from the tables
  GDEV void centre_to_cart(double xc[3]) const {
    const double a = x1v(), b = x2v(), c = x3v();
    switch (sys) {
    case ARTEMIS_SPHERICAL3D: xc[0] = a * sv * c3, xc[1] = a * sv * s3, xc[2] = a * cv; break;
    case ARTEMIS_SPHERICAL2D: xc[0] = a * sv * 1.0, xc[1] = a * sv * 0.0, xc[2] = a * cv; break;
    case ARTEMIS_SPHERICAL1D: xc[0] = a * 1.0 * 1.0, xc[1] = a * 1.0 * 0.0, xc[2] = a * 0.0; break;
    case ARTEMIS_CYLINDRICAL: xc[0] = a * cv, xc[1] = a * sv, xc[2] = c; break;
    case ARTEMIS_AXISYMMETRIC: xc[0] = a * c3, xc[1] = a * s3, xc[2] = b; break;
    default: xc[0] = a, xc[1] = b, xc[2] = c;
    }
  }
  // Scale factors at the centroid of the LOWER face of direction dir (ScaleMomentumFlux,
  // fluid_fluxes.hpp:56-66 with FaceCenX? of each system).
  GDEV void face_scale(int dir, double h[3]) const {
    h[0] = 1.0, h[1] = 1.0, h[2] = 1.0;
    if (dir == 1) { // FaceCenX1 = {x1f, x2v, x3v} for every system (geometry.hpp:182-186)
      if (sph() || sys == ARTEMIS_CYLINDRICAL) h[1] = x1[0];
      if (sph23()) h[2] = x1[0] * sv;
      else if (sys == ARTEMIS_AXISYMMETRIC) h[2] = x1[0];
    } else if (dir == 2) {
      // spherical*/axisymmetric use the area-weighted radius, cylindrical/cartesian x1v
      const bool rc = sph() || sys == ARTEMIS_AXISYMMETRIC;
      const double r = rc ? rcen() : x1v();
      if (sph() || sys == ARTEMIS_CYLINDRICAL) h[1] = r;
      if (sph23()) h[2] = r * sf[0];
      else if (sys == ARTEMIS_AXISYMMETRIC) h[2] = r;
    } else {
      const bool rc = sph() || sys == ARTEMIS_CYLINDRICAL;
      const double r = rc ? rcen() : x1v();
      if (sph() || sys == ARTEMIS_CYLINDRICAL) h[1] = r;
      if (sph23()) h[2] = r * sc;
      else if (sys == ARTEMIS_AXISYMMETRIC) h[2] = r;
    }
  }
};
using DCoords = DCoordsT<false>;

// ConvertToCylWithVec / ConvertToCartWithVec (geometry.hpp:438-482): the converted point x and the
// rows e1, e2, e3 = components of the problem's unit vectors in the target basis.  (ct, st) =
// cos / sin of xi[1] for the spherical systems, (cp, sp) = cos / sin of the azimuth (xi[1]
// cylindrical, xi[2] spherical3D / axisymmetric): the caller supplies them -- host libm values in
// problem generators, the tabulated values of a cell centre on the device.
struct Frame {
  double x[3], e1[3], e2[3], e3[3];
};
GDEV void set3(double a[3], double x, double y, double z) { a[0] = x, a[1] = y, a[2] = z; }
GDEV Frame cyl_frame(int sys, const double xi[3], double ct, double st) {
  Frame f;
  switch (sys) {
  case ARTEMIS_CARTESIAN: { // geometry.hpp:286-301; atan2 is not evaluated (no caller reads x[1])
    const double R = sqrt(xi[0] * xi[0] + xi[1] * xi[1]);
    const double cp = xi[0] / (R + 1e-99);
    const double sp = xi[1] / (R + 1e-99);
    set3(f.x, R, 0.0, xi[2]);
    set3(f.e1, cp, -sp, 0.0), set3(f.e2, sp, cp, 0.0), set3(f.e3, 0.0, 0.0, 1.0);
  } break;
  case ARTEMIS_CYLINDRICAL: // cylindrical.hpp:128-136
    set3(f.x, xi[0], xi[1], xi[2]);
    set3(f.e1, 1.0, 0.0, 0.0), set3(f.e2, 0.0, 1.0, 0.0), set3(f.e3, 0.0, 0.0, 1.0);
    break;
  case ARTEMIS_SPHERICAL3D: // spherical.hpp:202-220, :403-421
  case ARTEMIS_SPHERICAL2D:
    set3(f.x, xi[0] * st, (sys == ARTEMIS_SPHERICAL3D) ? xi[2] : 0.0, xi[0] * ct);
    set3(f.e1, st, 0.0, ct), set3(f.e2, ct, 0.0, -st), set3(f.e3, 0.0, 1.0, 0.0);
    break;
  case ARTEMIS_SPHERICAL1D: // :559-577: ct = 0, st = 1
    set3(f.x, xi[0] * 1.0, 0.0, xi[0] * 0.0);
    set3(f.e1, 1.0, 0.0, 0.0), set3(f.e2, 0.0, 0.0, -1.0), set3(f.e3, 0.0, 1.0, 0.0);
    break;
  default: // axisymmetric.hpp:135-145
    set3(f.x, xi[0], xi[2], xi[1]);
    set3(f.e1, 1.0, 0.0, 0.0), set3(f.e2, 0.0, 0.0, 1.0), set3(f.e3, 0.0, 1.0, 0.0);
  }
  return f;
}
GDEV Frame cart_frame(int sys, const double xi[3], double ct, double st, double cp, double sp) {
  Frame f;
  switch (sys) {
  case ARTEMIS_CYLINDRICAL: // cylindrical.hpp:96-107
    set3(f.x, xi[0] * cp, xi[0] * sp, xi[2]);
    set3(f.e1, cp, sp, 0.0), set3(f.e2, -sp, cp, 0.0), set3(f.e3, 0.0, 0.0, 1.0);
    break;
  case ARTEMIS_SPHERICAL3D: // spherical.hpp:172-189
    set3(f.x, xi[0] * st * cp, xi[0] * st * sp, xi[0] * ct);
    set3(f.e1, st * cp, st * sp, ct), set3(f.e2, ct * cp, ct * sp, -st), set3(f.e3, -sp, cp, 0.0);
    break;
  default: // Cartesian (the other systems do not come through here)
    set3(f.x, xi[0], xi[1], xi[2]);
    set3(f.e1, 1.0, 0.0, 0.0), set3(f.e2, 0.0, 1.0, 0.0), set3(f.e3, 0.0, 0.0, 1.0);
  }
  return f;
}
// RotatingFrame::RotationVelocity<GEOM>(xv, omf) (rotating_frame.hpp:31-47) at a cell centre
template <class CO>
GDEV void rotation_velocity(const CO &co, double omf, double vf[3]) {
  vf[0] = 0.0, vf[1] = omf, vf[2] = 0.0; // Cartesian (every dh/dx is zero there)
  if (co.sys == ARTEMIS_CARTESIAN) return;
  double xv[3];
  co.centre(xv);
  const Frame fr = cyl_frame(co.sys, xv, co.cv, co.sv);
  const double vp = omf * fr.x[0];
  vf[0] = fr.e1[1] * vp, vf[1] = fr.e2[1] * vp, vf[2] = fr.e3[1] * vp;
}
// ConvertToSph(xi)[0] (geometry.hpp:262-264, cylindrical.hpp:111-112, axisymmetric.hpp:116-117)
GDEV double sph_radius(int sys, const double xi[3]) {
  switch (sys) {
  case ARTEMIS_CARTESIAN: {
    const double R = sqrt(xi[0] * xi[0] + xi[1] * xi[1]);
    return sqrt(R * R + xi[2] * xi[2]);
  }
  case ARTEMIS_CYLINDRICAL: return sqrt(xi[0] * xi[0] + xi[2] * xi[2]);
  case ARTEMIS_AXISYMMETRIC: return sqrt(xi[0] * xi[0] + xi[1] * xi[1]);
  default: return xi[0];
  }
}

// Cell (k,j,i) of a block with edge table g6 = {x1f0, dx1, x2f0, dx2, x3f0, dx3}; `m` = the
// block's metric rows (stride nj+1) or null when the system needs none.
GDEV DCoords coords_of(int sys, const double *g, const double *m, int nj, int nk, int k, int j, int i) {
  DCoords c;
  c.sys = sys;
  c.x1[0] = g[0] + i * g[1], c.x1[1] = g[0] + (i + 1) * g[1];
  c.x2[0] = g[2] + j * g[3], c.x2[1] = g[2] + (j + 1) * g[3];
  c.x3[0] = g[4] + k * g[5], c.x3[1] = g[4] + (k + 1) * g[5];
  c.cf[0] = c.cf[1] = c.sf[0] = c.sf[1] = c.x2c = c.sv = c.sc = c.cv = 0.0;
  c.c3 = 1.0, c.s3 = 0.0;
  if (m == nullptr) return c;
  const int st = nj + 1;
  if (c.sph23()) {
    c.cf[0] = m[MT_COSF * st + j], c.cf[1] = m[MT_COSF * st + j + 1];
    c.sf[0] = m[MT_SINF * st + j], c.sf[1] = m[MT_SINF * st + j + 1];
    c.x2c = m[MT_X2V * st + j], c.sv = m[MT_SINV * st + j], c.sc = m[MT_SINC * st + j];
    c.cv = m[MT_COSV * st + j];
  } else if (sys == ARTEMIS_CYLINDRICAL) { // cos / sin of the azimuth x2v (ConvertCoordsToCart only)
    c.sv = m[MT_SINV * st + j], c.cv = m[MT_COSV * st + j];
  }
  if (sys == ARTEMIS_SPHERICAL3D || sys == ARTEMIS_AXISYMMETRIC) {
    const double *m3 = m + static_cast<long>(MT_ROWS) * st;
    c.c3 = m3[MT3_COS * (nk + 1) + k], c.s3 = m3[MT3_SIN * (nk + 1) + k];
  }
  return c;
}

} // namespace artemis
