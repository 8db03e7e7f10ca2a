// Register-level forms of the pointwise source tasks, shared by the per-task kernels
// (kernels_sources.hip) and the cell-centred fused stage (kernels_stage_cell.hip).
#pragma once
#include <cfloat>
#include <cmath>

#include "device_math.hpp"
#include "geometry.hpp"
#include "pack_view.hpp"

namespace artemis {

struct GasCons {
  double d, m1, m2, m3, e, eg;
};
struct DustCons {
  double d, m1, m2, m3;
};
struct FluidPrim { // rho, velocity and (gas) specific internal energy of one species in one cell
  double rho, v1, v2, v3, sie;
};

// Volume-averaged scale factors of any system (GetScaleFactors, geometry.hpp:384-388)
template <class CO>
ADEV void scale_factors_of(const CO &co, double hx[3]) {
  hx[0] = 1.0, hx[1] = co.hx2v(), hx[2] = co.hx3v();
}

// PrimToCons of one cell (fill_derived.cpp:229-274): floors are re-applied like the reference does
// (F: a FluidView, or any record with dfloor / siefloor -- the march kernels keep the floors in LDS)
template <class F>
ADEV GasCons prim_to_cons_gas(const F &f, double d, double v1, double v2, double v3,
                              double se, const double hx[3]) {
  GasCons u;
  const double w_d = (d > f.dfloor) ? d : f.dfloor;
  u.d = w_d;
  u.m1 = w_d * v1 * hx[0], u.m2 = w_d * v2 * hx[1], u.m3 = w_d * v3 * hx[2];
  const double w_s = (se > f.siefloor) ? se : f.siefloor;
  u.eg = w_s * w_d;
  const double ke = 0.5 * w_d * (sqr(v1) + sqr(v2) + sqr(v3));
  u.e = u.eg + ke;
  return u;
}
template <class F>
ADEV DustCons prim_to_cons_dust(const F &f, double d, double v1, double v2, double v3,
                                const double hx[3]) {
  DustCons u;
  const double w_d = (d > f.dfloor) ? d : f.dfloor;
  u.d = w_d;
  u.m1 = w_d * v1 * hx[0], u.m2 = w_d * v2 * hx[1], u.m3 = w_d * v3 * hx[2];
  return u;
}


// ---- Gravity::ExternalGravity (gravity.cpp:126-155) ---------------------------------------
// acceleration components along the coordinate basis and the sink fraction of one cell
struct GravAcc {
  double gx1, gx2, gx3, fd;
  bool uniform;
};
// FAST (the march kernels, which wait for their own instructions): the point-mass law of the Cartesian-frame systems
// with the hand-scheduled square root and division of device_math.hpp wherever the whole wave holds positive normal
// operands (the cylindrical radius of a zone centre vanishes only on the axis; such a wave takes the plain forms), and
// the sink's two quotients skipped when no lane has a sink rate: the fraction is then exactly + 0.0
// (min(0.5, x) * false, NaN-safe as written).  Same bits as FAST = false.
template <bool FAST = false, class CO, class GR>
ADEV GravAcc gravity_accel(const GR &G, const CO &co, int ndim, double dt) {
  GravAcc a;
  a.gx1 = 0.0, a.gx2 = 0.0, a.gx3 = 0.0, a.fd = 0.0;
  a.uniform = (G.type == ARTEMIS_GRAVITY_UNIFORM);
  if (a.uniform) { // uniform.cpp:39-41
    a.gx1 = G.g[0], a.gx2 = G.g[1], a.gx3 = G.g[2];
    return a;
  }
  const double dx[3] = {co.x1v(), co.x2v(), co.x3v()};
  const bool multi_d = ndim >= 2, three_d = ndim == 3;
  const double gm = G.gm, rsft2 = sqr(G.soft);
  if (G.type == ARTEMIS_GRAVITY_BINARY) { // binary_mass.cpp:86-160
    const bool cyl = (co.sys == ARTEMIS_CYLINDRICAL);
    const Frame fr = cart_frame(co.sys, dx, co.cv, co.sv, cyl ? co.cv : co.c3, cyl ? co.sv : co.s3);
    const double mu1 = 1. / (1.0 + G.q), mu2 = G.q / (1.0 + G.q);
    double dxc1[3] = {fr.x[0], fr.x[1], fr.x[2]}, dxc2[3];
    for (int n = 0; n < 3; n++) {
      dxc2[n] = dxc1[n] - G.pos2[n];
      dxc1[n] -= G.pos[n];
    }
    const double R1 = sqrt(dxc1[0] * dxc1[0] + dxc1[1] * dxc1[1]), R2 = sqrt(dxc2[0] * dxc2[0] + dxc2[1] * dxc2[1]);
    const double r1 = sqrt(R1 * R1 + dxc1[2] * dxc1[2]), r2 = sqrt(R2 * R2 + dxc2[2] * dxc2[2]);
    const double rad2_1 = sqr(r1) + sqr(G.soft);
    const double rad2_2 = sqr(r2) + sqr(G.soft2);
    const double idr3_1 = 1.0 / (sqrt(rad2_1) * rad2_1);
    const double idr3_2 = 1.0 / (sqrt(rad2_2) * rad2_2);
    const double g[3] = {-gm * (mu1 * dxc1[0] * idr3_1 + mu2 * dxc2[0] * idr3_2),
                         multi_d * (-gm * (mu1 * dxc1[1] * idr3_1 + mu2 * dxc2[1] * idr3_2)),
                         three_d * (-gm * (mu1 * dxc1[2] * idr3_1 + mu2 * dxc2[2] * idr3_2))};
    a.gx1 = g[0] * fr.e1[0] + g[1] * fr.e1[1] + g[2] * fr.e1[2];
    a.gx2 = g[0] * fr.e2[0] + g[1] * fr.e2[1] + g[2] * fr.e2[2];
    a.gx3 = g[0] * fr.e3[0] + g[1] * fr.e3[1] + g[2] * fr.e3[2];
    const double sr1 = dt * G.sink_rate, sr2 = dt * G.sink_rate2;
    const double sramp1 = sr1 * sqr((r1 - G.sink) / G.sink);
    const double sramp2 = sr2 * sqr((r2 - G.sink2) / G.sink2);
    const double s1 = sramp1 / (1.0 + sramp1), s2 = sramp2 / (1.0 + sramp2);
    double fd1 = (s1 < 0.25) ? s1 : 0.25, fd2 = (s2 < 0.25) ? s2 : 0.25; // std::min(0.25, s): NaN keeps 0.25
    fd1 *= ((sr1 > 0.0) && (G.sink > 0.0) && (r1 <= G.sink));
    fd2 *= ((sr2 > 0.0) && (G.sink2 > 0.0) && (r2 <= G.sink2));
    a.fd = fd1 + fd2;
    return a;
  }
  double dr;
  if (co.sys == ARTEMIS_SPHERICAL1D || co.sys == ARTEMIS_SPHERICAL2D) { // point_mass.cpp:78-81
    const double rad2 = sqr(dx[0]) + rsft2;
    a.gx1 = -gm / rad2;
    dr = sqrt(rad2);
  } else if (co.sys == ARTEMIS_AXISYMMETRIC) { // :82-89
    const double rsph = sqrt(dx[0] * dx[0] + dx[1] * dx[1]);
    const double ct = dx[1] / (rsph + 1e-99);
    const double st = dx[0] / (rsph + 1e-99);
    dr = rsph;
    const double rad2 = sqr(dr) + rsft2;
    const double g = -gm / rad2;
    a.gx1 = g * st;
    a.gx2 = g * ct;
  } else { // Cartesian, cylindrical, spherical3D: through the Cartesian frame (:91-112)
    const bool cyl = (co.sys == ARTEMIS_CYLINDRICAL);
    const Frame fr = cart_frame(co.sys, dx, co.cv, co.sv, cyl ? co.cv : co.c3, cyl ? co.sv : co.s3);
    double dxc[3] = {fr.x[0], fr.x[1], fr.x[2]};
    for (int n = 0; n < 3; n++) dxc[n] -= G.pos[n];
    const double R2 = dxc[0] * dxc[0] + dxc[1] * dxc[1];
    double idr3;
    if (FAST && !__any(!(R2 > 0x1p-400) || !(R2 < 0x1p400) || !(fabs(dxc[2]) < 0x1p200) || !(rsft2 < 0x1p400))) {
      const double R = sqrt_pos(R2);
      const double r = sqrt_pos(R * R + dxc[2] * dxc[2]);
      dr = r;
      const double rad2 = sqr(dr) + rsft2;
      idr3 = div(1.0, recip(sqrt_pos(rad2) * rad2));
    } else {
      const double R = sqrt(R2);
      const double r = sqrt(R * R + dxc[2] * dxc[2]);
      dr = r;
      const double rad2 = sqr(dr) + rsft2;
      idr3 = 1.0 / (sqrt(rad2) * rad2);
    }
    const double g[3] = {-gm * dxc[0] * idr3, (multi_d) * (-gm * dxc[1] * idr3),
                         (three_d) * (-gm * dxc[2] * idr3)};
    a.gx1 = g[0] * fr.e1[0] + g[1] * fr.e1[1] + g[2] * fr.e1[2];
    a.gx2 = g[0] * fr.e2[0] + g[1] * fr.e2[1] + g[2] * fr.e2[2];
    a.gx3 = g[0] * fr.e3[0] + g[1] * fr.e3[1] + g[2] * fr.e3[2];
  }
  const double sink_rate = dt * G.sink_rate;
  if (FAST && !__any(sink_rate != 0.0)) { // (sramp is 0 or NaN, sfrac 0 or NaN, the minimum 0 or 0.5, times false)
    a.fd = 0.0;
    return a;
  }
  const double sramp = sink_rate * sqr((dr - G.sink) / G.sink); // quad_ramp, gravity.hpp:116
  const double sfrac = sramp / (1.0 + sramp);
  a.fd = (sfrac < 0.5) ? sfrac : 0.5; // std::min(0.5, sfrac): a NaN ratio (sink = 0) keeps 0.5
  a.fd *= ((sink_rate > 0.0) && (dr <= G.sink));
  return a;
}
// uniform.cpp:58-68 / point_mass.cpp:137-156
ADEV void gravity_gas(const GravAcc &a, double dt, const double hx[3], const FluidPrim &w,
                      GasCons &u) {
  if (a.uniform) {
    const double rdt = dt * w.rho;
    u.m1 += rdt * hx[0] * a.gx1, u.m2 += rdt * hx[1] * a.gx2, u.m3 += rdt * hx[2] * a.gx3;
    u.e += rdt * (w.v1 * a.gx1 + w.v2 * a.gx2 + w.v3 * a.gx3);
  } else {
    const double tote = w.rho * (w.sie + 0.5 * (sqr(w.v1) + sqr(w.v2) + sqr(w.v3)));
    u.m1 += dt * w.rho * hx[0] * a.gx1, u.m2 += dt * w.rho * hx[1] * a.gx2;
    u.m3 += dt * w.rho * hx[2] * a.gx3;
    u.e += dt * w.rho * (w.v1 * a.gx1 + w.v2 * a.gx2 + w.v3 * a.gx3);
    u.d -= a.fd * w.rho;
    u.m1 -= a.fd * hx[0] * w.rho * w.v1, u.m2 -= a.fd * hx[1] * w.rho * w.v2;
    u.m3 -= a.fd * hx[2] * w.rho * w.v3;
    u.e -= a.fd * tote;
  }
}
// uniform.cpp:71-78 / point_mass.cpp:160-176
ADEV void gravity_dust(const GravAcc &a, double dt, const double hx[3], const FluidPrim &w,
                       DustCons &u) {
  if (a.uniform) {
    const double rdt = dt * w.rho;
    u.m1 += rdt * hx[0] * a.gx1, u.m2 += rdt * hx[1] * a.gx2, u.m3 += rdt * hx[2] * a.gx3;
  } else {
    u.m1 += dt * w.rho * hx[0] * a.gx1, u.m2 += dt * w.rho * hx[1] * a.gx2;
    u.m3 += dt * w.rho * hx[2] * a.gx3;
    u.d -= a.fd * w.rho;
    u.m1 -= a.fd * hx[0] * w.rho * w.v1, u.m2 -= a.fd * hx[1] * w.rho * w.v2;
    u.m3 -= a.fd * hx[2] * w.rho * w.v3;
  }
}

// ---- RotatingFrame::RotatingFrameImpl<GEOM> (rotating_frame_impl.hpp:95-199) ----------------
// flo / fup = the MASS flux through the lower / upper face of the cell along x1, x2, x3 (zero for
// inactive directions); ax* = face areas as the caller's task uses them.
struct RotFrame {
  double omdt, om2dt, R, eR[3], ep[3]; // e?[d] = component of the problem's unit vector d along R / phi
  double b1[2], b2[2], b3[2];
};
template <class CO>
ADEV RotFrame rotating_frame_terms(const CO &co, double om0, double dt) {
  RotFrame r;
  r.omdt = om0 * dt;
  r.om2dt = r.omdt * om0;
  double xv[3];
  co.centre(xv);
  const Frame fr = cyl_frame(co.sys, xv, co.cv, co.sv);
  r.R = fr.x[0];
  r.eR[0] = fr.e1[0], r.eR[1] = fr.e2[0], r.eR[2] = fr.e3[0];
  r.ep[0] = fr.e1[1], r.ep[1] = fr.e2[1], r.ep[2] = fr.e3[1];
  co.rf_weights(r.b1, r.b2, r.b3);
  return r;
}
ADEV double rotating_frame_divf(const RotFrame &r, int multi_d, int three_d, const double flo[3],
                                const double fup[3], const double ax1[2], const double ax2[2],
                                const double ax3[2]) {
  return (flo[0] * ax1[0] * r.b1[0] + fup[0] * ax1[1] * r.b1[1]) +
         multi_d * (flo[1] * ax2[0] * r.b2[0] + fup[1] * ax2[1] * r.b2[1]) +
         three_d * (flo[2] * ax3[0] * r.b3[0] + fup[2] * ax3[1] * r.b3[1]);
}
ADEV void rotating_frame_gas(const RotFrame &r, int multi_d, int three_d, const double flo[3],
                             const double fup[3], const double ax1[2], const double ax2[2],
                             const double ax3[2], double vol, GasCons &u) {
  const double divf = rotating_frame_divf(r, multi_d, three_d, flo, fup, ax1, ax2, ax3);
  u.m1 -= r.omdt * (divf / vol) * r.ep[0];
  u.m2 -= r.omdt * (divf / vol) * r.ep[1];
  u.m3 -= r.omdt * (divf / vol) * r.ep[2];
  const double fx[3] = {0.5 * (flo[0] + fup[0]), multi_d * 0.5 * (flo[1] + fup[1]),
                        three_d * 0.5 * (flo[2] + fup[2])};
  u.e += r.om2dt * r.R * (fx[0] * r.eR[0] + fx[1] * r.eR[1] + fx[2] * r.eR[2]);
}
ADEV void rotating_frame_dust(const RotFrame &r, int multi_d, int three_d, const double flo[3],
                              const double fup[3], const double ax1[2], const double ax2[2],
                              const double ax3[2], double vol, DustCons &u) {
  const double divf = rotating_frame_divf(r, multi_d, three_d, flo, fup, ax1, ax2, ax3);
  u.m1 -= r.omdt * (divf / vol) * r.ep[0];
  u.m2 -= r.omdt * (divf / vol) * r.ep[1];
  u.m3 -= r.omdt * (divf / vol) * r.ep[2];
}

// ---- Gas::Cooling::BetaCooling (beta_cooling.cpp:88-124) on one cell's conserved state ----------
template <class CO>
ADEV double cooling_omdt(const CO &co, double gm, double dt) {
  double xv[3];
  co.centre(xv);
  const Frame fr = cyl_frame(co.sys, xv, co.cv, co.sv);
  const double rsph2 = fr.x[0] * fr.x[0] + fr.x[2] * fr.x[2];
  const double ir1 = 1.0 / sqrt(rsph2);
  return dt * sqrt(gm * ir1 * ir1 * ir1);
}
ADEV void cooling_gas(const FluidView &G, double cv, double omdt, double T0, double beta, const double hx[3],
                      GasCons &u) {
  // GetSpecificInternalEnergy (artemis_utils.hpp:43-62)
  const double dens = u.d;
  const double u_d = amax(dens, G.dfloor);
  const double rv1 = u.m1 / hx[0], rv2 = u.m2 / hx[1], rv3 = u.m3 / hx[2];
  const double ke = 0.5 * (sqr(rv1) + sqr(rv2) + sqr(rv3)) / u_d;
  const double e_cons = u.e;
  const double ue_cons = e_cons - ke;
  double sie = (ue_cons > G.de_switch * e_cons) ? ue_cons / u_d : u.eg / u_d;
  sie = amax(sie, G.siefloor);
  const double Tn = amax(0.0, sie / cv);
  const double dE = -dens * cv * omdt / (beta + omdt) * (Tn - T0);
  u.e += dE, u.eg += dE;
}

// ---- Coords<GEOM>::ConvertToCylWithVec (geometry.hpp:476-482) ------------------------------------
// cylindrical radius of the cell centroid and the first component of each basis vector (geometry.hpp:289-306,
// cylindrical.hpp:117-126, spherical.hpp:191-205 / :382-396 / :556-577, axisymmetric.hpp:134-145).
struct CylVec {
  double R, e1, e2, e3;
};
template <class CO>
ADEV CylVec to_cyl_with_vec(const CO &co, const double xv[3]) {
  CylVec c;
  switch (co.sys) {
  case ARTEMIS_CARTESIAN: {
    const double R = sqrt(xv[0] * xv[0] + xv[1] * xv[1]);
    c.R = R, c.e1 = xv[0] / (R + 1e-99), c.e2 = xv[1] / (R + 1e-99), c.e3 = 0.0; // Fuzz<Real>()
  } break;
  case ARTEMIS_SPHERICAL3D:
  case ARTEMIS_SPHERICAL2D: c.R = xv[0] * co.sv, c.e1 = co.sv, c.e2 = co.cv, c.e3 = 0.0; break;
  case ARTEMIS_SPHERICAL1D: c.R = xv[0] * 1.0, c.e1 = 1.0, c.e2 = 0.0, c.e3 = 0.0; break;
  default: c.R = xv[0], c.e1 = 1.0, c.e2 = 0.0, c.e3 = 0.0;
  }
  return c;
}

// ---- Drag::DragSource, simple_dust (SimpleDragSourceImpl, drag.hpp:296-482) for ONE gas and ONE dust species on one
// zone's conserved state, followed by SetAuxillaryFields (fill_derived.cpp:58-71) and ConsToPrim (:132-164) of both
// fluids: the register-level form of kernels_sources.hip's simple_drag_kernel<FINISH = true, ND = 1> (same expression
// trees, statement by statement; that kernel reads and writes arrays, this one a march's registers).  No damp_to_visc
// (mu = 0 as DiffusionCoeff<null>::Get gives it); bg / bd = the damping ramps of the zone (+ 0.0 when no rate is set).
// FAST: every quotient through the hand-scheduled division of device_math.hpp -- the caller has checked that no momentum
// of the wave is tiny-but-nonzero and that every density is a positive normal number; otherwise IEEE `/` as written.
struct DragLaw1 {
  int stokes;
  double tau, scale, grain_density, size;
};
struct DragFinish1 {
  double gd, g1, g2, g3, gs; // gas rho, v, sie
  double dd, d1, d2, d3;     // dust rho, v
};
template <bool FAST, class FG, class FD>
ADEV DragFinish1 simple_drag1_finish(const DragLaw1 &D, const FG &G, const FD &F, const double gm1, const double dt,
                                     const double hx[3], const CylVec &cv, const double bg[3], const double bd[3],
                                     const GasCons &ug, const DustCons &ud) {
  auto dv = [](double num, double den) { return FAST ? div(num, den) : num / den; };
  const double dg = ug.d, e_cons = ug.e, eg_cons = ug.eg;
  const double mg0[3] = {ug.m1, ug.m2, ug.m3};
  const double dens = ud.d;
  const double md[3] = {ud.m1, ud.m2, ud.m3};
  const double vg[3] = {dv(mg0[0], hx[0] * dg), dv(mg0[1], hx[1] * dg), dv(mg0[2], hx[2] * dg)};
  double sieg; // GetSpecificInternalEnergy (artemis_utils.hpp:43-62), species 0
  {
    const double u_d = amax(dg, G.dfloor);
    const double rv1 = dv(mg0[0], hx[0]), rv2 = dv(mg0[1], hx[1]), rv3 = dv(mg0[2], hx[2]);
    const double ke = dv(0.5 * (sqr(rv1) + sqr(rv2) + sqr(rv3)), u_d);
    const double ue_cons = e_cons - ke;
    sieg = (ue_cons > G.de_switch * e_cons) ? dv(ue_cons, u_d) : dv(eg_cons, u_d);
    sieg = amax(sieg, G.siefloor);
  }
  const double mu = 0.0; // drag.hpp:392-393 with DiffusionCoeff<null>
  const double vR = -1.5 * mu / (cv.R * dg);
  const double vt[3] = {cv.e1 * vR, cv.e2 * vR, cv.e3 * vR};
  double fd[3] = {0., 0., 0.}, fvd[3] = {0., 0., 0.};
  double vth = 0.0;
  if (D.stokes) vth = sqrt(8.0 / M_PI * gm1 * sieg);
  const double vdt[3] = {0.0, 0.0, 0.0};
  const double vd[3] = {dv(md[0], hx[0] * dens), dv(md[1], hx[1] * dens), dv(md[2], hx[2] * dens)};
  double tc = D.tau;
  if (D.stokes) tc = D.scale * D.grain_density / dg * D.size / vth;
  const double alpha = dt * ((tc <= 0.0) ? DBL_MAX : 1.0 / tc);
  double rhopv[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const double rhop = dv(dens * alpha, 1.0 + alpha + bd[d]);
    rhopv[d] = rhop;
    fd[d] += rhop * (1.0 + bd[d]);
    fvd[d] += rhop * (vd[d] + bd[d] * vdt[d]);
  }
  double vgp[3];
#pragma unroll
  for (int d = 0; d < 3; d++) vgp[d] = dv(dg * (vg[d] + bg[d] * vt[d]) + fvd[d], dg * (1.0 + bg[d]) + fd[d]);
  double delta_g[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int d = 0; d < 3; d++) fvd[d] = 0.;
  DragFinish1 o;
  const double w_dd = (dens > F.dfloor) ? dens : F.dfloor;
  double newd[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    double delta_d = 0.;
    const double rhop = rhopv[d]; // (the reference's second pass re-evaluates the same expression)
    const double delta = rhop * ((vgp[d] - vd[d] + bd[d] * (vgp[d] - vdt[d])));
    delta_d += delta;
    delta_g[d] -= delta;
    delta_d -= dv(bd[d] * dens, 1. + alpha + bd[d]) * (vd[d] - vdt[d] + alpha * (vgp[d] - vdt[d]));
    fvd[d] += rhop * (vd[d] - vt[d] + bd[d] * (vdt[d] - vt[d]));
    const double m = md[d] + hx[d] * delta_d;
    newd[d] = dv(m, w_dd * hx[d]); // Dust ConsToPrim (fill_derived.cpp:155-164)
  }
  o.dd = w_dd, o.d1 = newd[0], o.d2 = newd[1], o.d3 = newd[2];
  double en = e_cons, mnew[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const double prefac = dv(dg * bg[d], 1.0 + bg[d] + fd[d]);
    delta_g[d] -= prefac * (dg * (vg[d] - vt[d]) + fvd[d]);
    mnew[d] = mg0[d] + hx[d] * delta_g[d];
    en += 0.5 * (vg[d] + vgp[d]) * delta_g[d];
  }
  // SetAuxillaryFields + ConsToPrim of the gas
  const double u_d = (dg > G.dfloor) ? dg : G.dfloor;
  const double u_d2 = amax(dg, G.dfloor);
  const double rv1 = dv(mnew[0], hx[0]), rv2 = dv(mnew[1], hx[1]), rv3 = dv(mnew[2], hx[2]);
  const double ke = dv(0.5 * (sqr(rv1) + sqr(rv2) + sqr(rv3)), u_d2);
  const double ue_cons = en - ke;
  double sie = (ue_cons > G.de_switch * en) ? dv(ue_cons, u_d2) : dv(eg_cons, u_d2);
  sie = amax(sie, G.siefloor);
  double u_u = sie * u_d;
  const double uflr = G.siefloor * u_d;
  u_u = (u_u > uflr) ? u_u : uflr;
  const double w_d = u_d;
  const double w_s = dv(u_u, w_d);
  o.gd = w_d;
  o.g1 = dv(mnew[0], w_d * hx[0]), o.g2 = dv(mnew[1], w_d * hx[1]), o.g3 = dv(mnew[2], w_d * hx[2]);
  o.gs = (w_s > G.siefloor) ? w_s : G.siefloor;
  return o;
}

// ---- RotatingFrame::ShearingBoxImpl (rotating_frame_impl.hpp:28-93) -------------------------
struct ShearAcc {
  double dpx, dpz, om0;
};
ADEV ShearAcc shear_terms(const double *g, int ndim, int k, int i, double om0, double qshear) {
  const double x1a = g[0] + i * g[1], x1b = g[0] + (i + 1) * g[1];
  const double x3a = g[4] + k * g[5], x3b = g[4] + (k + 1) * g[5];
  const int three_d = (ndim == 3);
  const double omsq = sqr(om0);
  const double dx = x1b - x1a;
  const double dz = x3b - x3a;
  const double phi_xm1 = -qshear * omsq * x1a * x1a;
  const double phi_xp1 = -qshear * omsq * x1b * x1b;
  const double phi_zm1 = 0.5 * omsq * x3a * x3a;
  const double phi_zp1 = 0.5 * omsq * x3b * x3b;
  ShearAcc s;
  s.dpx = (phi_xp1 - phi_xm1) / dx;
  s.dpz = three_d * ((phi_zp1 - phi_zm1) / dz);
  s.om0 = om0;
  return s;
}
ADEV void shear_gas(const ShearAcc &s, double dt, const FluidPrim &w, GasCons &u) {
  const double rdt = w.rho * dt;
  u.m1 -= rdt * (s.dpx - 2.0 * s.om0 * w.v2);
  u.m2 -= rdt * 2.0 * s.om0 * w.v1;
  u.m3 -= rdt * s.dpz;
  u.e -= rdt * (w.v1 * s.dpx + w.v3 * s.dpz);
}
ADEV void shear_dust(const ShearAcc &s, double dt, const FluidPrim &w, DustCons &u) {
  const double rdt = w.rho * dt;
  u.m1 -= rdt * (s.dpx - 2.0 * s.om0 * w.v2);
  u.m2 -= rdt * 2.0 * s.om0 * w.v1;
  u.m3 -= rdt * s.dpz;
}

} // namespace artemis
