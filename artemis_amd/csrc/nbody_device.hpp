// Gravity::NBodyGravity<GEOM> (gravity/nbody_gravity.hpp:28-221) per zone, with the particle functions of
// nbody/particle_base.hpp:96-258: what the N-body task adds to one fluid of one zone, particle by particle in the
// reference's order.  Shared by the task kernels (kernels_sources.hip) and by the one-kernel stages, which apply it in
// the task's slot of the stage (after DiffusionUpdate, before RotatingFrameForce: artemis_driver.cpp:218-236) on the
// conserved state they hold in registers.  The seven back-reaction sums per particle are formed by the task kernel's
// force-only instantiation (artemis_hip_nbody_force_sums): they depend on the stage's input primitives alone.
#pragma once
#include "device_math.hpp"
#include "fused_device.hpp"
#include "geometry.hpp"
#include "sources_device.hpp"

namespace artemis {

ADEV double nb_idr3(const artemis_nbody_particle_t &p, const double dr2) { // particle_base.hpp:146-166
  const double fuzz = 1e-99;
  const double rs2 = sqr(p.rs);
  // Plummer softening away from the particle: the spline expression is finite (its quotients have denominators above
  // 1e-300) and enters as 0 * finite = +-0 added to a positive number -- the Plummer term alone, same bits, one
  // division and one square root instead of four and three
  if (p.spline == 0 && dr2 > 1e-200) return 1.0 / (fuzz + sqrt(dr2 + rs2) * (dr2 + rs2)) * 1.0;
  const double idr3_p = 1.0 / (fuzz + sqrt(dr2 + rs2) * (dr2 + rs2));
  const double dr3 = dr2 * sqrt(dr2);
  const double u2 = dr2 / (rs2 + fuzz);
  const double u = sqrt(u2);
  const double u3 = u * u2;
  const double h3inv = 1. / (rs2 * p.rs + fuzz);
  const double idr3_s = (dr2 >= rs2) ? 1.0 / dr3
                                     : ((u < 0.5) ? h3inv * (32.0 / 3.0 - 192.0 / 5.0 * u2 + 32.0 * u3)
                                                  : h3inv * (64.0 / 3.0 - 48.0 * u + 192.0 / 5.0 * u2 - 32.0 / 3.0 * u3 -
                                                             1.0 / (15.0 * u3)));
  return idr3_p * (1 - p.spline) + p.spline * idr3_s;
}
ADEV void nb_accrete(const artemis_nbody_particle_t &p, const double x[3], const double den, const double v[3],
                     const double vb[3], const double dt, double &dm, double dmom[3], double &dEk) { // :190-245
  const double fuzz = 1e-99;
  const double vrel[3] = {v[0] + vb[0], v[1] + vb[1], v[2] + vb[2]};
  double dx[3], dv[3];
  for (int d = 0; d < 3; d++) dx[d] = x[d] - (p.pos[d] - p.xf[d]), dv[d] = vrel[d] - (p.vel[d] - p.vf[d]);
  const double dv2 = sqr(dv[0]) + sqr(dv[1]) + sqr(dv[2]);
  // Outside the accretion radius (every zone of a particle with racc <= 0, nearly every zone otherwise) `acc` is
  // false and gdt = bdt = +0, fm = -0.0, fp = +0.0: dm and dmom receive +-0 (unchanged), denp = den * (1 + -0.0) = den,
  // and what is left is dEk += 0.5 (v + vxp) den (vxp - v) with vxp = (den v) / den -- not always v in floating point,
  // so the three divisions stay; the other seven, the unit vectors and the ramp are skipped.  (Velocities and
  // positions are finite, so the skipped products are 0 * finite.)  Same bits as the full expression below.
  {
    bool acc_ = false;
    if (p.racc > 0.0) {
      const double R_ = sqrt(sqr(dx[0]) + sqr(dx[1]));
      const double r_ = sqrt(sqr(R_) + sqr(dx[2]));
      acc_ = (r_ <= p.racc) && (-p.gm / (r_ + fuzz) + 0.5 * dv2 <= 0.0);
    }
    if (!acc_) {
      for (int i = 0; i < 3; i++) {
        const double vxp = (den * v[i]) / den;
        dEk += 0.5 * (v[i] + vxp) * den * (vxp - v[i]);
      }
      return;
    }
  }
  const double R = sqrt(sqr(dx[0]) + sqr(dx[1]));
  const double r = sqrt(sqr(R) + sqr(dx[2]));
  const double ct = dx[2] / (r + fuzz), st = R / (r + fuzz);
  const double cp = dx[0] / (R + fuzz), sp = dx[1] / (R + fuzz);
  // particle_base.hpp:201 binds [dr, er, et, ep] to CartToSph's {xout, ex1, ex2, ex3} (:255-257): et / ep are the
  // second / third ROWS as written there, not the textbook unit vectors -- kept as the reference has it
  const double et[3] = {st * sp, ct * sp, cp}, ep[3] = {ct, -st, 0.0};
  const double dvt = dv[0] * et[0] + dv[1] * et[1] + dv[2] * et[2];
  const double dvp = dv[0] * ep[0] + dv[1] * ep[1] + dv[2] * ep[2];
  const bool acc = ((p.racc > 0.0) && (r <= p.racc) && (-p.gm / (r + fuzz) + 0.5 * dv2 <= 0.0));
  const double ramp = sqr((p.racc - r) / (p.racc + fuzz));
  const double gdt = acc * amin(ramp * p.gamma * dt, 1.0 / 9.0);
  const double bdt = acc * amin(ramp * p.beta * dt, 1.0 / 9.0);
  const double fm = -gdt / (1.0 + gdt);
  dm += den * fm;
  const double fp = (gdt - bdt) / ((1.0 + gdt) * (1.0 + bdt));
  const double denp = den * (1.0 + fm);
  for (int i = 0; i < 3; i++) {
    const double dmv = den * (fm * v[i] + fp * (dvt * et[i] + dvp * ep[i]));
    dmom[i] += dmv;
    const double vxp = (den * v[i] + dmv) / denp;
    dEk += 0.5 * (v[i] + vxp) * den * (vxp - v[i]) + 0.5 * den * fm * vxp * vxp;
  }
}

// What the task needs of a zone, whatever the particle: Cartesian position and unit vectors of its centre
// (nbody_gravity.hpp:66-72), scale factors, volume, the frame's velocity there in Cartesian components (:74-83)
struct NbZone {
  Frame fr;
  double hx[3], vol, vf[3];
};
template <class CO>
ADEV NbZone nb_zone(const CO &co, double omf) {
  NbZone z;
  double x[3];
  co.centre(x);
  const bool cyl = (co.sys == ARTEMIS_CYLINDRICAL);
  z.fr = cart_frame(co.sys, x, co.cv, co.sv, cyl ? co.cv : co.c3, cyl ? co.sv : co.s3);
  scale_factors_of(co, z.hx);
  z.vol = co.volume();
  z.vf[0] = z.vf[1] = z.vf[2] = 0.0;
  if (omf != 0.0) {
    double vrot[3];
    rotation_velocity(co, omf, vrot);
    z.vf[0] = z.fr.e1[0] * vrot[0] + z.fr.e2[0] * vrot[1] + z.fr.e3[0] * vrot[2];
    z.vf[1] = z.fr.e1[1] * vrot[0] + z.fr.e2[1] * vrot[1] + z.fr.e3[1] * vrot[2];
    z.vf[2] = z.fr.e1[2] * vrot[0] + z.fr.e2[2] * vrot[1] + z.fr.e3[2] * vrot[2];
  }
  return z;
}
ADEV void nb_cart_velocity(const Frame &fr, const double w[4], double vc[3]) { // w = rho, v1, v2, v3
  vc[0] = fr.e1[0] * w[1] + fr.e2[0] * w[2] + fr.e3[0] * w[3];
  vc[1] = fr.e1[1] * w[1] + fr.e2[1] * w[2] + fr.e3[1] * w[3];
  vc[2] = fr.e1[2] * w[1] + fr.e2[2] * w[2] + fr.e3[2] * w[3];
}
// One particle's pull at the zone: g (Cartesian) and its components along the zone's unit vectors (:95-117)
struct NbPull {
  double g[3], gx[3];
};
ADEV NbPull nb_pull(const artemis_nbody_particle_t &pl, const Frame &fr) {
  NbPull q;
  q.g[0] = q.g[1] = q.g[2] = 0.0;
  double dxp[3];
  for (int d = 0; d < 3; d++) dxp[d] = fr.x[d] - (pl.pos[d] - pl.xf[d]);
  const double dr2 = sqr(dxp[0]) + sqr(dxp[1]) + sqr(dxp[2]);
  const double idr3_ = nb_idr3(pl, dr2);
  for (int d = 0; d < 3; d++) q.g[d] += -pl.gm * idr3_ * dxp[d];
  q.gx[0] = q.g[0] * fr.e1[0] + q.g[1] * fr.e1[1] + q.g[2] * fr.e1[2];
  q.gx[1] = q.g[0] * fr.e2[0] + q.g[1] * fr.e2[1] + q.g[2] * fr.e2[2];
  q.gx[2] = q.g[0] * fr.e3[0] + q.g[1] * fr.e3[1] + q.g[2] * fr.e3[2];
  return q;
}
// One particle on one fluid of the zone (:119-215): w = the stage's input primitives (rho, v1, v2, v3), vcart their
// Cartesian velocity, u = conserved (d, m1, m2, m3[, e, eg]) -- updated when APPLY --, f7 = the particle's seven sums
// (mass rate, gravitational force, accreted momentum rate) -- updated when FORCE.
template <bool GAS, bool APPLY, bool FORCE>
ADEV void nb_fluid(const artemis_nbody_particle_t &pl, const NbZone &z, const NbPull &q, double dt, const double *w,
                   const double *vcart, double *u, double *f7) {
  const double dens = w[0];
  double dm = 0.0, dmom[3] = {0.0, 0.0, 0.0}, dek = 0.0;
  const double dei = 0.0;
  // (the sums alone: a particle without an accretion radius leaves dm and dmom at exactly zero -- particle_base.hpp:213,
  //  `acc` is false -- and the kinetic-energy term is not part of the sums)
  if (APPLY || pl.racc > 0.0) nb_accrete(pl, z.fr.x, dens, vcart, z.vf, dt, dm, dmom, dek);
  if constexpr (APPLY) {
    const Frame &fr = z.fr;
    const double dmx1 = dmom[0] * fr.e1[0] + dmom[1] * fr.e1[1] + dmom[2] * fr.e1[2];
    const double dmx2 = dmom[0] * fr.e2[0] + dmom[1] * fr.e2[1] + dmom[2] * fr.e2[2];
    const double dmx3 = dmom[0] * fr.e3[0] + dmom[1] * fr.e3[1] + dmom[2] * fr.e3[2];
    const double rdt = dens * dt;
    u[0] += dm;
    u[1] += z.hx[0] * (rdt * q.gx[0] + dmx1);
    u[2] += z.hx[1] * (rdt * q.gx[1] + dmx2);
    u[3] += z.hx[2] * (rdt * q.gx[2] + dmx3);
    if constexpr (GAS) {
      u[4] += dek + dei + rdt * (w[1] * q.gx[0] + w[2] * q.gx[1] + w[3] * q.gx[2]);
      u[5] += dei;
    }
  }
  if constexpr (FORCE) {
    // (a particle without an accretion radius in the sums-only form: dm and dmom are exactly +0, the four quotients are +-0
    //  and leave sums that started at +0 as they are -- four fp64 divisions per fluid, particle and zone not made)
    if (APPLY || pl.racc > 0.0) {
      f7[0] -= z.vol * dm / dt;
      f7[4] -= dmom[0] / dt;
      f7[5] -= dmom[1] / dt;
      f7[6] -= dmom[2] / dt;
    }
    f7[1] -= q.g[0] * dens * z.vol;
    f7[2] -= q.g[1] * dens * z.vol;
    f7[3] -= q.g[2] * dens * z.vol;
  }
}
// Every coupled particle in order on one fluid of a zone held in registers (the one-kernel stages)
template <bool GAS, class CO>
ADEV void nb_apply(const artemis_nbody_particle_t *pl, int npart, const CO &co, double omf, double dt, const double w[4],
                   double *u) {
  const NbZone z = nb_zone(co, omf);
  double vc[3];
  nb_cart_velocity(z.fr, w, vc);
#pragma unroll 1
  for (int n = 0; n < npart; ++n) {
    const artemis_nbody_particle_t p = fused::kload_record(pl + n); // (never written by a kernel: a scalar load also behind stores)
    if (!p.couple) continue;
    const NbPull q = nb_pull(p, z.fr);
    nb_fluid<GAS, true, false>(p, z, q, dt, w, vc, u, nullptr);
  }
}

} // namespace artemis
