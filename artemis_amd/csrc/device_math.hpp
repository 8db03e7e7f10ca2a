// Per-cell / per-face arithmetic of the Artemis hydro update as gfx950 device functions.
//
// Written for registers, not scratch pads: a thread carries one cell's primitive state in a
// small struct and every helper is __forceinline__.  Expression trees mirror the reference's
// (cited per function, paths relative to the reference's src/) and the library is compiled
// with -ffp-contract=off, so results are bit-identical to any other IEEE-754 evaluation of
// the same trees (fp64 division and sqrt lower to correctly rounded sequences on gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace artemis {

#define ADEV __device__ __forceinline__

// std::min / std::max semantics of the reference (first argument wins on ties / signed zero).
ADEV double amax(double a, double b) { return (a < b) ? b : a; }
ADEV double amin(double a, double b) { return (b < a) ? b : a; }
ADEV double sqr(double x) { return x * x; }

// ---- fp64 division and square root, hand-scheduled ------------------------------------------
// hipcc lowers `a / b` to  v_div_scale x2 -> v_rcp_f64 -> two Newton steps on the reciprocal ->
// q = a*y -> one Markstein correction (v_div_fmas) -> v_div_fixup.  The pre-scaling and the
// fix-up only matter when an operand or the quotient sits within ~2^100 of the fp64 exponent
// limits, is zero/inf/NaN or subnormal; everywhere else the quotient is produced by exactly
// the fma chain below, so `div` returns the same bits (checked against IEEE division on the
// device over wide-range random operands by tests/test_parity_fused.py::test_fast_div_sqrt and
// implicitly by every fused-kernel parity test).  Hydro operands here are floored densities,
// pressures, cell sizes and wave-speed differences: comfortably inside that window; a zero
// NUMERATOR is handled exactly (q = 0, r = 0).  What the split buys: a refined reciprocal is
// computed once per denominator and reused by every division that shares it (6 by the cell
// volume, 6 by the density, 2 by ml+mr, ...), and the quarter-rate v_rcp_f64 count drops by a
// third.  -ffp-contract=off does not touch explicit __builtin_fma.
struct Recip {
  double b, y; // denominator and its refined reciprocal
};
ADEV Recip recip(double b) {
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  Recip r;
  r.b = b, r.y = y;
  return r;
}
ADEV double div(double a, const Recip &r) {
  const double q = a * r.y;
  const double e = __builtin_fma(-r.b, q, a);
  return __builtin_fma(e, r.y, q);
}
ADEV double div(double a, double b) { return div(a, recip(b)); }
ADEV Recip pick(bool c, const Recip &a, const Recip &b) {
  Recip r;
  r.b = c ? a.b : b.b, r.y = c ? a.y : b.y;
  return r;
}
// Square root of a strictly positive, normal x: the Goldschmidt chain hipcc emits for sqrt()
// minus its 2^256 range scaling and its 0/inf pass-through.
ADEV double sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double s0 = x * y;
  const double h0 = y * 0.5;
  const double r0 = __builtin_fma(-h0, s0, 0.5);
  const double s1 = __builtin_fma(s0, r0, s0);
  const double h1 = __builtin_fma(h0, r0, h0);
  const double d0 = __builtin_fma(-s1, s1, x);
  const double s2 = __builtin_fma(d0, h1, s1);
  const double d1 = __builtin_fma(-s2, s2, x);
  return __builtin_fma(d1, h1, s2);
}

struct Prim6 { // gas face/cell state in sweep-local order
  double d, vx, vy, vz, p, e;
};
struct Prim4 { // dust
  double d, vx, vy, vz;
};
struct FaceFlux {
  double fd, fmx, fmy, fmz, fe, feg, pf, vf;
};

// ---- reconstruction --------------------------------------------------------------------
// utils/fluxes/reconstruction/plm.hpp:32-47: van-Leer harmonic slope; ql(i+1) = q + dqm,
// qr(i) = q - dqm.
ADEV double plm_dqm(double qm, double q, double qp) {
  const double dql = q - qm;
  const double dqr = qp - q;
  const double dq2 = dql * dqr;
  const double dqm = dq2 / (dql + dqr);
  return (dq2 <= 0.0) ? 0.0 : dqm;
}

// same slope with the hand-scheduled division (fused kernel)
ADEV double plm_dqm_fast(double qm, double q, double qp) {
  const double dql = q - qm;
  const double dqr = qp - q;
  const double dq2 = dql * dqr;
  const double dqm = div(dq2, dql + dqr);
  return (dq2 <= 0.0) ? 0.0 : dqm;
}

// utils/fluxes/reconstruction/ppm.hpp:33-66 (PPM4).  Returns ql(i+1) in `qlp`, qr(i) in `qr`.
ADEV void ppm4(double qmm, double qm, double q, double qp, double qpp, double &qlp, double &qr) {
  double qlv = (7. * (q + qm) - (qmm + qp)) / 12.0;
  double qrv = (7. * (q + qp) - (qm + qpp)) / 12.0;
  qlv = amax(qlv, amin(q, qm));
  qlv = amin(qlv, amax(q, qm));
  qrv = amax(qrv, amin(q, qp));
  qrv = amin(qrv, amax(q, qp));
  const double qc = qrv - q;
  const double qd = qlv - q;
  if ((qc * qd) >= 0.0) {
    qlv = q;
    qrv = q;
  } else {
    if (fabs(qc) >= 2.0 * fabs(qd)) qrv = q - 2.0 * qd;
    if (fabs(qd) >= 2.0 * fabs(qc)) qlv = q - 2.0 * qc;
  }
  qlp = qrv;
  qr = qlv;
}

// RECON: 0 pcm (pcm.hpp:34-88), 1 plm, 2 ppm.  `w` points at the cell, `st` = stride of the
// sweep direction.  Outputs the cell's two face values: upper (= ql of face+1) and lower
// (= qr of its own lower face).
template <int RECON>
ADEV void recon_cell(const double *__restrict__ w, long st, double &q_up, double &q_lo) {
  if constexpr (RECON == 0) {
    q_up = w[0];
    q_lo = w[0];
  } else if constexpr (RECON == 1) {
    const double q = w[0];
    const double dqm = plm_dqm(w[-st], q, w[st]);
    q_up = q + dqm;
    q_lo = q - dqm;
  } else {
    ppm4(w[-2 * st], w[-st], w[0], w[st], w[2 * st], q_up, q_lo);
  }
}

// ---- Riemann solvers --------------------------------------------------------------------
// All three write the pressure-free momentum flux, the interface pressure `pf`, the
// upwinded internal-energy flux and the face velocity vf = F_rho / rho_upwind.

// utils/fluxes/riemann/hllc.hpp:50-182
ADEV void hllc_gas(const double gm1, const Prim6 &L, const Prim6 &R, FaceFlux &F) {
  const double igm1 = 1.0 / gm1;
  const double gamma = gm1 + 1.0;
  const double alpha = (gamma + 1.0) / (2.0 * gamma);
  const double cl = sqrt(gamma * L.p / L.d);
  const double cr = sqrt(gamma * R.p / R.d);
  const double el = L.p * igm1 + 0.5 * L.d * (sqr(L.vx) + sqr(L.vy) + sqr(L.vz));
  const double er = R.p * igm1 + 0.5 * R.d * (sqr(R.vx) + sqr(R.vy) + sqr(R.vz));
  const double rc_avg = 0.25 * (L.d + R.d) * (cl + cr);
  const double pmid = 0.5 * (L.p + R.p + (L.vx - R.vx) * rc_avg);
  const double ql = (pmid <= L.p) ? 1.0 : sqrt(1.0 + alpha * ((pmid / L.p) - 1.0));
  const double qr = (pmid <= R.p) ? 1.0 : sqrt(1.0 + alpha * ((pmid / R.p) - 1.0));
  const double sl = L.vx - cl * ql;
  const double sr = R.vx + cr * qr;
  const double bp = sr > 0.0 ? sr : 1.0e-20;
  const double bm = sl < 0.0 ? sl : -1.0e-20;
  const double vxl = L.vx - sl;
  const double vxr = R.vx - sr;
  const double tl = L.p + vxl * L.d * L.vx;
  const double tr = R.p + vxr * R.d * R.vx;
  const double ml = L.d * vxl;
  const double mr = -(R.d * vxr);
  const double am = (tl - tr) / (ml + mr);
  double cp = (ml * tr + mr * tl) / (ml + mr);
  cp = cp > 0.0 ? cp : 0.0;
  const double fld = L.d * (L.vx - bm);
  const double frd = R.d * (R.vx - bp);
  const double fle = el * (L.vx - bm) + L.p * L.vx;
  const double fre = er * (R.vx - bp) + R.p * R.vx;
  double wl_, wr_, wc_;
  if (am >= 0.0) {
    wl_ = am / (am - bm);
    wr_ = 0.0;
    wc_ = -bm / (am - bm);
  } else {
    wl_ = 0.0;
    wr_ = -am / (bp - am);
    wc_ = bp / (bp - am);
  }
  F.pf = wl_ * L.p + wr_ * R.p + wc_ * cp;
  const double frho = wl_ * fld + wr_ * frd;
  F.fd = frho;
  F.fmx = wl_ * (fld * L.vx) + wr_ * (frd * R.vx);
  F.fmy = wl_ * (fld * L.vy) + wr_ * (frd * R.vy);
  F.fmz = wl_ * (fld * L.vz) + wr_ * (frd * R.vz);
  F.fe = wl_ * fle + wr_ * fre + wc_ * cp * am;
  F.feg = frho * ((frho >= 0.0) ? L.e : R.e);
  F.vf = frho / ((frho >= 0.0) ? L.d : R.d);
}

// hllc.hpp:50-182 again, same expression trees, with shared refined reciprocals: rho_l and rho_r
// serve the sound speeds and the face velocity, ml+mr serves am and cp, and the two flux
// weights share one denominator (the am >= 0 / < 0 branches select operands, not results).
ADEV void hllc_gas_fast(const double gm1, const double igm1, const double gamma, const double alpha,
                        const Prim6 &L, const Prim6 &R, FaceFlux &F) {
  const Recip rdl = recip(L.d), rdr = recip(R.d);
  const double cl = sqrt_pos(div(gamma * L.p, rdl));
  const double cr = sqrt_pos(div(gamma * R.p, rdr));
  const double el = L.p * igm1 + 0.5 * L.d * (sqr(L.vx) + sqr(L.vy) + sqr(L.vz));
  const double er = R.p * igm1 + 0.5 * R.d * (sqr(R.vx) + sqr(R.vy) + sqr(R.vz));
  const double rc_avg = 0.25 * (L.d + R.d) * (cl + cr);
  const double pmid = 0.5 * (L.p + R.p + (L.vx - R.vx) * rc_avg);
  const double ql = (pmid <= L.p) ? 1.0 : sqrt_pos(1.0 + alpha * (div(pmid, L.p) - 1.0));
  const double qr = (pmid <= R.p) ? 1.0 : sqrt_pos(1.0 + alpha * (div(pmid, R.p) - 1.0));
  const double sl = L.vx - cl * ql;
  const double sr = R.vx + cr * qr;
  const double bp = sr > 0.0 ? sr : 1.0e-20;
  const double bm = sl < 0.0 ? sl : -1.0e-20;
  const double vxl = L.vx - sl;
  const double vxr = R.vx - sr;
  const double tl = L.p + vxl * L.d * L.vx;
  const double tr = R.p + vxr * R.d * R.vx;
  const double ml = L.d * vxl;
  const double mr = -(R.d * vxr);
  const Recip rm = recip(ml + mr);
  const double am = div(tl - tr, rm);
  double cp = div(ml * tr + mr * tl, rm);
  cp = cp > 0.0 ? cp : 0.0;
  const double fld = L.d * (L.vx - bm);
  const double frd = R.d * (R.vx - bp);
  const double fle = el * (L.vx - bm) + L.p * L.vx;
  const double fre = er * (R.vx - bp) + R.p * R.vx;
  const bool pos = (am >= 0.0);
  const Recip rw = recip(pos ? (am - bm) : (bp - am));
  const double wa = div(pos ? am : -am, rw);
  const double wc_ = div(pos ? -bm : bp, rw);
  const double wl_ = pos ? wa : 0.0;
  const double wr_ = pos ? 0.0 : wa;
  F.pf = wl_ * L.p + wr_ * R.p + wc_ * cp;
  const double frho = wl_ * fld + wr_ * frd;
  F.fd = frho;
  F.fmx = wl_ * (fld * L.vx) + wr_ * (frd * R.vx);
  F.fmy = wl_ * (fld * L.vy) + wr_ * (frd * R.vy);
  F.fmz = wl_ * (fld * L.vz) + wr_ * (frd * R.vz);
  F.fe = wl_ * fle + wr_ * fre + wc_ * cp * am;
  const bool up = (frho >= 0.0);
  F.feg = frho * (up ? L.e : R.e);
  F.vf = div(frho, pick(up, rdl, rdr));
}

// utils/fluxes/riemann/hlle.hpp:56-222, gas branch
ADEV void hlle_gas(const double gm1, const Prim6 &L, const Prim6 &R, FaceFlux &F) {
  const double igm1 = 1.0 / gm1;
  const double gamma = gm1 + 1.0;
  const double sqrtdl = sqrt(L.d);
  const double sqrtdr = sqrt(R.d);
  const double isdlpdr = 1.0 / (sqrtdl + sqrtdr);
  const double ux = (sqrtdl * L.vx + sqrtdr * R.vx) * isdlpdr;
  const double uy = (sqrtdl * L.vy + sqrtdr * R.vy) * isdlpdr;
  const double uz = (sqrtdl * L.vz + sqrtdr * R.vz) * isdlpdr;
  const double el = L.p * igm1 + 0.5 * L.d * (sqr(L.vx) + sqr(L.vy) + sqr(L.vz));
  const double er = R.p * igm1 + 0.5 * R.d * (sqr(R.vx) + sqr(R.vy) + sqr(R.vz));
  const double hroe = ((el + L.p) / sqrtdl + (er + R.p) / sqrtdr) * isdlpdr;
  const double cl = sqrt(gamma * L.p / L.d);
  const double cr = sqrt(gamma * R.p / R.d);
  double a = hroe - 0.5 * (sqr(ux) + sqr(uy) + sqr(uz));
  a = (a < 0.0) ? 0.0 : sqrt(gm1 * a);
  const double sl = amin(ux - a, L.vx - cl);
  const double sr = amax(ux + a, R.vx + cr);
  const double bp = (sr > 0.0) ? sr : 1.0e-20;
  const double bm = (sl < 0.0) ? sl : -1.0e-20;
  const double ql = L.vx - bm;
  const double qr = R.vx - bp;
  const double fl_d = L.d * ql, fr_d = R.d * qr;
  const double fl_mx = L.d * L.vx * ql, fr_mx = R.d * R.vx * qr;
  const double fl_my = L.d * L.vy * ql, fr_my = R.d * R.vy * qr;
  const double fl_mz = L.d * L.vz * ql, fr_mz = R.d * R.vz * qr;
  const double fl_e = el * ql + L.p * L.vx, fr_e = er * qr + R.p * R.vx;
  double w = 0.0;
  if (bp != bm) w = 0.5 * (bp + bm) / (bp - bm);
  F.pf = 0.5 * (L.p + R.p) + w * (L.p - R.p);
  const double frho = 0.5 * (fl_d + fr_d) + w * (fl_d - fr_d);
  F.fd = frho;
  F.fmx = 0.5 * (fl_mx + fr_mx) + w * (fl_mx - fr_mx);
  F.fmy = 0.5 * (fl_my + fr_my) + w * (fl_my - fr_my);
  F.fmz = 0.5 * (fl_mz + fr_mz) + w * (fl_mz - fr_mz);
  F.fe = 0.5 * (fl_e + fr_e) + w * (fl_e - fr_e);
  F.feg = frho * ((frho >= 0.0) ? L.e : R.e);
  F.vf = frho / ((frho >= 0.0) ? L.d : R.d);
}

// hlle.hpp:56-222 (gas) again, same expression trees, with the hand-scheduled division and square root: the
// reciprocals of sqrt(rho_l) + sqrt(rho_r), sqrt(rho_l), sqrt(rho_r), rho_l, rho_r and bp - bm are refined once and
// every quotient that shares one takes three fmas (`div` returns the bits of `/` for these operands: densities are
// floored, sound speeds positive, bp - bm >= 2e-20).  The one square root whose argument can vanish -- gm1 * a, a the
// Roe enthalpy minus the kinetic energy -- takes the IEEE sqrt unless the whole wave holds normal positive values.
// Callers pass through here only where hllc_gas_fast would be admissible too (no tiny-but-nonzero velocity in reach).
ADEV void hlle_gas_fast(const double gm1, const double igm1, const double gamma, const Prim6 &L, const Prim6 &R,
                        FaceFlux &F) {
  const double sqrtdl = sqrt_pos(L.d);
  const double sqrtdr = sqrt_pos(R.d);
  const double isdlpdr = div(1.0, recip(sqrtdl + sqrtdr));
  const double ux = (sqrtdl * L.vx + sqrtdr * R.vx) * isdlpdr;
  const double uy = (sqrtdl * L.vy + sqrtdr * R.vy) * isdlpdr;
  const double uz = (sqrtdl * L.vz + sqrtdr * R.vz) * isdlpdr;
  const double el = L.p * igm1 + 0.5 * L.d * (sqr(L.vx) + sqr(L.vy) + sqr(L.vz));
  const double er = R.p * igm1 + 0.5 * R.d * (sqr(R.vx) + sqr(R.vy) + sqr(R.vz));
  const double hroe = (div(el + L.p, recip(sqrtdl)) + div(er + R.p, recip(sqrtdr))) * isdlpdr;
  const Recip rdl = recip(L.d), rdr = recip(R.d);
  const double cl = sqrt_pos(div(gamma * L.p, rdl));
  const double cr = sqrt_pos(div(gamma * R.p, rdr));
  double a = hroe - 0.5 * (sqr(ux) + sqr(uy) + sqr(uz));
  {
    const double x = gm1 * a;
    // (a < 0 gives 0 whatever the root; a NaN or a vanishing argument sends the wave through sqrt())
    const bool plain = (a >= 0.0) && !(x > 0x1p-900);
    double r;
    if (__builtin_amdgcn_ballot_w64(plain) != 0 || __builtin_amdgcn_ballot_w64(a != a) != 0) r = sqrt(x);
    else r = sqrt_pos(x);
    a = (a < 0.0) ? 0.0 : r;
  }
  const double sl = amin(ux - a, L.vx - cl);
  const double sr = amax(ux + a, R.vx + cr);
  const double bp = (sr > 0.0) ? sr : 1.0e-20;
  const double bm = (sl < 0.0) ? sl : -1.0e-20;
  const double ql = L.vx - bm;
  const double qr = R.vx - bp;
  const double fl_d = L.d * ql, fr_d = R.d * qr;
  const double fl_mx = L.d * L.vx * ql, fr_mx = R.d * R.vx * qr;
  const double fl_my = L.d * L.vy * ql, fr_my = R.d * R.vy * qr;
  const double fl_mz = L.d * L.vz * ql, fr_mz = R.d * R.vz * qr;
  const double fl_e = el * ql + L.p * L.vx, fr_e = er * qr + R.p * R.vx;
  const double w = (bp != bm) ? div(0.5 * (bp + bm), recip(bp - bm)) : 0.0;
  F.pf = 0.5 * (L.p + R.p) + w * (L.p - R.p);
  const double frho = 0.5 * (fl_d + fr_d) + w * (fl_d - fr_d);
  F.fd = frho;
  F.fmx = 0.5 * (fl_mx + fr_mx) + w * (fl_mx - fr_mx);
  F.fmy = 0.5 * (fl_my + fr_my) + w * (fl_my - fr_my);
  F.fmz = 0.5 * (fl_mz + fr_mz) + w * (fl_mz - fr_mz);
  F.fe = 0.5 * (fl_e + fr_e) + w * (fl_e - fr_e);
  const bool up = (frho >= 0.0);
  F.feg = frho * (up ? L.e : R.e);
  F.vf = div(frho, pick(up, rdl, rdr));
}

// utils/fluxes/riemann/llf.hpp:47-170, gas branch
ADEV void llf_gas(const double gm1, const Prim6 &L, const Prim6 &R, FaceFlux &F) {
  const double igm1 = 1.0 / gm1;
  const double gamma = gm1 + 1.0;
  const double ml = L.d * L.vx;
  const double mr = R.d * R.vx;
  const double fsum_d = ml + mr;
  const double fsum_mx = ml * L.vx + mr * R.vx;
  const double fsum_my = ml * L.vy + mr * R.vy;
  const double fsum_mz = ml * L.vz + mr * R.vz;
  const double el = L.p * igm1 + 0.5 * L.d * (sqr(L.vx) + sqr(L.vy) + sqr(L.vz));
  const double er = R.p * igm1 + 0.5 * R.d * (sqr(R.vx) + sqr(R.vy) + sqr(R.vz));
  const double fsum_e = (el + L.p) * L.vx + (er + R.p) * R.vx;
  const double cl = sqrt(gamma * L.p / L.d);
  const double cr = sqrt(gamma * R.p / R.d);
  const double a = amax((fabs(L.vx) + cl), (fabs(R.vx) + cr));
  const double du_d = a * (R.d - L.d);
  const double du_mx = a * (R.d * R.vx - L.d * L.vx);
  const double du_my = a * (R.d * R.vy - L.d * L.vy);
  const double du_mz = a * (R.d * R.vz - L.d * L.vz);
  const double du_e = a * (er - el);
  F.pf = 0.5 * (L.p + R.p);
  const double frho = 0.5 * (fsum_d - du_d);
  F.fd = frho;
  F.fmx = 0.5 * (fsum_mx - du_mx);
  F.fmy = 0.5 * (fsum_my - du_my);
  F.fmz = 0.5 * (fsum_mz - du_mz);
  F.fe = 0.5 * (fsum_e - du_e);
  F.feg = frho * ((frho >= 0.0) ? L.e : R.e);
  F.vf = frho / ((frho >= 0.0) ? L.d : R.d);
}

// llf.hpp:47-170 (gas) with the hand-scheduled division and square root (rho_l, rho_r refined once: sound speeds and
// the face velocity)
ADEV void llf_gas_fast(const double gm1, const double igm1, const double gamma, const Prim6 &L, const Prim6 &R,
                       FaceFlux &F) {
  const double ml = L.d * L.vx;
  const double mr = R.d * R.vx;
  const double fsum_d = ml + mr;
  const double fsum_mx = ml * L.vx + mr * R.vx;
  const double fsum_my = ml * L.vy + mr * R.vy;
  const double fsum_mz = ml * L.vz + mr * R.vz;
  const double el = L.p * igm1 + 0.5 * L.d * (sqr(L.vx) + sqr(L.vy) + sqr(L.vz));
  const double er = R.p * igm1 + 0.5 * R.d * (sqr(R.vx) + sqr(R.vy) + sqr(R.vz));
  const double fsum_e = (el + L.p) * L.vx + (er + R.p) * R.vx;
  const Recip rdl = recip(L.d), rdr = recip(R.d);
  const double cl = sqrt_pos(div(gamma * L.p, rdl));
  const double cr = sqrt_pos(div(gamma * R.p, rdr));
  const double a = amax((fabs(L.vx) + cl), (fabs(R.vx) + cr));
  const double du_d = a * (R.d - L.d);
  const double du_mx = a * (R.d * R.vx - L.d * L.vx);
  const double du_my = a * (R.d * R.vy - L.d * L.vy);
  const double du_mz = a * (R.d * R.vz - L.d * L.vz);
  const double du_e = a * (er - el);
  F.pf = 0.5 * (L.p + R.p);
  const double frho = 0.5 * (fsum_d - du_d);
  F.fd = frho;
  F.fmx = 0.5 * (fsum_mx - du_mx);
  F.fmy = 0.5 * (fsum_my - du_my);
  F.fmz = 0.5 * (fsum_mz - du_mz);
  F.fe = 0.5 * (fsum_e - du_e);
  const bool up = (frho >= 0.0);
  F.feg = frho * (up ? L.e : R.e);
  F.vf = div(frho, pick(up, rdl, rdr));
}

template <int RIEMANN>
ADEV void riemann_gas(const double gm1, const Prim6 &L, const Prim6 &R, FaceFlux &F) {
  if constexpr (RIEMANN == 0) hllc_gas(gm1, L, R, F);
  else if constexpr (RIEMANN == 1) hlle_gas(gm1, L, R, F);
  else llf_gas(gm1, L, R, F);
}

// utils/fluxes/riemann/hlle.hpp:56-222, dust branch (pressureless: Roe velocity bounds)
ADEV void hlle_dust(const Prim4 &L, const Prim4 &R, FaceFlux &F) {
  const double sqrtdl = sqrt(L.d);
  const double sqrtdr = sqrt(R.d);
  const double isdlpdr = 1.0 / (sqrtdl + sqrtdr);
  const double ux = (sqrtdl * L.vx + sqrtdr * R.vx) * isdlpdr;
  const double sl = amin(ux, L.vx);
  const double sr = amax(ux, R.vx);
  const double bp = (sr > 0.0) ? sr : 1.0e-20;
  const double bm = (sl < 0.0) ? sl : -1.0e-20;
  const double ql = L.vx - bm;
  const double qr = R.vx - bp;
  const double fl_d = L.d * ql, fr_d = R.d * qr;
  const double fl_mx = L.d * L.vx * ql, fr_mx = R.d * R.vx * qr;
  const double fl_my = L.d * L.vy * ql, fr_my = R.d * R.vy * qr;
  const double fl_mz = L.d * L.vz * ql, fr_mz = R.d * R.vz * qr;
  double w = 0.0;
  if (bp != bm) w = 0.5 * (bp + bm) / (bp - bm);
  F.fd = 0.5 * (fl_d + fr_d) + w * (fl_d - fr_d);
  F.fmx = 0.5 * (fl_mx + fr_mx) + w * (fl_mx - fr_mx);
  F.fmy = 0.5 * (fl_my + fr_my) + w * (fl_my - fr_my);
  F.fmz = 0.5 * (fl_mz + fr_mz) + w * (fl_mz - fr_mz);
}

// hlle.hpp:56-222 (dust) with the hand-scheduled division and square root (dust densities are floored)
ADEV void hlle_dust_fast(const Prim4 &L, const Prim4 &R, FaceFlux &F) {
  const double sqrtdl = sqrt_pos(L.d);
  const double sqrtdr = sqrt_pos(R.d);
  const double isdlpdr = div(1.0, recip(sqrtdl + sqrtdr));
  const double ux = (sqrtdl * L.vx + sqrtdr * R.vx) * isdlpdr;
  const double sl = amin(ux, L.vx);
  const double sr = amax(ux, R.vx);
  const double bp = (sr > 0.0) ? sr : 1.0e-20;
  const double bm = (sl < 0.0) ? sl : -1.0e-20;
  const double ql = L.vx - bm;
  const double qr = R.vx - bp;
  const double fl_d = L.d * ql, fr_d = R.d * qr;
  const double fl_mx = L.d * L.vx * ql, fr_mx = R.d * R.vx * qr;
  const double fl_my = L.d * L.vy * ql, fr_my = R.d * R.vy * qr;
  const double fl_mz = L.d * L.vz * ql, fr_mz = R.d * R.vz * qr;
  const double w = (bp != bm) ? div(0.5 * (bp + bm), recip(bp - bm)) : 0.0;
  F.fd = 0.5 * (fl_d + fr_d) + w * (fl_d - fr_d);
  F.fmx = 0.5 * (fl_mx + fr_mx) + w * (fl_mx - fr_mx);
  F.fmy = 0.5 * (fl_my + fr_my) + w * (fl_my - fr_my);
  F.fmz = 0.5 * (fl_mz + fr_mz) + w * (fl_mz - fr_mz);
}

// utils/fluxes/riemann/llf.hpp:47-170, dust branch
ADEV void llf_dust(const Prim4 &L, const Prim4 &R, FaceFlux &F) {
  const double ml = L.d * L.vx;
  const double mr = R.d * R.vx;
  const double fsum_d = ml + mr;
  const double fsum_mx = ml * L.vx + mr * R.vx;
  const double fsum_my = ml * L.vy + mr * R.vy;
  const double fsum_mz = ml * L.vz + mr * R.vz;
  const double a = amax(fabs(L.vx), fabs(R.vx));
  const double du_d = a * (R.d - L.d);
  const double du_mx = a * (R.d * R.vx - L.d * L.vx);
  const double du_my = a * (R.d * R.vy - L.d * L.vy);
  const double du_mz = a * (R.d * R.vz - L.d * L.vz);
  F.fd = 0.5 * (fsum_d - du_d);
  F.fmx = 0.5 * (fsum_mx - du_mx);
  F.fmy = 0.5 * (fsum_my - du_my);
  F.fmz = 0.5 * (fsum_mz - du_mz);
}

template <int RIEMANN>
ADEV void riemann_dust(const Prim4 &L, const Prim4 &R, FaceFlux &F) {
  if constexpr (RIEMANN == 1) hlle_dust(L, R, F);
  else llf_dust(L, R, F);
}

// ---- Cartesian cell geometry (geometry/geometry.hpp:65-72,199-225) -----------------------
// Xf(idx) = f0 + idx*dx is recomputed per cell exactly as the reference's BBox does, so cell
// widths carry the same last-bit wobble.
struct CellGeom {
  double dx1, dx2, dx3;
};
ADEV CellGeom cell_geom(const double *__restrict__ g, int k, int j, int i) {
  CellGeom c;
  c.dx1 = (g[0] + (i + 1) * g[1]) - (g[0] + i * g[1]);
  c.dx2 = (g[2] + (j + 1) * g[3]) - (g[2] + j * g[3]);
  c.dx3 = (g[4] + (k + 1) * g[5]) - (g[4] + k * g[5]);
  return c;
}

} // namespace artemis
