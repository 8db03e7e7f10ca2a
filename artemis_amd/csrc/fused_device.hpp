// Register-level pieces shared by the streaming fused stage kernels (kernels_fused.hip: 3-D march with an LDS
// tile; kernels_stage2d.hip: 2-D row march with wave shuffles): cell / face records, PCM / PLM face values,
// the Riemann problem of one face in global momentum order.  Same expression trees as the per-task kernels.
#pragma once
#include "device_math.hpp"

namespace artemis {
namespace fused {

struct Cell6 {
  double d, v1, v2, v3, p, e;
};

// Pointers fetched from the pack's tables are generic to the compiler; the arrays live in HBM,
// so tell it: global_load/global_store instead of flat_* (no LDS-aperture check, vmcnt only).
typedef const double __attribute__((address_space(1))) *gcptr;
typedef double __attribute__((address_space(1))) *gptr;
ADEV double gld(const double *p, long c) { return ((gcptr)p)[c]; }
ADEV void gst(double *p, long c, double v) { ((gptr)p)[c] = v; }
// The same with a 32-bit element index: the byte offset c * 8 is formed in 32 bits and zero-extended, i.e. exactly the
// `global_load ... v_off, s[base:base+1]` form (uniform 64-bit base in SGPRs + one 32-bit VGPR offset shared by every
// array of the cell) -- no 64-bit address arithmetic per load.  Callers guarantee arrays below 2^29 elements (4 GiB).
typedef const char __attribute__((address_space(1))) *gcbytes;
typedef char __attribute__((address_space(1))) *gbytes;
ADEV double gld(const double *p, unsigned c) { return *(gcptr)((gcbytes)p + (c << 3)); }
ADEV void gst(double *p, unsigned c, double v) { *(gptr)((gbytes)p + (c << 3)) = v; }

// Tables the kernel never writes -- the pack's pointer tables, a block's edge table -- read INSIDE a march.  As a plain
// global load such a read is a VECTOR load (the kernel has stores in flight, so the compiler may not use the scalar
// cache) of a wave-uniform address, and the s_waitcnt vmcnt(0) behind it also waits for every prefetch of the trip.
// Through the constant address space it is a scalar load (K-cache, lgkmcnt).  `opaque` keeps the index from being
// loop-invariant to the compiler, which would otherwise hoist twenty such loads out of the march and park the results
// in scalar registers it does not have (v_writelane / v_readlane around every use).
template <class T>
ADEV T kload(const T *p) {
  return *(const __attribute__((address_space(4))) T *)(p);
}
// ... a plain-data record (size a multiple of 8 bytes) word by word
template <class T>
ADEV T kload_record(const T *p) {
  static_assert(sizeof(T) % 8 == 0, "kload_record: whole 8-byte words");
  T out;
  const unsigned long long *src = reinterpret_cast<const unsigned long long *>(p);
  unsigned long long *dst = reinterpret_cast<unsigned long long *>(&out);
#pragma unroll
  for (unsigned q = 0; q < sizeof(T) / 8; ++q) dst[q] = kload(src + q);
  return out;
}
ADEV int opaque(int i) {
  asm volatile("" : "+s"(i));
  return i;
}

template <class IDX>
ADEV Cell6 load_cell(const double *__restrict__ r, const double *__restrict__ v1,
                     const double *__restrict__ v2, const double *__restrict__ v3,
                     const double *__restrict__ se, IDX c, double gm1) {
  Cell6 q;
  q.d = gld(r, c), q.v1 = gld(v1, c), q.v2 = gld(v2, c), q.v3 = gld(v3, c), q.e = gld(se, c);
  q.p = amax(0.0, gm1 * q.d * q.e); // fill_derived.cpp:247 (IdealGas P)
  return q;
}

struct Raw5 { // a cell's five stored primitives, as loaded (pressure not yet derived)
  double d, v1, v2, v3, e;
};
ADEV Raw5 load_raw(const double *r, const double *v1, const double *v2, const double *v3,
                   const double *se, unsigned c) {
  Raw5 q;
  q.d = gld(r, c), q.v1 = gld(v1, c), q.v2 = gld(v2, c), q.v3 = gld(v3, c), q.e = gld(se, c);
  return q;
}
ADEV Cell6 finish_cell(const Raw5 &r, double gm1) {
  Cell6 q;
  q.d = r.d, q.v1 = r.v1, q.v2 = r.v2, q.v3 = r.v3, q.e = r.e;
  q.p = amax(0.0, gm1 * q.d * q.e); // fill_derived.cpp:247 (IdealGas P)
  return q;
}

template <int RECON>
ADEV double slope(double qm, double q, double qp) {
  if constexpr (RECON == 0) return 0.0;
  else return plm_dqm_fast(qm, q, qp);
}
// uniform-mesh slope with the division the caller may take (wave-uniform `fast`: no tiny velocity in reach, so the
// hand-scheduled division gives the bits of `/`)
template <int RECON>
ADEV double slope_sel(double qm, double q, double qp, bool fast) {
  if constexpr (RECON == 0) return 0.0;
  else return fast ? plm_dqm_fast(qm, q, qp) : plm_dqm(qm, q, qp);
}
// a velocity the hand-scheduled divisions cannot take: non-zero and below 2^-200 (differences of admitted values are
// then 0 or at least 2^-252, their squares and cubes normal); a momentum whose square over a density must keep its
// last place
ADEV bool tiny_vel(double v) { return v != 0.0 && fabs(v) < 0x1p-200; }
ADEV bool tiny_mom(double m) { return m != 0.0 && fabs(m) < 0x1p-480; }
// The same two tests on three components at once through the exponent (v_frexp_exp_i32_f64: 0 for zero, inf and NaN,
// the true exponent for subnormals; |v| < 2^-200 <=> frexp exponent < -199): five instructions instead of six
// compares and their mask arithmetic.
ADEV bool tiny_vel3(double a, double b, double c) {
  return min(min(__builtin_amdgcn_frexp_exp(a), __builtin_amdgcn_frexp_exp(b)), __builtin_amdgcn_frexp_exp(c)) < -199;
}
ADEV bool tiny_mom3(double a, double b, double c) {
  return min(min(__builtin_amdgcn_frexp_exp(a), __builtin_amdgcn_frexp_exp(b)), __builtin_amdgcn_frexp_exp(c)) < -479;
}
// q + 0.0 == q and q - 0.0 == q bitwise for every finite q except that -0.0 + 0.0 = +0.0;
// PCM therefore bypasses the add to stay identical to pcm.hpp:34-88.
template <int RECON>
ADEV double up_val(double q, double dqm) {
  if constexpr (RECON == 0) return q;
  else return q + dqm;
}
template <int RECON>
ADEV double lo_val(double q, double dqm) {
  if constexpr (RECON == 0) return q;
  else return q - dqm;
}

// Riemann problem of sweep direction DIR (1..3) between global-order states L and R; the
// result is returned in GLOBAL momentum order (f.m1, f.m2, f.m3).
struct Flux8 {
  double d, m1, m2, m3, e, eg, pf, vf;
};
struct GasK { // per-thread constants of the solver (hllc.hpp:75-77)
  double gm1, igm1, gamma, alpha;
};
ADEV GasK gas_constants(double gm1) {
  GasK g;
  g.gm1 = gm1, g.igm1 = 1.0 / gm1, g.gamma = gm1 + 1.0;
  g.alpha = (g.gamma + 1.0) / (2.0 * g.gamma);
  return g;
}
template <int RIEMANN, int DIR>
ADEV Flux8 solve_face(const GasK &gk, const Cell6 &L, const Cell6 &R, const bool fast = true) {
  Prim6 l, r;
  l.d = L.d, l.p = L.p, l.e = L.e, r.d = R.d, r.p = R.p, r.e = R.e;
  if constexpr (DIR == 1) {
    l.vx = L.v1, l.vy = L.v2, l.vz = L.v3, r.vx = R.v1, r.vy = R.v2, r.vz = R.v3;
  } else if constexpr (DIR == 2) { // hllc.hpp:67-69: (ivx,ivy,ivz) = (v2,v3,v1)
    l.vx = L.v2, l.vy = L.v3, l.vz = L.v1, r.vx = R.v2, r.vy = R.v3, r.vz = R.v1;
  } else { // (v3,v1,v2)
    l.vx = L.v3, l.vy = L.v1, l.vz = L.v2, r.vx = R.v3, r.vy = R.v1, r.vz = R.v2;
  }
  FaceFlux F;
  if constexpr (RIEMANN == 0) {
    // wave-uniform: hllc_gas_fast's hand-scheduled divisions are the bits of hllc_gas's only while no numerator is
    // tiny (callers without that knowledge pass true and live with DESIGN.md section 4's limit)
    if (fast) hllc_gas_fast(gk.gm1, gk.igm1, gk.gamma, gk.alpha, l, r, F);
    else hllc_gas(gk.gm1, l, r, F);
  } else if constexpr (RIEMANN == 1) { // (the same wave-uniform choice for HLLE and LLF)
    if (fast) hlle_gas_fast(gk.gm1, gk.igm1, gk.gamma, l, r, F);
    else hlle_gas(gk.gm1, l, r, F);
  } else {
    if (fast) llf_gas_fast(gk.gm1, gk.igm1, gk.gamma, l, r, F);
    else llf_gas(gk.gm1, l, r, F);
  }
  Flux8 o;
  o.d = F.fd, o.e = F.fe, o.eg = F.feg, o.pf = F.pf, o.vf = F.vf;
  if constexpr (DIR == 1) o.m1 = F.fmx, o.m2 = F.fmy, o.m3 = F.fmz;
  else if constexpr (DIR == 2) o.m2 = F.fmx, o.m3 = F.fmy, o.m1 = F.fmz;
  else o.m3 = F.fmx, o.m1 = F.fmy, o.m2 = F.fmz;
  return o;
}


} // namespace fused
} // namespace artemis
