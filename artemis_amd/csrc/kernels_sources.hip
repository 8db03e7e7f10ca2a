// Pointwise source-term tasks that sit between FluxSource and SetAuxillaryFields in the
// reference's stage (artemis_driver.cpp:222-241): ExternalGravity, RotatingFrameForce,
// DragSource.  Each reads primitives / conserved variables of one cell and updates the
// conserved variables of the same cell: pure streaming kernels, thread x walks i.
#include <cfloat>
#include <type_traits>

#include <cstdlib>

#include "device_math.hpp"
#include "diffusion_device.hpp"
#include "geometry.hpp"
#include "kernels.hpp"
#include "options.hpp"
#include "nbody_device.hpp"
#include "pack_view.hpp"
#include "sources_device.hpp"

namespace artemis {
namespace {
constexpr int TX = 64, TY = 4;

#define INTERIOR_CELL                                                                      \
  const int i = P.is + blockIdx.x * blockDim.x + threadIdx.x;                              \
  const int j = P.js + blockIdx.y * blockDim.y + threadIdx.y;                              \
  const int nkr = P.ke - P.ks + 1;                                                         \
  const int b = blockIdx.z / nkr;                                                          \
  const int k = P.ks + blockIdx.z % nkr;                                                   \
  if (i > P.ie || j > P.je) return;                                                        \
  const long c = (static_cast<long>(k) * P.nj + j) * P.ni + i;

// Thread shape of the one-thread-per-zone kernels: 64 x 4 by default; for narrow mesh blocks (refined meshes
// run 16^3 blocks) the x1 extent of the workgroup shrinks to the next power of two >= nx and the rows it
// frees fold along x2, so that a wave's 64 lanes stay on real zones (a 16-zone row filled a quarter of them).
inline dim3 tile_threads(int nx) {
  int tx = TX;
  while (tx > 8 && tx / 2 >= nx) tx >>= 1;
  return dim3(tx, TX * TY / tx);
}
inline dim3 interior_threads(const PackView &P) { return tile_threads(P.ie - P.is + 1); }
inline dim3 interior_grid(const PackView &P) {
  const dim3 t = interior_threads(P);
  return dim3((P.ie - P.is + t.x) / t.x, (P.je - P.js + t.y) / t.y, (P.ke - P.ks + 1) * P.nb);
}


// (Coords<GEOM>::ConvertToCylWithVec: sources_device.hpp to_cyl_with_vec)

// ---------------------------------------------------------------------------------------
// Gravity::ExternalGravity (gravity.cpp:126-155): UniformGravity (uniform.cpp:28-84) and
// PointMassGravity (point_mass.cpp:27-198; Cartesian, spherical1D/2D, axisymmetric).
ADEV FluidPrim load_prim(const FluidView &f, int nv, int b, int n, long c, bool gas) {
  FluidPrim w;
  const int ns = f.ns;
  w.rho = f.prim[b * nv + n][c];
  w.v1 = f.prim[b * nv + ns + 3 * n + 0][c];
  w.v2 = f.prim[b * nv + ns + 3 * n + 1][c];
  w.v3 = f.prim[b * nv + ns + 3 * n + 2][c];
  w.sie = gas ? f.prim[b * nv + 5 * ns + n][c] : 0.0;
  return w;
}
ADEV GasCons load_gas_cons(const FluidView &f, int b, int n, long c) {
  const int ns = f.ns, nv = 6 * ns;
  GasCons u;
  u.d = f.cons0[b * nv + n][c];
  u.m1 = f.cons0[b * nv + ns + 3 * n + 0][c], u.m2 = f.cons0[b * nv + ns + 3 * n + 1][c];
  u.m3 = f.cons0[b * nv + ns + 3 * n + 2][c];
  u.e = f.cons0[b * nv + 4 * ns + n][c], u.eg = f.cons0[b * nv + 5 * ns + n][c];
  return u;
}
ADEV void store_gas_cons(const FluidView &f, int b, int n, long c, const GasCons &u, bool with_d) {
  const int ns = f.ns, nv = 6 * ns;
  if (with_d) f.cons0[b * nv + n][c] = u.d;
  f.cons0[b * nv + ns + 3 * n + 0][c] = u.m1, f.cons0[b * nv + ns + 3 * n + 1][c] = u.m2;
  f.cons0[b * nv + ns + 3 * n + 2][c] = u.m3, f.cons0[b * nv + 4 * ns + n][c] = u.e;
}
ADEV DustCons load_dust_cons(const FluidView &f, int b, int n, long c) {
  const int ns = f.ns, nv = 4 * ns;
  DustCons u;
  u.d = f.cons0[b * nv + n][c];
  u.m1 = f.cons0[b * nv + ns + 3 * n + 0][c], u.m2 = f.cons0[b * nv + ns + 3 * n + 1][c];
  u.m3 = f.cons0[b * nv + ns + 3 * n + 2][c];
  return u;
}
ADEV void store_dust_cons(const FluidView &f, int b, int n, long c, const DustCons &u, bool with_d) {
  const int ns = f.ns, nv = 4 * ns;
  if (with_d) f.cons0[b * nv + n][c] = u.d;
  f.cons0[b * nv + ns + 3 * n + 0][c] = u.m1, f.cons0[b * nv + ns + 3 * n + 1][c] = u.m2;
  f.cons0[b * nv + ns + 3 * n + 2][c] = u.m3;
}

__global__ __launch_bounds__(TX *TY) void gravity_kernel(const PackView P, const artemis_gravity_t G,
                                                         double dt) {
  INTERIOR_CELL
  const DCoords co = make_coords(P, b, k, j, i);
  double hx[3];
  scale_factors_of(co, hx);
  const GravAcc a = gravity_accel(G, co, P.ndim, dt);
  for (int n = 0; n < P.gas.ns; ++n) {
    GasCons u = load_gas_cons(P.gas, b, n, c);
    gravity_gas(a, dt, hx, load_prim(P.gas, 6 * P.gas.ns, b, n, c, true), u);
    store_gas_cons(P.gas, b, n, c, u, !a.uniform);
  }
  for (int n = 0; n < P.dust.ns; ++n) {
    DustCons u = load_dust_cons(P.dust, b, n, c);
    gravity_dust(a, dt, hx, load_prim(P.dust, 4 * P.dust.ns, b, n, c, false), u);
    store_dust_cons(P.dust, b, n, c, u, !a.uniform);
  }
}

// ---------------------------------------------------------------------------------------
// Gravity::NBodyGravity<GEOM> (gravity/nbody_gravity.hpp:28-221) with the particle functions of
// nbody/particle_base.hpp:96-258.  A fixed grid of workgroups strides over the zones; particle by particle
// (the reference's order: the conserved state accumulates in that order) each thread updates its zones and
// sums the seven back-reaction terms, the workgroup reduces them in a fixed tree and stores one partial row.
struct NBodyView {
  const artemis_nbody_particle_t *pl; // device
  int npart;
  double omf, dt;
  const double *dt_ptr; // optional device scalar replacing dt (one-kernel kernel only)
  double *partial;      // [npart][gridDim.x][7]
};
// Zone-outer, particle-inner: the geometry of a zone (coordinates, Cartesian frame, scale factors, volume, frame
// velocity) is formed once and the mesh is walked once, whatever the number of particles; at a zone the additions
// happen in particle order, as before (round 2 walked the mesh once per particle: 5.8 ms per stage on the 29 M-zone
// configs[4] mesh, the most expensive kernel of its stage).  The seven running sums of up to NB_CHUNK particles stay
// in registers (predicated adds: the particle loop is not unrolled); more particles take more passes.
constexpr int NB_CHUNK = 4;
__global__ __launch_bounds__(256) void nbody_gravity_kernel(const PackView P, const NBodyView N) {
  __shared__ double red[4][7];
  __shared__ artemis_nbody_particle_t spl[NB_CHUNK];
  const int nx1 = P.ie - P.is + 1, nx2 = P.je - P.js + 1, nx3 = P.ke - P.ks + 1;
  const long per_block = static_cast<long>(nx1) * nx2 * nx3, total = per_block * P.nb;
  for (int np0 = 0; np0 < N.npart; np0 += NB_CHUNK) {
    const int nloc = (N.npart - np0 < NB_CHUNK) ? N.npart - np0 : NB_CHUNK;
    __syncthreads();
    if (threadIdx.x < nloc) spl[threadIdx.x] = N.pl[np0 + threadIdx.x];
    __syncthreads();
    double lf[NB_CHUNK][7];
#pragma unroll
    for (int q = 0; q < NB_CHUNK; ++q)
#pragma unroll
      for (int m = 0; m < 7; ++m) lf[q][m] = 0.0;
    for (long t = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
         t += static_cast<long>(gridDim.x) * blockDim.x) {
      const int b = static_cast<int>(t / per_block);
      const long r = t - b * per_block;
      const int i = P.is + static_cast<int>(r % nx1), j = P.js + static_cast<int>((r / nx1) % nx2);
      const int k = P.ks + static_cast<int>(r / (static_cast<long>(nx1) * nx2));
      const long c = static_cast<long>(k) * P.sk + static_cast<long>(j) * P.sj + i;
      const DCoords co = make_coords(P, b, k, j, i);
      double x[3];
      co.centre(x);
      const bool cyl = (co.sys == ARTEMIS_CYLINDRICAL);
      const Frame fr = cart_frame(co.sys, x, co.cv, co.sv, cyl ? co.cv : co.c3, cyl ? co.sv : co.s3);
      double hx[3];
      scale_factors_of(co, hx);
      const double vol = co.volume();
      double vf[3] = {0.0, 0.0, 0.0};
      if (N.omf != 0.0) {
        double vrot[3];
        rotation_velocity(co, N.omf, vrot);
        vf[0] = fr.e1[0] * vrot[0] + fr.e2[0] * vrot[1] + fr.e3[0] * vrot[2];
        vf[1] = fr.e1[1] * vrot[0] + fr.e2[1] * vrot[1] + fr.e3[1] * vrot[2];
        vf[2] = fr.e1[2] * vrot[0] + fr.e2[2] * vrot[1] + fr.e3[2] * vrot[2];
      }
#pragma unroll 1
      for (int q = 0; q < nloc; ++q) {
        const artemis_nbody_particle_t &pl = spl[q];
        if (!pl.couple) continue;
        double g[3] = {0.0, 0.0, 0.0};
        {
          double dxp[3];
          for (int d = 0; d < 3; d++) dxp[d] = fr.x[d] - (pl.pos[d] - pl.xf[d]);
          const double dr2 = sqr(dxp[0]) + sqr(dxp[1]) + sqr(dxp[2]);
          const double idr3_ = nb_idr3(pl, dr2);
          for (int d = 0; d < 3; d++) g[d] += -pl.gm * idr3_ * dxp[d];
        }
        const double gx1 = g[0] * fr.e1[0] + g[1] * fr.e1[1] + g[2] * fr.e1[2];
        const double gx2 = g[0] * fr.e2[0] + g[1] * fr.e2[1] + g[2] * fr.e2[2];
        const double gx3 = g[0] * fr.e3[0] + g[1] * fr.e3[1] + g[2] * fr.e3[2];
        double f7[7] = {0, 0, 0, 0, 0, 0, 0}; // this particle's terms from this zone, in the old order of additions
        auto fluid = [&](const FluidView &f, int nvar, int n, bool gas) {
          const int ns = f.ns;
          const double dens = f.prim[b * nvar + n][c];
          const double v[3] = {f.prim[b * nvar + ns + 3 * n + 0][c], f.prim[b * nvar + ns + 3 * n + 1][c],
                               f.prim[b * nvar + ns + 3 * n + 2][c]};
          double vcart[3];
          vcart[0] = fr.e1[0] * v[0] + fr.e2[0] * v[1] + fr.e3[0] * v[2];
          vcart[1] = fr.e1[1] * v[0] + fr.e2[1] * v[1] + fr.e3[1] * v[2];
          vcart[2] = fr.e1[2] * v[0] + fr.e2[2] * v[1] + fr.e3[2] * v[2];
          double dm = 0.0, dmom[3] = {0.0, 0.0, 0.0}, dek = 0.0;
          const double dei = 0.0;
          nb_accrete(pl, fr.x, dens, vcart, vf, N.dt, dm, dmom, dek);
          const double dmx1 = dmom[0] * fr.e1[0] + dmom[1] * fr.e1[1] + dmom[2] * fr.e1[2];
          const double dmx2 = dmom[0] * fr.e2[0] + dmom[1] * fr.e2[1] + dmom[2] * fr.e2[2];
          const double dmx3 = dmom[0] * fr.e3[0] + dmom[1] * fr.e3[1] + dmom[2] * fr.e3[2];
          const double rdt = dens * N.dt;
          f.cons0[b * nvar + n][c] += dm;
          f.cons0[b * nvar + ns + 3 * n + 0][c] += hx[0] * (rdt * gx1 + dmx1);
          f.cons0[b * nvar + ns + 3 * n + 1][c] += hx[1] * (rdt * gx2 + dmx2);
          f.cons0[b * nvar + ns + 3 * n + 2][c] += hx[2] * (rdt * gx3 + dmx3);
          if (gas) {
            f.cons0[b * nvar + 4 * ns + n][c] += dek + dei + rdt * (v[0] * gx1 + v[1] * gx2 + v[2] * gx3);
            f.cons0[b * nvar + 5 * ns + n][c] += dei;
          }
          f7[0] -= vol * dm / N.dt;
          f7[1] -= g[0] * dens * vol;
          f7[2] -= g[1] * dens * vol;
          f7[3] -= g[2] * dens * vol;
          f7[4] -= dmom[0] / N.dt;
          f7[5] -= dmom[1] / N.dt;
          f7[6] -= dmom[2] / N.dt;
        };
        for (int n = 0; n < P.gas.ns; ++n) fluid(P.gas, 6 * P.gas.ns, n, true);
        for (int n = 0; n < P.dust.ns; ++n) fluid(P.dust, 4 * P.dust.ns, n, false);
#pragma unroll
        for (int qq = 0; qq < NB_CHUNK; ++qq)
#pragma unroll
          for (int m = 0; m < 7; ++m) lf[qq][m] += (qq == q) ? f7[m] : 0.0;
      }
    }
    // fixed-tree reduction: wave shuffles, then the four waves of the workgroup in order
#pragma unroll
    for (int q = 0; q < NB_CHUNK; ++q) {
      if (q >= nloc) continue; // (workgroup-uniform)
      for (int m = 0; m < 7; ++m) {
        double vq = lf[q][m];
        for (int off = 32; off > 0; off >>= 1) vq += __shfl_down(vq, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][m] = vq;
      }
      __syncthreads();
      if (threadIdx.x < 7)
        N.partial[(static_cast<long>(np0 + q) * gridDim.x + blockIdx.x) * 7 + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
      __syncthreads();
    }
  }
}

// The usual case -- at most one species per fluid, fewer than 2^31 zones -- with the zone's state in registers: the
// four / four primitives and six / four conserved variables of gas / dust are loaded together at the top of the
// iteration (independent loads: one memory latency per zone instead of one per read-modify-write), every particle
// acts on the registers in order, one store each at the end.  Same additions in the same order as the kernel above.
// APPLY = false: the seven sums alone (artemis_hip_nbody_force_sums: the stage kernels apply the accelerations themselves).
template <bool GAS, bool DUST, bool APPLY = true>
__global__ __launch_bounds__(256) void nbody_gravity_one_kernel(const PackView P, const NBodyView N_in) {
  __shared__ double red[4][7];
  __shared__ artemis_nbody_particle_t spl[NB_CHUNK];
  NBodyView N = N_in;
  if (N.dt_ptr) N.dt = *N.dt_ptr;
  const unsigned nx1 = P.ie - P.is + 1, nx2 = P.je - P.js + 1, nx3 = P.ke - P.ks + 1;
  const unsigned per_block = nx1 * nx2 * nx3, total = per_block * P.nb;
  for (int np0 = 0; np0 < N.npart; np0 += NB_CHUNK) {
    const int nloc = (N.npart - np0 < NB_CHUNK) ? N.npart - np0 : NB_CHUNK;
    __syncthreads();
    if (threadIdx.x < nloc) spl[threadIdx.x] = N.pl[np0 + threadIdx.x];
    __syncthreads();
    double lf[NB_CHUNK][7];
#pragma unroll
    for (int q = 0; q < NB_CHUNK; ++q)
#pragma unroll
      for (int m = 0; m < 7; ++m) lf[q][m] = 0.0;
    bool any = false, any_acc = false; // (any_acc: some coupled particle of the chunk accretes -- velocities are needed)
    for (int q = 0; q < nloc; ++q) any = any || spl[q].couple, any_acc = any_acc || (spl[q].couple && spl[q].racc > 0.0);
    const int nprim = (APPLY || any_acc) ? 4 : 1; // (workgroup-uniform: the sums of non-accreting particles read densities only)
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; any && t < total; t += gridDim.x * blockDim.x) {
      const unsigned b = t / per_block, r = t - b * per_block;
      const unsigned row = r / nx1;
      const int i = P.is + static_cast<int>(r - row * nx1), j = P.js + static_cast<int>(row % nx2), k = P.ks + static_cast<int>(row / nx2);
      const long c = static_cast<long>(k) * P.sk + static_cast<long>(j) * P.sj + i;
      // every load of the iteration up front
      double wg[4] = {0, 0, 0, 0}, ug[6] = {0, 0, 0, 0, 0, 0}, wd[4] = {0, 0, 0, 0}, ud[4] = {0, 0, 0, 0};
      if constexpr (GAS) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (m < nprim) wg[m] = P.gas.prim[b * 6 + m][c];
        if constexpr (APPLY) {
#pragma unroll
          for (int m = 0; m < 6; ++m) ug[m] = P.gas.cons0[b * 6 + m][c];
        }
      }
      if constexpr (DUST) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (m < nprim) wd[m] = P.dust.prim[b * 4 + m][c];
        if constexpr (APPLY) {
#pragma unroll
          for (int m = 0; m < 4; ++m) ud[m] = P.dust.cons0[b * 4 + m][c];
        }
      }
      const NbZone z = nb_zone(make_coords(P, b, k, j, i), N.omf);
      double vcg[3], vcd[3]; // Cartesian velocities of the two fluids (particle-independent)
      nb_cart_velocity(z.fr, wg, vcg), nb_cart_velocity(z.fr, wd, vcd);
#pragma unroll 1
      for (int q = 0; q < nloc; ++q) {
        const artemis_nbody_particle_t &pl = spl[q];
        if (!pl.couple) continue;
        const NbPull pull = nb_pull(pl, z.fr);
        double f7[7] = {0, 0, 0, 0, 0, 0, 0};
        if constexpr (GAS) nb_fluid<true, APPLY, true>(pl, z, pull, N.dt, wg, vcg, ug, f7);
        if constexpr (DUST) nb_fluid<false, APPLY, true>(pl, z, pull, N.dt, wd, vcd, ud, f7);
#pragma unroll
        for (int qq = 0; qq < NB_CHUNK; ++qq)
#pragma unroll
          for (int m = 0; m < 7; ++m) lf[qq][m] += (qq == q) ? f7[m] : 0.0;
      }
      if constexpr (GAS && APPLY) {
#pragma unroll
        for (int m = 0; m < 6; ++m) P.gas.cons0[b * 6 + m][c] = ug[m];
      }
      if constexpr (DUST && APPLY) {
#pragma unroll
        for (int m = 0; m < 4; ++m) P.dust.cons0[b * 4 + m][c] = ud[m];
      }
    }
#pragma unroll
    for (int q = 0; q < NB_CHUNK; ++q) {
      if (q >= nloc) continue; // (workgroup-uniform)
      for (int m = 0; m < 7; ++m) {
        double vq = lf[q][m];
        for (int off = 32; off > 0; off >>= 1) vq += __shfl_down(vq, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][m] = vq;
      }
      __syncthreads();
      if (threadIdx.x < 7)
        N.partial[(static_cast<long>(np0 + q) * gridDim.x + blockIdx.x) * 7 + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
      __syncthreads();
    }
  }
}

// The seven sums alone (artemis_hip_nbody_force_sums) as a march: a workgroup takes NX x NY columns of one block at a time
// and walks x3.  What a zone needs of its column -- table pointers, the x1 / x2 parts of the coordinates -- is formed once
// per column and no zone index is recovered by division.  One partial row per workgroup as above; the order of the
// additions differs from the kernel above (the sums are compared to round-off everywhere, never bit for bit).
template <bool GAS, bool DUST, int NX>
__global__ __launch_bounds__(256) void nbody_force_march_kernel(const PackView P, const NBodyView N_in, int tiles_i, int tiles_j) {
  constexpr int NY = 256 / NX;
  __shared__ double red[4][7];
  __shared__ artemis_nbody_particle_t spl[NB_CHUNK];
  NBodyView N = N_in;
  if (N.dt_ptr) N.dt = *N.dt_ptr;
  const int x = threadIdx.x % NX, y = threadIdx.x / NX;
  const int per = tiles_i * tiles_j, units = per * P.nb;
  for (int np0 = 0; np0 < N.npart; np0 += NB_CHUNK) {
    const int nloc = (N.npart - np0 < NB_CHUNK) ? N.npart - np0 : NB_CHUNK;
    __syncthreads();
    if (threadIdx.x < nloc) spl[threadIdx.x] = N.pl[np0 + threadIdx.x];
    __syncthreads();
    double lf[NB_CHUNK][7];
#pragma unroll
    for (int q = 0; q < NB_CHUNK; ++q)
#pragma unroll
      for (int m = 0; m < 7; ++m) lf[q][m] = 0.0;
    bool any = false, any_acc = false;
    for (int q = 0; q < nloc; ++q) any = any || spl[q].couple, any_acc = any_acc || (spl[q].couple && spl[q].racc > 0.0);
    const int nprim = any_acc ? 4 : 1; // (the sums of non-accreting particles read densities only)
    for (int u = blockIdx.x; any && u < units; u += gridDim.x) {
      const int b = u / per, tt = u - b * per, tj = tt / tiles_i, ti = tt - tj * tiles_i;
      const int i = P.is + ti * NX + x, j = P.js + tj * NY + y;
      if (i > P.ie || j > P.je) continue;
      const double *gp[4] = {nullptr, nullptr, nullptr, nullptr}, *dp[4] = {nullptr, nullptr, nullptr, nullptr};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if constexpr (GAS)
          if (m < nprim) gp[m] = P.gas.prim[b * 6 + m];
        if constexpr (DUST)
          if (m < nprim) dp[m] = P.dust.prim[b * 4 + m];
      }
      for (int k = P.ks; k <= P.ke; ++k) {
        const long c = static_cast<long>(k) * P.sk + static_cast<long>(j) * P.sj + i;
        double wg[4] = {0, 0, 0, 0}, wd[4] = {0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if constexpr (GAS)
            if (m < nprim) wg[m] = gp[m][c];
          if constexpr (DUST)
            if (m < nprim) wd[m] = dp[m][c];
        }
        const NbZone z = nb_zone(make_coords(P, b, k, j, i), N.omf);
        double vcg[3], vcd[3];
        nb_cart_velocity(z.fr, wg, vcg), nb_cart_velocity(z.fr, wd, vcd);
#pragma unroll 1
        for (int q = 0; q < nloc; ++q) {
          const artemis_nbody_particle_t &pl = spl[q];
          if (!pl.couple) continue;
          const NbPull pull = nb_pull(pl, z.fr);
          double f7[7] = {0, 0, 0, 0, 0, 0, 0};
          if constexpr (GAS) nb_fluid<true, false, true>(pl, z, pull, N.dt, wg, vcg, nullptr, f7);
          if constexpr (DUST) nb_fluid<false, false, true>(pl, z, pull, N.dt, wd, vcd, nullptr, f7);
#pragma unroll
          for (int qq = 0; qq < NB_CHUNK; ++qq)
#pragma unroll
            for (int m = 0; m < 7; ++m) lf[qq][m] += (qq == q) ? f7[m] : 0.0;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NB_CHUNK; ++q) {
      if (q >= nloc) continue; // (workgroup-uniform)
      for (int m = 0; m < 7; ++m) {
        double vq = lf[q][m];
        for (int off = 32; off > 0; off >>= 1) vq += __shfl_down(vq, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][m] = vq;
      }
      __syncthreads();
      if (threadIdx.x < 7)
        N.partial[(static_cast<long>(np0 + q) * gridDim.x + blockIdx.x) * 7 + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------
// RotatingFrame::ShearingBoxImpl (rotating_frame_impl.hpp:28-93), Cartesian.
__global__ __launch_bounds__(TX *TY) void shearing_box_kernel(const PackView P, double om0,
                                                              double qshear, double dt) {
  INTERIOR_CELL
  const ShearAcc sa = shear_terms(P.geom + 6 * b, P.ndim, k, i, om0, qshear);
  for (int n = 0; n < P.gas.ns; ++n) {
    GasCons u = load_gas_cons(P.gas, b, n, c);
    shear_gas(sa, dt, load_prim(P.gas, 6 * P.gas.ns, b, n, c, true), u);
    store_gas_cons(P.gas, b, n, c, u, false);
  }
  for (int n = 0; n < P.dust.ns; ++n) {
    DustCons u = load_dust_cons(P.dust, b, n, c);
    shear_dust(sa, dt, load_prim(P.dust, 4 * P.dust.ns, b, n, c, false), u);
    store_dust_cons(P.dust, b, n, c, u, false);
  }
}

// RotatingFrame::RotatingFrameImpl<GEOM> (rotating_frame_impl.hpp:95-199), every non-Cartesian
// system: the angular-momentum-conserving form built from the stage's MASS fluxes with the
// RFWeights of the cell, plus the centrifugal work on the gas energy.
__global__ __launch_bounds__(TX *TY) void rotating_frame_kernel(const PackView P, double om0, double dt) {
  INTERIOR_CELL
  const int multi_d = (P.ndim >= 2), three_d = (P.ndim == 3);
  const DCoords co = make_coords(P, b, k, j, i);
  const RotFrame r = rotating_frame_terms(co, om0, dt);
  const double ax1[2] = {co.area1(0), co.area1(1)};
  const double ax2[2] = {multi_d ? co.area2(0) : 0.0, multi_d ? co.area2(1) : 0.0};
  const double ax3[2] = {three_d ? co.area3(0) : 0.0, three_d ? co.area3(1) : 0.0};
  const double vol = co.volume();
  const long c2 = c + multi_d * P.sj, c3 = c + three_d * P.sk;
  const int d2 = multi_d ? 1 : 0, d3 = three_d ? 2 : 0; // inactive directions have no flux table
  for (int n = 0; n < P.gas.ns; ++n) {
    const int nv = 6 * P.gas.ns;
    const double *f1 = P.gas.flux[0][b * nv + n], *f2 = P.gas.flux[d2][b * nv + n], *f3 = P.gas.flux[d3][b * nv + n];
    const double flo[3] = {f1[c], f2[c], f3[c]}, fup[3] = {f1[c + 1], f2[c2], f3[c3]};
    GasCons u = load_gas_cons(P.gas, b, n, c);
    rotating_frame_gas(r, multi_d, three_d, flo, fup, ax1, ax2, ax3, vol, u);
    store_gas_cons(P.gas, b, n, c, u, false);
  }
  for (int n = 0; n < P.dust.ns; ++n) {
    const int nv = 4 * P.dust.ns;
    const double *f1 = P.dust.flux[0][b * nv + n], *f2 = P.dust.flux[d2][b * nv + n], *f3 = P.dust.flux[d3][b * nv + n];
    const double flo[3] = {f1[c], f2[c], f3[c]}, fup[3] = {f1[c + 1], f2[c2], f3[c3]};
    DustCons u = load_dust_cons(P.dust, b, n, c);
    rotating_frame_dust(r, multi_d, three_d, flo, fup, ax1, ax2, ax3, vol, u);
    store_dust_cons(P.dust, b, n, c, u, false);
  }
}

// Gas::Cooling::BetaCooling<GEOM, powerlaw> (beta_cooling.cpp:40-126); Tref and beta of the cell come
// from the host-filled tables.
__global__ __launch_bounds__(TX *TY) void cooling_kernel(const PackView P, const artemis_cooling_t C, double dt) {
  INTERIOR_CELL
  const DCoords co = make_coords(P, b, k, j, i);
  double hx[3];
  scale_factors_of(co, hx);
  const double omdt = cooling_omdt(co, C.gm, dt);
  const double T0 = C.tref[b][c], beta = C.beta[b][c];
  for (int n = 0; n < P.gas.ns; ++n) {
    GasCons u = load_gas_cons(P.gas, b, n, c);
    cooling_gas(P.gas, C.cv, omdt, T0, beta, hx, u);
    const int nv = 6 * P.gas.ns;
    P.gas.cons0[b * nv + 4 * P.gas.ns + n][c] = u.e, P.gas.cons0[b * nv + 5 * P.gas.ns + n][c] = u.eg;
  }
}

// ---------------------------------------------------------------------------------------
// Drag::DragSource (drag.cpp:89-175); D.damp_visc != NULL = damp_to_visc: the viscosity it points at (host
// memory) travels to the kernels by value as V.
static artemis_diffcoeff_t damp_visc_of(const artemis_drag_t &D) {
  artemis_diffcoeff_t v{};
  if (D.damp_visc) v = *D.damp_visc;
  return v;
}
// any <gas|dust/damping> rate non-zero?  Otherwise every ramp is dt * (0 * a + 0 * b) with finite a, b -- the zone
// centre lies strictly inside the damping bounds the deck reader fills in -- i.e. + 0.0, and the kernels skip them.
static int damping_on(const artemis_drag_t &D) {
  bool on = false;
  for (int d = 0; d < 3; ++d) {
    on = on || D.gas.irate[d] != 0.0 || D.gas.orate[d] != 0.0 || D.dust.irate[d] != 0.0 || D.dust.orate[d] != 0.0;
    // (a threshold ON the mesh bound makes the ramp's quotient x / 0: evaluated as written then)
    on = on || D.gas.ix[d] == D.xmin[d] || D.gas.ox[d] == D.xmax[d] || D.dust.ix[d] == D.xmin[d] || D.dust.ox[d] == D.xmax[d];
  }
  return on ? 1 : 0;
}
__device__ __forceinline__ void damping_ramps(const artemis_damping_t &p, const artemis_drag_t &D,
                                              int ndim, const double xv[3], double dt,
                                              double f[3]) {
  const int multi_d = (ndim >= 2), three_d = (ndim == 3);
  f[0] = dt * (p.irate[0] * ((xv[0] < p.ix[0]) * sqr((xv[0] - p.ix[0]) / (p.ix[0] - D.xmin[0]))) +
               p.orate[0] * ((xv[0] > p.ox[0]) * sqr((xv[0] - p.ox[0]) / (p.ox[0] - D.xmax[0]))));
  f[1] = multi_d * dt *
         (p.irate[1] * ((xv[1] < p.ix[1]) * sqr((xv[1] - p.ix[1]) / (p.ix[1] - D.xmin[1]))) +
          p.orate[1] * ((xv[1] > p.ox[1]) * sqr((xv[1] - p.ox[1]) / (p.ox[1] - D.xmax[1]))));
  f[2] = three_d * dt *
         (p.irate[2] * ((xv[2] < p.ix[2]) * sqr((xv[2] - p.ix[2]) / (p.ix[2] - D.xmin[2]))) +
          p.orate[2] * ((xv[2] > p.ox[2]) * sqr((xv[2] - p.ox[2]) / (p.ox[2] - D.xmax[2]))));
}

// SelfDragSourceImpl (drag.hpp:171-294)
__global__ __launch_bounds__(TX *TY) void self_drag_kernel(const PackView P, const artemis_drag_t D, const artemis_diffcoeff_t V,
                                                           double dt_host, const double *dt_dev) {
  INTERIOR_CELL
  const double dt = dt_dev ? *dt_dev : dt_host;
  const DCoords co = make_coords(P, b, k, j, i);
  const double xv[3] = {co.x1v(), co.x2v(), co.x3v()};
  double hx[3];
  scale_factors_of(co, hx);
  const CylVec cv = to_cyl_with_vec(co, xv);
  double bg[3], bd[3];
  damping_ramps(D.gas, D, P.ndim, xv, dt, bg);
  damping_ramps(D.dust, D, P.ndim, xv, dt, bd);
  {
    const FluidView &f = P.gas;
    const int ns = f.ns, nv = 6 * ns;
    for (int n = 0; n < ns; ++n) {
      const double dens = f.cons0[b * nv + n][c];
      double *m1 = f.cons0[b * nv + ns + 3 * n + 0], *m2 = f.cons0[b * nv + ns + 3 * n + 1];
      double *m3 = f.cons0[b * nv + ns + 3 * n + 2];
      const double vg[3] = {m1[c] / (hx[0] * dens), m2[c] / (hx[1] * dens), m3[c] / (hx[2] * dens)};
      double mu = 0.0; // DiffusionCoeff<null>::Get (diffusion_coeff.hpp:185-189)
      if (D.damp_visc) { // drag.hpp:234-240: GetSpecificInternalEnergy of species n, then the viscosity
        const double u_d = amax(dens, f.dfloor);
        const double rv1 = m1[c] / hx[0], rv2 = m2[c] / hx[1], rv3 = m3[c] / hx[2];
        const double ke = 0.5 * (sqr(rv1) + sqr(rv2) + sqr(rv3)) / u_d;
        const double e_cons = f.cons0[b * nv + 4 * ns + n][c];
        const double ue_cons = e_cons - ke;
        double sie = (ue_cons > f.de_switch * e_cons) ? ue_cons / u_d : f.cons0[b * nv + 5 * ns + n][c] / u_d;
        sie = amax(sie, f.siefloor);
        mu = coeff_of(V, 0.0, P.gm1, dens, sie, b, c);
      }
      const double vR = -1.5 * mu / (cv.R * dens);
      const double vd[3] = {cv.e1 * vR, cv.e2 * vR, cv.e3 * vR};
      const double dm1 = -bg[0] * dens * (vg[0] - vd[0]) / (1.0 + bg[0]);
      const double dm2 = -bg[1] * dens * (vg[1] - vd[1]) / (1.0 + bg[1]);
      const double dm3 = -bg[2] * dens * (vg[2] - vd[2]) / (1.0 + bg[2]);
      m1[c] += hx[0] * dm1;
      m2[c] += hx[1] * dm2;
      m3[c] += hx[2] * dm3;
      f.cons0[b * nv + 4 * ns + n][c] += dm1 * (vg[0] + 0.5 * dm1 / dens) +
                                         dm2 * (vg[1] + 0.5 * dm2 / dens) +
                                         dm3 * (vg[2] + 0.5 * dm3 / dens);
    }
  }
  {
    const FluidView &f = P.dust;
    const int ns = f.ns, nv = 4 * ns;
    for (int n = 0; n < ns; ++n)
      for (int d = 0; d < 3; ++d) {
        double *m = f.cons0[b * nv + ns + 3 * n + d];
        const double mom = m[c];
        m[c] = mom - bd[d] * mom / (1.0 + bd[d]);
      }
  }
}

// SimpleDragSourceImpl (drag.hpp:296-482): backward-Euler gas-dust momentum exchange for one
// gas species and ns dust species; two passes over the dust species (sum, then update).
// FINISH: instead of storing the coupled momenta / energy back to cons0, carry on with
// SetAuxillaryFields (fill_derived.cpp:58-71) and ConsToPrim (:132-164) of the cell and write the
// new primitives to P.{gas,dust}.prim -- the tail of the general fused stage in one pass
// (one gas species).
// Every input (pointer-table entries and values) is read BEFORE the first store and every result is stored at the end:
// behind a store the table entries -- wave-uniform addresses -- could no longer come through the scalar cache, and each
// would be a vector load with a full memory wait behind it (one thread per zone: nothing else hides it).  MAXD: the
// dust species the registers hold (more species: the generic loop below, stores interleaved as before).
constexpr int DRAG_MAXD = 4;
template <bool FINISH, int ND> // ND: the number of dust species as a compile-time constant (0 .. DRAG_MAXD), or -1 = any
ADEV void simple_drag_zone(const PackView &P, const artemis_drag_t &D, const artemis_diffcoeff_t &V, const double dt,
                           const int damp_on, const int b, const int k, const int j, const int i, const long c) {
  const DCoords co = make_coords(P, b, k, j, i);
  const double xv[3] = {co.x1v(), co.x2v(), co.x3v()};
  double hx[3];
  scale_factors_of(co, hx);
  const CylVec cv = to_cyl_with_vec(co, xv);
  double bg[3] = {0.0, 0.0, 0.0}, bd[3] = {0.0, 0.0, 0.0};
  if (damp_on) { // (uniform; twelve divisions per zone otherwise spent on dt * (0 * x + 0 * y) = +0: damping_on below)
    damping_ramps(D.gas, D, P.ndim, xv, dt, bg);
    damping_ramps(D.dust, D, P.ndim, xv, dt, bd);
  }
  const FluidView &G = P.gas, &F = P.dust;
  const int nsg = G.ns, nvg = 6 * nsg, nsd = (ND >= 0) ? ND : F.ns, nvd = 4 * nsd;
  // ---- inputs ----------------------------------------------------------------------------------------------------------
  double *const pg_m[3] = {G.cons0[b * nvg + nsg + 0], G.cons0[b * nvg + nsg + 1], G.cons0[b * nvg + nsg + 2]};
  double *const pg_e = G.cons0[b * nvg + 4 * nsg];
  double *const og_d = FINISH ? G.prim[b * nvg + 0] : nullptr, *const og_s = FINISH ? G.prim[b * nvg + 5 * nsg] : nullptr;
  double *const og_v[3] = {FINISH ? G.prim[b * nvg + nsg + 0] : nullptr, FINISH ? G.prim[b * nvg + nsg + 1] : nullptr,
                           FINISH ? G.prim[b * nvg + nsg + 2] : nullptr};
  const double dg = G.cons0[b * nvg + 0][c];
  const double mg0[3] = {pg_m[0][c], pg_m[1][c], pg_m[2][c]};
  const double e_cons = pg_e[c];
  const double eg_cons = G.cons0[b * nvg + 5 * nsg][c];
  constexpr int nreg = (ND >= 0) ? ND : 0; // species held in registers
  constexpr int NR = nreg > 0 ? nreg : 1;
  double dens_[NR], md_[NR][3];
  double *pd_m[NR][3], *od_d[NR], *od_v[NR][3];
#pragma unroll
  for (int n = 0; n < nreg; ++n)
    {
      dens_[n] = F.cons0[b * nvd + n][c];
      od_d[n] = FINISH ? F.prim[b * nvd + n] : nullptr;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        pd_m[n][d] = F.cons0[b * nvd + nsd + 3 * n + d];
        md_[n][d] = pd_m[n][d][c];
        od_v[n][d] = FINISH ? F.prim[b * nvd + nsd + 3 * n + d] : nullptr;
      }
    }
  auto dust_dens = [&](int n) { return (ND >= 0) ? dens_[ND >= 0 ? n : 0] : F.cons0[b * nvd + n][c]; };
  auto dust_mom = [&](int n, int d) { return (ND >= 0) ? md_[ND >= 0 ? n : 0][d] : F.cons0[b * nvd + nsd + 3 * n + d][c]; };
  double en = 0.0, mnew[3] = {0.0, 0.0, 0.0};
  constexpr int NRr = (ND > 0) ? ND : 1;
  double newd[NRr][3];
  auto coupled = [&](auto FT) {
    constexpr bool FAST = decltype(FT)::value;
    auto dv = [](double num, double den) { return FAST ? div(num, den) : num / den; };
  // ---- the coupled update (drag.hpp:296-482) ------------------------------------------------------------------------
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
  const double mu = D.damp_visc ? coeff_of(V, 0.0, P.gm1, dg, sieg, b, c) : 0.0; // drag.hpp:392-393
  const double vR = -1.5 * mu / (cv.R * dg);
  const double vt[3] = {cv.e1 * vR, cv.e2 * vR, cv.e3 * vR};
  double fd[3] = {0., 0., 0.}, fvd[3] = {0., 0., 0.};
  double vth = 0.0;
  const bool stokes = (D.model == ARTEMIS_DRAG_STOKES);
  if (stokes) vth = sqrt(8.0 / M_PI * P.gm1 * sieg);
  const double vdt[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int n = 0; n < nsd; ++n) {
    const double dens = dust_dens(n);
    const double vd[3] = {dv(dust_mom(n, 0), hx[0] * dens), dv(dust_mom(n, 1), hx[1] * dens), dv(dust_mom(n, 2), hx[2] * dens)};
    double tc = D.tau[n];
    if (stokes) tc = D.scale * D.grain_density / dg * D.sizes[n] / vth;
    const double alpha = dt * ((tc <= 0.0) ? DBL_MAX : 1.0 / tc);
    for (int d = 0; d < 3; d++) {
      const double rhop = dv(dens * alpha, 1.0 + alpha + bd[d]);
      fd[d] += rhop * (1.0 + bd[d]);
      fvd[d] += rhop * (vd[d] + bd[d] * vdt[d]);
    }
  }
  double vgp[3];
  for (int d = 0; d < 3; d++)
    vgp[d] = dv(dg * (vg[d] + bg[d] * vt[d]) + fvd[d], dg * (1.0 + bg[d]) + fd[d]);
  double delta_g[3] = {0.0, 0.0, 0.0};
  for (int d = 0; d < 3; d++) fvd[d] = 0.;
#pragma unroll
  for (int n = 0; n < nsd; ++n) {
    const double dens = dust_dens(n);
    const double vd[3] = {dv(dust_mom(n, 0), hx[0] * dens), dv(dust_mom(n, 1), hx[1] * dens), dv(dust_mom(n, 2), hx[2] * dens)};
    double tc = D.tau[n];
    if (stokes) tc = D.scale * D.grain_density / dg * D.sizes[n] / vth;
    const double alpha = dt * ((tc <= 0.0) ? DBL_MAX : 1.0 / tc);
    for (int d = 0; d < 3; d++) {
      double delta_d = 0.;
      const double rhop = dv(dens * alpha, 1.0 + alpha + bd[d]);
      const double delta = rhop * ((vgp[d] - vd[d] + bd[d] * (vgp[d] - vdt[d])));
      delta_d += delta;
      delta_g[d] -= delta;
      delta_d -= dv(bd[d] * dens, 1. + alpha + bd[d]) * (vd[d] - vdt[d] + alpha * (vgp[d] - vdt[d]));
      fvd[d] += rhop * (vd[d] - vt[d] + bd[d] * (vdt[d] - vt[d]));
      const double m = dust_mom(n, d) + hx[d] * delta_d;
      double out = m;
      if constexpr (FINISH) {
        const double w_d = (dens > F.dfloor) ? dens : F.dfloor;
        out = dv(m, w_d * hx[d]);
      }
      if constexpr (ND >= 0) {
        newd[n][d] = out;
      } else if constexpr (FINISH) { // (more species than the registers hold: stored at once, as the first form of this kernel did)
        F.prim[b * nvd + nsd + 3 * n + d][c] = out;
        if (d == 0) F.prim[b * nvd + n][c] = (dens > F.dfloor) ? dens : F.dfloor;
      } else {
        F.cons0[b * nvd + nsd + 3 * n + d][c] = out;
      }
    }
  }
  en = e_cons;
  for (int d = 0; d < 3; d++) {
    const double prefac = dv(dg * bg[d], 1.0 + bg[d] + fd[d]);
    delta_g[d] -= prefac * (dg * (vg[d] - vt[d]) + fvd[d]);
    mnew[d] = mg0[d] + hx[d] * delta_g[d];
    en += 0.5 * (vg[d] + vgp[d]) * delta_g[d];
  }
  };
  // (IEEE divisions throughout: this one-thread-per-zone kernel waits for memory, not for its ~35 divisions -- the guarded
  // hand-scheduled form measured 10 % slower on the 29 M-zone configs[4] mesh)
  coupled(std::false_type{});
  // ---- stores ------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int n = 0; n < nreg; ++n)
    {
      if constexpr (FINISH) {
        od_d[n][c] = (dens_[n] > F.dfloor) ? dens_[n] : F.dfloor;
#pragma unroll
        for (int d = 0; d < 3; ++d) od_v[n][d][c] = newd[n][d];
      } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) pd_m[n][d][c] = newd[n][d];
      }
    }
  if constexpr (!FINISH) {
    for (int d = 0; d < 3; ++d) pg_m[d][c] = mnew[d];
    pg_e[c] = en;
  } else {
    const double u_d = (dg > G.dfloor) ? dg : G.dfloor;
    const double u_d2 = amax(dg, G.dfloor);
    const double rv1 = mnew[0] / hx[0], rv2 = mnew[1] / hx[1], rv3 = mnew[2] / hx[2];
    const double ke = 0.5 * (sqr(rv1) + sqr(rv2) + sqr(rv3)) / u_d2;
    const double ue_cons = en - ke;
    double sie = (ue_cons > G.de_switch * en) ? ue_cons / u_d2 : eg_cons / u_d2;
    sie = amax(sie, G.siefloor);
    double u_u = sie * u_d;
    const double uflr = G.siefloor * u_d;
    u_u = (u_u > uflr) ? u_u : uflr;
    const double w_d = u_d;
    const double w_s = u_u / w_d;
    og_d[c] = w_d;
    og_v[0][c] = mnew[0] / (w_d * hx[0]);
    og_v[1][c] = mnew[1] / (w_d * hx[1]);
    og_v[2][c] = mnew[2] / (w_d * hx[2]);
    og_s[c] = (w_s > G.siefloor) ? w_s : G.siefloor;
  }
}
template <bool FINISH, int ND>
__global__ __launch_bounds__(TX *TY) void simple_drag_kernel(const PackView P, const artemis_drag_t D, const artemis_diffcoeff_t V,
                                                             double dt_host, const double *dt_dev, const int damp_on) {
  INTERIOR_CELL
  simple_drag_zone<FINISH, ND>(P, D, V, dt_dev ? *dt_dev : dt_host, damp_on, b, k, j, i, c);
}
// ... of the LISTED zones only (a refined mesh's fix-up zones, whose conserved state artemis_hip_ml_stage_fixup has just
// rewritten: artemis_hip_stage_finish_cells)
template <int ND>
__global__ __launch_bounds__(256) void simple_drag_cells_kernel(const PackView P, const artemis_drag_t D, const artemis_diffcoeff_t V,
                                                                double dt, const int damp_on,
                                                                const artemis_ml_fix_cell_t *__restrict__ cells, const int ncells) {
  const int q = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
  if (q >= ncells) return;
  const artemis_ml_fix_cell_t z = cells[q];
  const long c = (static_cast<long>(z.k) * P.nj + z.j) * P.ni + z.i;
  simple_drag_zone<true, ND>(P, D, V, dt, damp_on, z.block, z.k, z.j, z.i, c);
}

// force[7 n + q] += the rows of `partial` in index order.  The rows come through LDS in chunks (coalesced loads by the
// whole workgroup); thread (n, q) then adds its column of the chunk in order.
__global__ __launch_bounds__(1024) void nbody_force_sum_kernel(const artemis_nbody_particle_t *pl, int npart, int grid,
                                                              int chunk_rows, const double *partial, double *force) {
  extern __shared__ double rows[]; // [chunk_rows][npart][7]
  const int t = threadIdx.x, nq = 7 * npart;
  const int n = t / 7, q = t - 7 * n;
  const bool mine = t < nq && pl[n].couple;
  double sum = 0.0;
  for (int w0 = 0; w0 < grid; w0 += chunk_rows) {
    const int nr = min(chunk_rows, grid - w0);
    __syncthreads();
    for (int e = t; e < nr * nq; e += blockDim.x) {
      const int w = e / nq, col = e - w * nq, pn = col / 7, pq = col - 7 * pn;
      rows[e] = partial[(static_cast<long>(pn) * grid + (w0 + w)) * 7 + pq];
    }
    __syncthreads();
    if (mine)
      for (int w = 0; w < nr; ++w) sum += rows[w * nq + t];
  }
  if (mine) force[7 * n + q] += sum;
}

} // namespace

int nbody_grid(const PackView &P) {
  const long total = static_cast<long>(P.ie - P.is + 1) * (P.je - P.js + 1) * (P.ke - P.ks + 1) * P.nb;
  return static_cast<int>(std::max<long>(1, std::min<long>(2048, (total + 255) / 256)));
}
void launch_nbody_gravity(const PackView &P, const artemis_nbody_particle_t *pl_dev, int npart, double omf, double dt,
                          double *partial_dev, hipStream_t s) {
  NBodyView N;
  N.pl = pl_dev, N.npart = npart, N.omf = omf, N.dt = dt, N.dt_ptr = nullptr, N.partial = partial_dev;
  const long total = static_cast<long>(P.ie - P.is + 1) * (P.je - P.js + 1) * (P.ke - P.ks + 1) * P.nb;
  if (P.gas.ns <= 1 && P.dust.ns <= 1 && total < (1L << 31) - (1L << 20) && !opt(OPT_NBODY_GENERAL)) {
    const dim3 grid(nbody_grid(P)), block(256);
    if (P.gas.ns && P.dust.ns) hipLaunchKernelGGL((nbody_gravity_one_kernel<true, true>), grid, block, 0, s, P, N);
    else if (P.gas.ns) hipLaunchKernelGGL((nbody_gravity_one_kernel<true, false>), grid, block, 0, s, P, N);
    else if (P.dust.ns) hipLaunchKernelGGL((nbody_gravity_one_kernel<false, true>), grid, block, 0, s, P, N);
    return;
  }
  hipLaunchKernelGGL(nbody_gravity_kernel, dim3(nbody_grid(P)), dim3(256), 0, s, P, N);
}
// The seven sums per particle alone, accumulated on the device: force[7 n + q] += sum over the workgroups' partial rows
// in index order (the additions artemis_hip_nbody_gravity makes on the host, same order).
bool nbody_force_sums_covers(const PackView &P) {
  const long total = static_cast<long>(P.ie - P.is + 1) * (P.je - P.js + 1) * (P.ke - P.ks + 1) * P.nb;
  return P.gas.ns <= 1 && P.dust.ns <= 1 && total < (1L << 31) - (1L << 20);
}
void launch_nbody_force_sums(const PackView &P, const artemis_nbody_particle_t *pl_dev, int npart, double omf, double dt,
                             const double *dt_dev, double *partial_dev, double *force_dev, hipStream_t s) {
  NBodyView N;
  N.pl = pl_dev, N.npart = npart, N.omf = omf, N.dt = dt, N.dt_ptr = dt_dev, N.partial = partial_dev;
  const int nx1 = P.ie - P.is + 1, nx2 = P.je - P.js + 1;
  const int NX = nx1 <= 16 ? 16 : (nx1 <= 32 ? 32 : 64), NY = 256 / NX;
  const int tiles_i = (nx1 + NX - 1) / NX, tiles_j = (nx2 + NY - 1) / NY;
  // (one partial row per workgroup in the caller's scratch: artemis_hip_nbody_force_scratch rows at most)
  const dim3 grid(static_cast<unsigned>(std::min<long>(nbody_grid(P), static_cast<long>(tiles_i) * tiles_j * P.nb))), block(256);
  if (!P.gas.ns && !P.dust.ns) return;
#define FORCE_MARCH(G, D)                                                                                              \
  do {                                                                                                                 \
    if (NX == 16) hipLaunchKernelGGL((nbody_force_march_kernel<G, D, 16>), grid, block, 0, s, P, N, tiles_i, tiles_j); \
    else if (NX == 32) hipLaunchKernelGGL((nbody_force_march_kernel<G, D, 32>), grid, block, 0, s, P, N, tiles_i, tiles_j); \
    else hipLaunchKernelGGL((nbody_force_march_kernel<G, D, 64>), grid, block, 0, s, P, N, tiles_i, tiles_j);          \
  } while (0)
  if (P.gas.ns && P.dust.ns) FORCE_MARCH(true, true);
  else if (P.gas.ns) FORCE_MARCH(true, false);
  else FORCE_MARCH(false, true);
#undef FORCE_MARCH
  const int nq = 7 * npart; // (<= 896: artemis_hip_nbody_force_sums takes at most 128 particles)
  const int chunk_rows = std::max(1, std::min<int>(grid.x, 32768 / (8 * nq)));
  hipLaunchKernelGGL(nbody_force_sum_kernel, dim3(1), dim3(std::max(256, 64 * ((nq + 63) / 64))), sizeof(double) * chunk_rows * nq, s,
                     pl_dev, npart, static_cast<int>(grid.x), chunk_rows, partial_dev, force_dev);
}
void launch_external_gravity(const PackView &P, const artemis_gravity_t &G, double dt, hipStream_t s) {
  hipLaunchKernelGGL(gravity_kernel, interior_grid(P), interior_threads(P), 0, s, P, G, dt);
}
void launch_shearing_box(const PackView &P, double omega, double qshear, double dt, hipStream_t s) {
  hipLaunchKernelGGL(shearing_box_kernel, interior_grid(P), interior_threads(P), 0, s, P, omega, qshear, dt);
}
void launch_cooling(const PackView &P, const artemis_cooling_t &C, double dt, hipStream_t s) {
  hipLaunchKernelGGL(cooling_kernel, interior_grid(P), interior_threads(P), 0, s, P, C, dt);
}
void launch_rotating_frame(const PackView &P, double omega, double dt, hipStream_t s) {
  hipLaunchKernelGGL(rotating_frame_kernel, interior_grid(P), interior_threads(P), 0, s, P, omega, dt);
}
template <bool FINISH>
static void launch_simple_drag(const PackView &P, const artemis_drag_t &D, double dt, const double *dt_dev, hipStream_t s) {
#define DRAG_ND(N)                                                                                                            \
  hipLaunchKernelGGL((simple_drag_kernel<FINISH, N>), interior_grid(P), interior_threads(P), 0, s, P, D, damp_visc_of(D), dt, \
                     dt_dev, damping_on(D))
  switch (P.dust.ns) {
  case 0: DRAG_ND(0); break;
  case 1: DRAG_ND(1); break;
  case 2: DRAG_ND(2); break;
  case 3: DRAG_ND(3); break;
  case 4: DRAG_ND(4); break;
  default: DRAG_ND(-1);
  }
#undef DRAG_ND
}
void launch_drag_source(const PackView &P, const artemis_drag_t &D, double dt, const double *dt_dev,
                        hipStream_t s) {
  if (D.type == ARTEMIS_DRAG_SELF)
    hipLaunchKernelGGL(self_drag_kernel, interior_grid(P), interior_threads(P), 0, s, P, D, damp_visc_of(D), dt, dt_dev);
  else
    launch_simple_drag<false>(P, D, dt, dt_dev, s);
}
// simple_dust drag + SetAuxillaryFields + ConsToPrim of a one-gas-species pack in one pass: reads
// cons0, writes the primitives of P (the general fused stage points them at its out tables)
bool launch_drag_finish(const PackView &P, const artemis_drag_t &D, double dt, const double *dt_dev,
                        hipStream_t s) {
  if (D.type != ARTEMIS_DRAG_SIMPLE_DUST || P.gas.ns != 1) return false;
  launch_simple_drag<true>(P, D, dt, dt_dev, s);
  return true;
}
bool launch_drag_finish_cells(const PackView &P, const artemis_drag_t &D, double dt, const artemis_ml_fix_cell_t *cells,
                              int ncells, hipStream_t s) {
  if (D.type != ARTEMIS_DRAG_SIMPLE_DUST || P.gas.ns != 1) return false;
  if (ncells <= 0) return true;
#define DRAG_ND(N)                                                                                                     \
  hipLaunchKernelGGL((simple_drag_cells_kernel<N>), dim3((ncells + 255) / 256), dim3(256), 0, s, P, D, damp_visc_of(D), dt, \
                     damping_on(D), cells, ncells)
  switch (P.dust.ns) {
  case 0: DRAG_ND(0); break;
  case 1: DRAG_ND(1); break;
  case 2: DRAG_ND(2); break;
  case 3: DRAG_ND(3); break;
  case 4: DRAG_ND(4); break;
  default: DRAG_ND(-1);
  }
#undef DRAG_ND
  return true;
}
// can the dust march do the drag finish itself (kernels_curv.hip, simple_drag1_finish)?
bool drag_finish_in_march(const PackView &P, const artemis_drag_t &D) {
  return D.type == ARTEMIS_DRAG_SIMPLE_DUST && P.gas.ns == 1 && P.dust.ns == 1 && !D.damp_visc && !damping_on(D);
}

} // namespace artemis
