// Multilevel (SMR / AMR) block-graph data path: ghost exchange across refinement levels, restriction of
// ghost halos, prolongation from coarse buffers and flux correction -- what Parthenon's boundary
// communication does around Artemis' tasks on a refined mesh (artemis_driver.cpp:196-202, :258), with
// Artemis' own refinement operators (utils/refinement/restriction.hpp:42-114, prolongation.hpp:83-184).
//
// gfx950 shape: a refined mesh is thousands of small boxes (a 16x16x8 block has 26 neighbour boxes of 64 to
// 1024 zones), so nothing here is launched per box.  The host uploads a table of box-to-box operations once
// per mesh; ONE launch walks the whole table, one 256-thread workgroup per operation, x1 fastest across
// lanes so every row segment is a coalesced 8-byte-per-lane stream.  Remote operations pack into / unpack
// from one contiguous message buffer per peer in the same launch.  HBM-bound copy work; no LDS needed.
#include "device_math.hpp"
#include "geometry_core.hpp"
#include "kernels.hpp"

namespace artemis {
namespace {

struct MlView {
  PackView P;
  double *const *gc, *const *dc; // coarse-buffer tables
  const double *cgeom, *cmetric;
  int cni, cnj, cnk, cis, cjs, cks; // coarse-buffer extents and first interior indices
  int cnx1, cnx2, cnx3;             // coarse interior zones
};

// v-th FillGhost variable (gas rho, v, sie with the pressure slot skipped, gas.cpp:244-270; dust rho, v) of
// block b in the fine tables or in the coarse-buffer tables
ADEV double *fill_ptr(const MlView &M, bool coarse, int b, int v) {
  const int nsg = M.P.gas.ns, nsd = M.P.dust.ns;
  const int ngas = 5 * nsg;
  if (v < ngas) {
    const int slot = (v < 4 * nsg) ? v : v + nsg;
    return (coarse ? M.gc : M.P.gas.prim)[b * 6 * nsg + slot];
  }
  return (coarse ? M.dc : M.P.dust.prim)[b * 4 * nsd + (v - ngas)];
}

ADEV DCoords fine_coords(const PackView &P, int b, int k, int j, int i) {
  const double *m = P.metric ? P.metric + b * metric_block_stride(P.nj, P.nk) : nullptr;
  return coords_of(P.coords, P.geom + 6 * b, m, P.nj, P.nk, k, j, i);
}
ADEV DCoords coarse_coords(const MlView &M, int b, int k, int j, int i) {
  const double *m = M.cmetric ? M.cmetric + b * metric_block_stride(M.cnj, M.cnk) : nullptr;
  return coords_of(M.P.coords, M.cgeom + 6 * b, m, M.cnj, M.cnk, k, j, i);
}

// RestrictAverage<GEOM, el = CC> of the 2^ndim fine zones at (fk, fj, fi) of block b (restriction.hpp:73-112:
// volume weights, pairwise summation order kept)
struct RestrictW {
  double vol[2][2][2], tvol;
};
ADEV RestrictW restrict_weights(const PackView &P, int b, int fk, int fj, int fi) {
  RestrictW w;
  const bool X1 = P.ndim > 0, X2 = P.ndim > 1, X3 = P.ndim > 2;
  // (fixed 2 x 2 x 2 trip counts with the activity test inside: fully unrolled, every index a constant, so the eight
  //  weights live in registers -- with run-time loop bounds they were 64 bytes of scratch per lane)
#pragma unroll
  for (int ok = 0; ok < 2; ++ok)
#pragma unroll
    for (int oj = 0; oj < 2; ++oj)
#pragma unroll
      for (int oi = 0; oi < 2; ++oi) {
        const bool on = (ok == 0 || X3) && (oj == 0 || X2) && (oi == 0 || X1);
        w.vol[ok][oj][oi] = on ? fine_coords(P, b, fk + (X3 ? ok : 0), fj + (X2 ? oj : 0), fi + (X1 ? oi : 0)).volume() : 0.0;
      }
  w.tvol = ((w.vol[0][0][0] + w.vol[0][1][0]) + (w.vol[0][0][1] + w.vol[0][1][1])) +
           ((w.vol[1][0][0] + w.vol[1][1][0]) + (w.vol[1][0][1] + w.vol[1][1][1]));
  return w;
}
ADEV double restrict_value(const PackView &P, const RestrictW &w, const double *q, int fk, int fj, int fi) {
  const bool X1 = P.ndim > 0, X2 = P.ndim > 1, X3 = P.ndim > 2;
  double t[2][2][2];
#pragma unroll
  for (int ok = 0; ok < 2; ++ok)
#pragma unroll
    for (int oj = 0; oj < 2; ++oj)
#pragma unroll
      for (int oi = 0; oi < 2; ++oi) {
        const bool on = (ok == 0 || X3) && (oj == 0 || X2) && (oi == 0 || X1);
        // (an inactive offset reads the zone itself: a valid address, the value is not used)
        const double v = q[(static_cast<long>(fk + (X3 ? ok : 0)) * P.nj + (fj + (X2 ? oj : 0))) * P.ni + fi + (X1 ? oi : 0)];
        t[ok][oj][oi] = on ? w.vol[ok][oj][oi] * v : 0.0;
      }
  return (((t[0][0][0] + t[0][1][0]) + (t[0][0][1] + t[0][1][1])) + ((t[1][0][0] + t[1][1][0]) + (t[1][0][1] + t[1][1][1]))) /
         w.tvol;
}

// ---- ghost exchange --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ml_exchange_kernel(const MlView M, const artemis_ml_op_t *__restrict__ ops,
                                                          double *__restrict__ sbuf, const double *__restrict__ rbuf) {
  const artemis_ml_op_t op = ops[blockIdx.x];
  const PackView &P = M.P;
  const int nfill = 5 * P.gas.ns + 4 * P.dust.ns;
  const long ncell = static_cast<long>(op.n[0]) * op.n[1] * op.n[2];
  const bool to_coarse = (op.kind == ARTEMIS_ML_FROM_COARSER);
  const int dni = to_coarse ? M.cni : P.ni, dnj = to_coarse ? M.cnj : P.nj;
  for (long t = threadIdx.x; t < ncell; t += blockDim.x) {
    const int q0 = static_cast<int>(t % op.n[0]), q1 = static_cast<int>((t / op.n[0]) % op.n[1]);
    const int q2 = static_cast<int>(t / (static_cast<long>(op.n[0]) * op.n[1]));
    const int i = op.lo[0] + q0, j = op.lo[1] + q1, k = op.lo[2] + q2;
    const long dcell = (static_cast<long>(k) * dnj + j) * dni + i;
    if (op.src_block < 0) { // unpack
      for (int v = 0; v < nfill; ++v) fill_ptr(M, to_coarse, op.dst_block, v)[dcell] = rbuf[op.buf + v * ncell + t];
      continue;
    }
    if (op.kind == ARTEMIS_ML_FROM_FINER) {
      const int fi = 2 * i + op.off[0], fj = (P.ndim > 1) ? 2 * j + op.off[1] : j, fk = (P.ndim > 2) ? 2 * k + op.off[2] : k;
      const RestrictW w = restrict_weights(P, op.src_block, fk, fj, fi);
      for (int v = 0; v < nfill; ++v) {
        const double val = restrict_value(P, w, fill_ptr(M, false, op.src_block, v), fk, fj, fi);
        if (op.dst_block < 0) sbuf[op.buf + v * ncell + t] = val;
        else fill_ptr(M, false, op.dst_block, v)[dcell] = val;
      }
    } else { // SAME, FROM_COARSER: plain copies out of a fine interior
      const long scell = (static_cast<long>(k + op.off[2]) * P.nj + (j + op.off[1])) * P.ni + (i + op.off[0]);
      for (int v = 0; v < nfill; ++v) {
        const double val = fill_ptr(M, false, op.src_block, v)[scell];
        if (op.dst_block < 0) sbuf[op.buf + v * ncell + t] = val;
        else fill_ptr(M, to_coarse, op.dst_block, v)[dcell] = val;
      }
    }
  }
}

// ---- restriction of a block's own fine array (interior + ghost halo) into its coarse buffer ----------------
__global__ __launch_bounds__(256) void ml_restrict_halos_kernel(const MlView M, const int *__restrict__ blocks) {
  const PackView &P = M.P;
  const int b = blocks[blockIdx.y];
  const int h = P.ng / 2; // coarse zones of ghost halo
  const int e1 = M.cnx1 + 2 * h, e2 = (P.ndim > 1) ? M.cnx2 + 2 * h : 1, e3 = (P.ndim > 2) ? M.cnx3 + 2 * h : 1;
  const long t = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<long>(e1) * e2 * e3) return;
  const int ci = M.cis - h + static_cast<int>(t % e1);
  const int cj = (P.ndim > 1) ? M.cjs - h + static_cast<int>((t / e1) % e2) : 0;
  const int ck = (P.ndim > 2) ? M.cks - h + static_cast<int>(t / (static_cast<long>(e1) * e2)) : 0;
  const int fi = (ci - M.cis) * 2 + P.is, fj = (P.ndim > 1) ? (cj - M.cjs) * 2 + P.js : 0;
  const int fk = (P.ndim > 2) ? (ck - M.cks) * 2 + P.ks : 0;
  const RestrictW w = restrict_weights(P, b, fk, fj, fi);
  const long ccell = (static_cast<long>(ck) * M.cnj + cj) * M.cni + ci;
  const int nfill = 5 * P.gas.ns + 4 * P.dust.ns;
  for (int v = 0; v < nfill; ++v)
    fill_ptr(M, true, b, v)[ccell] = restrict_value(P, w, fill_ptr(M, false, b, v), fk, fj, fi);
}

// ---- ProlongateSharedMinMod<GEOM> from the coarse buffer into fine ghost zones ----------------------------
ADEV double centre_of(const DCoords &co, int d) { return d == 1 ? co.x1v() : (d == 2 ? co.x2v() : co.x3v()); }
__global__ __launch_bounds__(256) void ml_prolongate_kernel(const MlView M, const artemis_ml_box_t *__restrict__ boxes) {
  const artemis_ml_box_t bx = boxes[blockIdx.x];
  const PackView &P = M.P;
  const int b = bx.block;
  const bool X1 = P.ndim > 0, X2 = P.ndim > 1, X3 = P.ndim > 2;
  const long ncell = static_cast<long>(bx.n[0]) * bx.n[1] * bx.n[2];
  const int nfill = 5 * P.gas.ns + 4 * P.dust.ns;
  auto cidx = [&](int k, int j, int i) { return (static_cast<long>(k) * M.cnj + j) * M.cni + i; };
  auto fidx = [&](int k, int j, int i) { return (static_cast<long>(k) * P.nj + j) * P.ni + i; };
  for (long t = threadIdx.x; t < ncell; t += blockDim.x) {
    const int ci = bx.lo[0] + static_cast<int>(t % bx.n[0]), cj = bx.lo[1] + static_cast<int>((t / bx.n[0]) % bx.n[1]);
    const int ck = bx.lo[2] + static_cast<int>(t / (static_cast<long>(bx.n[0]) * bx.n[1]));
    const int fi = (ci - M.cis) * 2 + P.is, fj = X2 ? (cj - M.cjs) * 2 + P.js : 0, fk = X3 ? (ck - M.cks) * 2 + P.ks : 0;
    // GetGridSpacings<GEOM, d> (prolongation.hpp:39-68): geometry only, shared by the variables
    double dxm[3] = {1, 1, 1}, dxp[3] = {1, 1, 1}, dxfm[3] = {0, 0, 0}, dxfp[3] = {0, 0, 0};
    for (int d = 1; d <= P.ndim; ++d) {
      const int dk = (d == 3), dj = (d == 2), di = (d == 1);
      const double xm = centre_of(coarse_coords(M, b, ck - dk, cj - dj, ci - di), d);
      const double xc = centre_of(coarse_coords(M, b, ck, cj, ci), d);
      const double xp = centre_of(coarse_coords(M, b, ck + dk, cj + dj, ci + di), d);
      const double fxm = centre_of(fine_coords(P, b, fk, fj, fi), d);
      const double fxp = centre_of(fine_coords(P, b, fk + dk, fj + dj, fi + di), d);
      dxm[d - 1] = xc - xm, dxp[d - 1] = xp - xc, dxfm[d - 1] = xc - fxm, dxfp[d - 1] = fxp - xc;
    }
    for (int v = 0; v < nfill; ++v) {
      const double *q = fill_ptr(M, true, b, v);
      const double fc = q[cidx(ck, cj, ci)];
      double g[3] = {0, 0, 0};
      for (int d = 1; d <= P.ndim; ++d) { // GradMinMod (:73-80); SIGN(a) = (a < 0) ? -1 : 1 (parthenon, upstream)
        const int dk = (d == 3), dj = (d == 2), di = (d == 1);
        const double gxm = (fc - q[cidx(ck - dk, cj - dj, ci - di)]) / dxm[d - 1];
        const double gxp = (q[cidx(ck + dk, cj + dj, ci + di)] - fc) / dxp[d - 1];
        const double sm = (gxm < 0.) ? -1. : 1., sp = (gxp < 0.) ? -1. : 1.;
        const double am = fabs(gxm), ap = fabs(gxp);
        g[d - 1] = 0.5 * (sm + sp) * ((ap < am) ? ap : am);
      }
      const double gx1m = g[0], gx1p = g[0], gx2m = g[1], gx2p = g[1], gx3m = g[2], gx3p = g[2];
      const double dx1fm = dxfm[0], dx1fp = dxfp[0], dx2fm = dxfm[1], dx2fp = dxfp[1], dx3fm = dxfm[2], dx3fp = dxfp[2];
      double *o = fill_ptr(M, false, b, v);
      o[fidx(fk, fj, fi)] = fc - (gx1m * dx1fm + gx2m * dx2fm + gx3m * dx3fm);
      if (X1) o[fidx(fk, fj, fi + 1)] = fc + (gx1p * dx1fp - gx2m * dx2fm - gx3m * dx3fm);
      if (X2) o[fidx(fk, fj + 1, fi)] = fc - (gx1m * dx1fm - gx2p * dx2fp + gx3m * dx3fm);
      if (X2 && X1) o[fidx(fk, fj + 1, fi + 1)] = fc + (gx1p * dx1fp + gx2p * dx2fp - gx3m * dx3fm);
      if (X3) o[fidx(fk + 1, fj, fi)] = fc - (gx1m * dx1fm + gx2m * dx2fm - gx3p * dx3fp);
      if (X3 && X1) o[fidx(fk + 1, fj, fi + 1)] = fc + (gx1p * dx1fp - gx2m * dx2fm + gx3p * dx3fp);
      if (X3 && X2) o[fidx(fk + 1, fj + 1, fi)] = fc - (gx1m * dx1fm - gx2p * dx2fp - gx3p * dx3fp);
      if (X3 && X2 && X1) o[fidx(fk + 1, fj + 1, fi + 1)] = fc + (gx1p * dx1fp + gx2p * dx2fp + gx3p * dx3fp);
    }
  }
}

// ---- PrimToCons's primitive floors on the ghost zones of the blocks next to a level boundary ----------------------
// The reference converts every zone of every block after the boundary fill (fill_derived.cpp:227, :245, :262) and that
// conversion floors gas density, gas sie and dust density where they sit.  The one-kernel stages keep no conserved ghost
// zones and skip the conversion; same-level copies of floored zones are floored, but a restricted average can round one
// ulp below a floor its eight zones sit on and the sum of three limited slopes of a prolongation can undershoot the lowest
// neighbour.  This pass applies the floors to the ghost zones of the listed blocks, behind the physical conditions (which
// read the zones as the fill left them, like the reference).  Reads alone almost everywhere: stored only where a floor acts.
__global__ __launch_bounds__(256) void ml_floor_ghosts_kernel(const PackView P, const int *__restrict__ blocks) {
  const int b = blocks[blockIdx.y];
  const long n = static_cast<long>(P.ni) * P.nj * P.nk;
  const long c = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const int i = c % P.ni, j = (c / P.ni) % P.nj, k = c / (static_cast<long>(P.ni) * P.nj);
  if (i >= P.is && i <= P.ie && j >= P.js && j <= P.je && k >= P.ks && k <= P.ke) return;
  const int nsg = P.gas.ns, nsd = P.dust.ns;
  for (int m = 0; m < nsg; ++m) {
    double *rho = P.gas.prim[b * 6 * nsg + m], *se = P.gas.prim[b * 6 * nsg + 5 * nsg + m];
    const double w_d = rho[c], w_s = se[c];
    // (`!(w > floor)`: the reference's `(w > floor) ? w : floor`, NaN going to the floor as well)
    if (!(w_d > P.gas.dfloor)) rho[c] = P.gas.dfloor;
    if (!(w_s > P.gas.siefloor)) se[c] = P.gas.siefloor;
  }
  for (int m = 0; m < nsd; ++m) {
    double *rho = P.dust.prim[b * 4 * nsd + m];
    const double w_d = rho[c];
    if (!(w_d > P.dust.dfloor)) rho[c] = P.dust.dfloor;
  }
}

// ---- flux correction: RestrictAverage<GEOM, el = F_dir> (restriction.hpp:57-112) --------------------------
// v-th flux array of direction d of block b: gas cons fluxes (6 ns), the pressure flux (ns), the diffusion
// fluxes (4 ns, when the caller has them), dust fluxes (4 ns)
ADEV double *flux_ptr(const PackView &P, int b, int d, int v, bool with_diff) {
  const int nsg = P.gas.ns, nsd = P.dust.ns;
  if (v < 6 * nsg) return P.gas.flux[d][b * 6 * nsg + v];
  v -= 6 * nsg;
  if (v < nsg) return P.gas.pflux[d][b * nsg + v];
  v -= nsg;
  if (with_diff) {
    if (v < 4 * nsg) return P.gas.dflux[d][b * 4 * nsg + v];
    v -= 4 * nsg;
  }
  return P.dust.flux[d][b * 4 * nsd + v];
}
__global__ __launch_bounds__(256) void ml_flux_kernel(const PackView P, const artemis_ml_op_t *__restrict__ ops,
                                                      double *__restrict__ sbuf, const double *__restrict__ rbuf,
                                                      int nflux, int with_diff) {
  const artemis_ml_op_t op = ops[blockIdx.x];
  const long ncell = static_cast<long>(op.n[0]) * op.n[1] * op.n[2];
  const int d = op.dir;
  // included offsets: the two tangential directions that are active (restriction.hpp:60-65)
  const bool I1 = (P.ndim > 0) && d != 0, I2 = (P.ndim > 1) && d != 1, I3 = (P.ndim > 2) && d != 2;
  for (long t = threadIdx.x; t < ncell; t += blockDim.x) {
    const int q0 = static_cast<int>(t % op.n[0]), q1 = static_cast<int>((t / op.n[0]) % op.n[1]);
    const int q2 = static_cast<int>(t / (static_cast<long>(op.n[0]) * op.n[1]));
    const int i = op.lo[0] + q0, j = op.lo[1] + q1, k = op.lo[2] + q2;
    const long dcell = (static_cast<long>(k) * P.nj + j) * P.ni + i;
    if (op.src_block < 0) {
      for (int v = 0; v < nflux; ++v) flux_ptr(P, op.dst_block, d, v, with_diff)[dcell] = rbuf[op.buf + v * ncell + t];
      continue;
    }
    const int fi = (d == 0) ? op.off[0] : ((P.ndim > 0) ? 2 * i + op.off[0] : i);
    const int fj = (d == 1) ? op.off[1] : ((P.ndim > 1) ? 2 * j + op.off[1] : j);
    const int fk = (d == 2) ? op.off[2] : ((P.ndim > 2) ? 2 * k + op.off[2] : k);
    double area[2][2][2];
#pragma unroll
    for (int ok = 0; ok < 2; ++ok)
#pragma unroll
      for (int oj = 0; oj < 2; ++oj)
#pragma unroll
        for (int oi = 0; oi < 2; ++oi) {
          const bool on = (ok == 0 || I3) && (oj == 0 || I2) && (oi == 0 || I1);
          double a_ = 0.0;
          if (on) {
            const DCoords co = fine_coords(P, op.src_block, fk + ok, fj + oj, fi + oi);
            a_ = (d == 0) ? co.area1(0) : ((d == 1) ? co.area2(0) : co.area3(0));
          }
          area[ok][oj][oi] = a_;
        }
    const double tvol = ((area[0][0][0] + area[0][1][0]) + (area[0][0][1] + area[0][1][1])) +
                        ((area[1][0][0] + area[1][1][0]) + (area[1][0][1] + area[1][1][1]));
    for (int v = 0; v < nflux; ++v) {
      const double *q = flux_ptr(P, op.src_block, d, v, with_diff);
      double tm[2][2][2];
#pragma unroll
      for (int ok = 0; ok < 2; ++ok)
#pragma unroll
        for (int oj = 0; oj < 2; ++oj)
#pragma unroll
          for (int oi = 0; oi < 2; ++oi) {
            const bool on = (ok == 0 || I3) && (oj == 0 || I2) && (oi == 0 || I1);
            const double v = q[(static_cast<long>(fk + (I3 ? ok : 0)) * P.nj + (fj + (I2 ? oj : 0))) * P.ni + fi + (I1 ? oi : 0)];
            tm[ok][oj][oi] = on ? area[ok][oj][oi] * v : 0.0;
          }
      const double val = (((tm[0][0][0] + tm[0][1][0]) + (tm[0][0][1] + tm[0][1][1])) +
                          ((tm[1][0][0] + tm[1][1][0]) + (tm[1][0][1] + tm[1][1][1]))) /
                         tvol;
      if (op.dst_block < 0) sbuf[op.buf + v * ncell + t] = val;
      else flux_ptr(P, op.dst_block, d, v, with_diff)[dcell] = val;
    }
  }
}

MlView make_ml_view(const PackView &P, const artemis_ml_pack_t &ml) {
  MlView M;
  M.P = P;
  M.gc = ml.gas_coarse, M.dc = ml.dust_coarse, M.cgeom = ml.cgeom, M.cmetric = ml.cmetric;
  const int nx1 = P.ie - P.is + 1, nx2 = P.je - P.js + 1, nx3 = P.ke - P.ks + 1;
  M.cnx1 = nx1 / 2, M.cnx2 = (P.ndim > 1) ? nx2 / 2 : 1, M.cnx3 = (P.ndim > 2) ? nx3 / 2 : 1;
  M.cis = P.ng, M.cjs = (P.ndim > 1) ? P.ng : 0, M.cks = (P.ndim > 2) ? P.ng : 0;
  M.cni = M.cnx1 + 2 * M.cis, M.cnj = M.cnx2 + 2 * M.cjs, M.cnk = M.cnx3 + 2 * M.cks;
  return M;
}

} // namespace

void launch_ml_exchange(const PackView &P, const artemis_ml_pack_t &ml, const artemis_ml_op_t *ops, int nops, double *sbuf,
                        const double *rbuf, hipStream_t s) {
  if (nops <= 0) return;
  hipLaunchKernelGGL(ml_exchange_kernel, dim3(nops), dim3(256), 0, s, make_ml_view(P, ml), ops, sbuf, rbuf);
}
void launch_ml_flux_correction(const PackView &P, const artemis_ml_op_t *ops, int nops, double *sbuf, const double *rbuf,
                               hipStream_t s) {
  if (nops <= 0) return;
  const int with_diff = (P.gas.ns > 0 && P.gas.dflux[0] != nullptr) ? 1 : 0;
  const int nflux = 7 * P.gas.ns + (with_diff ? 4 * P.gas.ns : 0) + 4 * P.dust.ns;
  hipLaunchKernelGGL(ml_flux_kernel, dim3(nops), dim3(256), 0, s, P, ops, sbuf, rbuf, nflux, with_diff);
}
void launch_ml_restrict_halos(const PackView &P, const artemis_ml_pack_t &ml, const int *blocks, int nblocks, hipStream_t s) {
  if (nblocks <= 0) return;
  const MlView M = make_ml_view(P, ml);
  const int h = P.ng / 2;
  const long n = static_cast<long>(M.cnx1 + 2 * h) * ((P.ndim > 1) ? M.cnx2 + 2 * h : 1) * ((P.ndim > 2) ? M.cnx3 + 2 * h : 1);
  hipLaunchKernelGGL(ml_restrict_halos_kernel, dim3((n + 255) / 256, nblocks), dim3(256), 0, s, M, blocks);
}
void launch_ml_prolongate(const PackView &P, const artemis_ml_pack_t &ml, const artemis_ml_box_t *boxes, int nboxes,
                          hipStream_t s) {
  if (nboxes <= 0) return;
  hipLaunchKernelGGL(ml_prolongate_kernel, dim3(nboxes), dim3(256), 0, s, make_ml_view(P, ml), boxes);
}

void launch_ml_floor_ghosts(const PackView &P, const int *blocks, int nblocks, hipStream_t s) {
  if (nblocks <= 0) return;
  const long n = static_cast<long>(P.ni) * P.nj * P.nk;
  hipLaunchKernelGGL(ml_floor_ghosts_kernel, dim3((n + 255) / 256, nblocks), dim3(256), 0, s, P, blocks);
}

} // namespace artemis
