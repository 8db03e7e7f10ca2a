// Cell-centred fused stage: the general one-kernel-per-fluid form of the reference's stage
// (artemis_driver.cpp:182-255) for everything the tuned gas kernel (kernels_fused.hip) does not
// cover -- dust, several species, PPM, curvilinear coordinates, gravity / rotating-frame / drag.
//
// One thread owns one cell.  For each species it computes the 2*ndim face fluxes of its own cell
// straight from the input primitives (each face is therefore solved by both cells that share
// it: twice the Riemann work of the per-task flux kernel, in exchange for no flux, pressure-flux
// or face-velocity arrays in HBM, no LDS and no barrier), rebuilds u0 = PrimToCons(prim_in) and
// u1 = PrimToCons(prim_u1) in registers (PrimToCons runs over the whole block at the end of every
// stage, fill_derived.cpp:212-276, so this is an identity), applies ApplyUpdate, FluxSource,
// ExternalGravity and RotatingFrameForce, and -- unless drag couples the fluids -- SetAuxillaryFields
// and ConsToPrim, writing the new primitives to a second buffer.  With drag the conserved state
// goes to cons0 and the per-task DragSource / SetAuxillaryFields / ConsToPrim kernels finish.
// HBM traffic per cell-stage and species: ~5 reads (stencil neighbours hit L2) + 5 for u1 +
// 5 writes for gas (4+4+4 for dust) instead of the per-task chain's ~124 (gas) doubles.
// Gas pressure of stencil cells is recomputed as max(0, gm1*rho*sie) (fill_derived.cpp:247), so
// ghost zones only need the FillGhost variables.  Every expression tree is shared with the
// per-task kernels (task_device.hpp, sources_device.hpp): results are bit-identical.
#include <cstdlib>

#include "device_math.hpp"
#include "diffusion_device.hpp"
#include "geometry.hpp"
#include "kernels.hpp"
#include "options.hpp"
#include "nbody_device.hpp"
#include "pack_view.hpp"
#include "sources_device.hpp"
#include "task_device.hpp"

namespace artemis {
namespace {
constexpr int TX = 64, TY = 4;
// Thread shape of the one-thread-per-zone kernels: 64 x 4 by default; for narrow mesh blocks (refined meshes
// run 16^3 blocks) the x1 extent of the workgroup shrinks to the next power of two >= nx and the rows it
// frees fold along x2, so that a wave's 64 lanes stay on real zones (a 16-zone row filled a quarter of them).
inline dim3 tile_threads(int nx) {
  int tx = TX;
  while (tx > 8 && tx / 2 >= nx) tx >>= 1;
  return dim3(tx, TX * TY / tx);
}
inline dim3 interior_grid(const PackView &P) {
  const dim3 t = tile_threads(P.ie - P.is + 1);
  return dim3((P.ie - P.is + t.x) / t.x, (P.je - P.js + t.y) / t.y, (P.ke - P.ks + 1) * P.nb);
}

struct CellStageArgs {
  double gam0, gam1, beta_dt, bdt;
  const double *bdt_ptr; // optional device scalar beta*dt (replaces beta_dt and bdt)
  double *const *in, *const *u1, *const *out; // prim tables of this fluid
  int to_cons;                                // 1: store the post-source conserved state in cons0
  int grav_on, rf_on;
  artemis_gravity_t grav;
  double rf_omega, rf_qshear;
  // EXTRA instantiations only: DiffusionUpdate from the stored diffusion fluxes, the curvilinear
  // rotating frame from this cell's own mass fluxes, beta cooling
  int diff_on, do_viscosity, rfc_on, cool_on;
  double *const *dsum; // artemis_stage_general_args_t.diffusion_sums (one gas species) or null
  artemis_cooling_t cool;
  // Gravity::NBodyGravity in the gravity task's slot (artemis_stage_general_args_t.nbody_dev)
  const artemis_nbody_particle_t *nb_pl;
  int nb_n;
  double nb_omf;
  // FIX instantiations only (refined meshes, artemis_hip_ml_stage_fixup): the zones to redo
  const artemis_ml_fix_cell_t *fix;
  int nfix;
};

// The flux fields SetFluxCorrections replaces on a coarse-fine face (kernels_amr.hip flux_ptr: the six conserved
// fluxes and the pressure flux of gas, the four of dust), read from the arrays at the zone that stores the face;
// the face velocity is not a flux field and keeps the value the zone computed.
template <int FLUID>
ADEV void load_corrected(const FluidView &f, int b, int nv, int ns, int n, int dd, long cf, FaceFlux &F) {
  double *const *fx = f.flux[dd];
  const int m0 = b * nv + ns + 3 * n;
  F.fd = fx[b * nv + n][cf];
  F.fmx = fx[m0 + dd][cf], F.fmy = fx[m0 + (dd + 1) % 3][cf], F.fmz = fx[m0 + (dd + 2) % 3][cf];
  if constexpr (FLUID == 0) {
    F.fe = fx[b * nv + 4 * ns + n][cf], F.feg = fx[b * nv + 5 * ns + n][cf];
    F.pf = f.pflux[dd][b * ns + n][cf];
  }
}

// Stencil of one variable around cell c along a stride: w[3] = cell c, w[3+m] = cell c + m*st.
template <int RECON>
ADEV void load_stencil(const double *q, long c, long st, double w[7]) {
  constexpr int R = (RECON == 2) ? 3 : ((RECON == 1) ? 2 : 1);
#pragma unroll
  for (int m = -R; m <= R; ++m) w[3 + m] = q[c + m * st];
}

// Cartesian PLM: the two faces of a cell need the limited slopes of three cells (c-1, c, c+1); the
// cell's own slope serves both faces.  plm_dqm_fast is plm.hpp:32-47 with the hand-scheduled
// division of device_math.hpp (same bits as `/`, checked by artemis_hip_selftest_divsqrt and by
// the parity tests of this kernel against the CPU restatement of the reference).
ADEV void plm_pair_fast(const double w[7], double &Llo, double &Rlo, double &Lup, double &Rup) {
  const double sm = plm_dqm_fast(w[1], w[2], w[3]);
  const double sc = plm_dqm_fast(w[2], w[3], w[4]);
  const double sp = plm_dqm_fast(w[3], w[4], w[5]);
  Llo = w[2] + sm, Rlo = w[3] - sc; // lower face: ql from cell c-1, qr from cell c
  Lup = w[3] + sc, Rup = w[4] - sp; // upper face: ql from cell c, qr from cell c+1
}
template <int FLUID, int RIEMANN>
ADEV void solve_pair_fast(const PackView &P, int d, const double wd[7], const double w1[7],
                          const double w2[7], const double w3[7], const double wp[7],
                          const double we[7], FaceFlux &lo, FaceFlux &up) {
  const double *vx = (d == 0) ? w1 : ((d == 1) ? w2 : w3);
  const double *vy = (d == 0) ? w2 : ((d == 1) ? w3 : w1);
  const double *vz = (d == 0) ? w3 : ((d == 1) ? w1 : w2);
  if constexpr (FLUID == 0) {
    Prim6 Ll, Rl, Lu, Ru;
    plm_pair_fast(wd, Ll.d, Rl.d, Lu.d, Ru.d), plm_pair_fast(vx, Ll.vx, Rl.vx, Lu.vx, Ru.vx);
    plm_pair_fast(vy, Ll.vy, Rl.vy, Lu.vy, Ru.vy), plm_pair_fast(vz, Ll.vz, Rl.vz, Lu.vz, Ru.vz);
    plm_pair_fast(wp, Ll.p, Rl.p, Lu.p, Ru.p), plm_pair_fast(we, Ll.e, Rl.e, Lu.e, Ru.e);
    if constexpr (RIEMANN == 0) {
      const double gm1 = P.gm1, igm1 = 1.0 / gm1, gamma = gm1 + 1.0; // hllc.hpp:75-77
      const double alpha = (gamma + 1.0) / (2.0 * gamma);
      hllc_gas_fast(gm1, igm1, gamma, alpha, Ll, Rl, lo);
      hllc_gas_fast(gm1, igm1, gamma, alpha, Lu, Ru, up);
    } else {
      riemann_gas<RIEMANN>(P.gm1, Ll, Rl, lo);
      riemann_gas<RIEMANN>(P.gm1, Lu, Ru, up);
    }
  } else {
    Prim4 Ll, Rl, Lu, Ru;
    plm_pair_fast(wd, Ll.d, Rl.d, Lu.d, Ru.d), plm_pair_fast(vx, Ll.vx, Rl.vx, Lu.vx, Ru.vx);
    plm_pair_fast(vy, Ll.vy, Rl.vy, Lu.vy, Ru.vy), plm_pair_fast(vz, Ll.vz, Rl.vz, Lu.vz, Ru.vz);
    riemann_dust<RIEMANN>(Ll, Rl, lo);
    riemann_dust<RIEMANN>(Lu, Ru, up);
  }
}

// Lower (side 0) or upper (side 1) face of cell (k,j,i) along dir for species n: the body of the
// per-task flux kernel (fluid_fluxes.hpp:105-126 per direction) evaluated from register stencils.
template <int FLUID, int RIEMANN, int RECON, bool CURV>
ADEV FaceFlux face_of_cell(const PackView &P, const FluidView &f, double *const *prim, int b, int n,
                           int dir, int side, int k, int j, int i, const double wd[7],
                           const double w1[7], const double w2[7], const double w3[7],
                           const double wp[7], const double we[7]) {
  // face index along dir: the cell that stores this face (its lower face)
  const int fk = k + ((dir == 3) ? side : 0), fj = j + ((dir == 2) ? side : 0),
            fi = i + ((dir == 1) ? side : 0);
  PlmGeo gl{}, gr{};
  double hs[3] = {1.0, 1.0, 1.0};
  if constexpr (CURV) {
    if constexpr (RECON == 1) {
      gl = plm_geo(P, b, dir, fk - (dir == 3), fj - (dir == 2), fi - (dir == 1));
      gr = plm_geo(P, b, dir, fk, fj, fi);
    }
    make_coords(P, b, fk, fj, fi).face_scale(dir, hs);
  }
  const int d = dir - 1;
  const int o = 3 + side; // stencil slot of the cell above the face
  const double *vx = (d == 0) ? w1 : ((d == 1) ? w2 : w3);
  const double *vy = (d == 0) ? w2 : ((d == 1) ? w3 : w1);
  const double *vz = (d == 0) ? w3 : ((d == 1) ? w1 : w2);
  FaceFlux F;
  if constexpr (FLUID == 0) {
    Prim6 L, R;
    face_states<RECON, CURV>(wd + o, 1, gl, gr, L.d, R.d);
    face_states<RECON, CURV>(vx + o, 1, gl, gr, L.vx, R.vx);
    face_states<RECON, CURV>(vy + o, 1, gl, gr, L.vy, R.vy);
    face_states<RECON, CURV>(vz + o, 1, gl, gr, L.vz, R.vz);
    face_states<RECON, CURV>(wp + o, 1, gl, gr, L.p, R.p);
    face_states<RECON, CURV>(we + o, 1, gl, gr, L.e, R.e);
    riemann_gas<RIEMANN>(P.gm1, L, R, F);
  } else {
    Prim4 L, R;
    face_states<RECON, CURV>(wd + o, 1, gl, gr, L.d, R.d);
    face_states<RECON, CURV>(vx + o, 1, gl, gr, L.vx, R.vx);
    face_states<RECON, CURV>(vy + o, 1, gl, gr, L.vy, R.vy);
    face_states<RECON, CURV>(vz + o, 1, gl, gr, L.vz, R.vz);
    riemann_dust<RIEMANN>(L, R, F);
  }
  if constexpr (CURV) F.fmx *= hs[d], F.fmy *= hs[(d + 1) % 3], F.fmz *= hs[(d + 2) % 3];
  (void)f, (void)prim, (void)n;
  return F;
}

// STORED: the epilogue form -- face fluxes come from the flux / pressure-flux / face-velocity arrays a
// preceding CalculateFluxes left in the pack, u0 / u1 from cons0 / cons1, and the new primitives are
// written in place (nothing here reads a neighbour's primitives).
// FIX: the zones of a list instead of the interior of every block (flat 256-thread workgroups), with the faces
// flagged per zone taking the corrected fluxes of a refined mesh from the flux arrays (include/artemis_hip.h,
// "flux correction as a thin fix-up").
template <int FLUID, int RIEMANN, int RECON, bool CURV, bool EXTRA, bool STORED = false, bool FIX = false>
__global__ __launch_bounds__(TX *TY) void stage_cell_kernel(const PackView P, const CellStageArgs a_in) {
  int i, j, k, b;
  unsigned fixed_faces = 0u;
  if constexpr (FIX) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a_in.nfix) return;
    const artemis_ml_fix_cell_t fc = a_in.fix[t];
    i = fc.i, j = fc.j, k = fc.k, b = fc.block, fixed_faces = fc.faces;
  } else {
    i = P.is + blockIdx.x * blockDim.x + threadIdx.x;
    j = P.js + blockIdx.y * blockDim.y + threadIdx.y;
    const int nkr = P.ke - P.ks + 1;
    b = blockIdx.z / nkr;
    k = P.ks + blockIdx.z % nkr;
    if (i > P.ie || j > P.je) return;
  }
  const long c = (static_cast<long>(k) * P.nj + j) * P.ni + i;
  CellStageArgs a = a_in;
  if (a.bdt_ptr) a.beta_dt = a.bdt = *a.bdt_ptr; // wave-uniform scalar load
  const FluidView &f = (FLUID == 0) ? P.gas : P.dust;
  const int ns = f.ns, nv = (FLUID == 0 ? 6 : 4) * ns;
  const bool multi_d = P.ndim >= 2, three_d = P.ndim == 3;
  const CellMetric g = cell_metric<CURV>(P, b, k, j, i);
  double hx[3];
  scale_factors<CURV>(P, b, k, j, i, hx);
  DCoords co;
  if constexpr (CURV) co = make_coords(P, b, k, j, i);
  GravAcc ga{};
  if (a.grav_on) {
    const DCoords cg = make_coords(P, b, k, j, i);
    ga = gravity_accel(a.grav, cg, P.ndim, a.bdt);
  }
  ShearAcc sa{};
  if (a.rf_on) sa = shear_terms(P.geom + 6 * b, P.ndim, k, i, a.rf_omega, a.rf_qshear);
  RotFrame rfc{};
  DiffCell dcell{};
  double cool_omdt = 0.0, cool_T0 = 0.0, cool_beta = 0.0;
  if constexpr (EXTRA) {
    if (a.rfc_on) rfc = rotating_frame_terms(make_coords(P, b, k, j, i), a.rf_omega, a.bdt);
    if (FLUID == 0 && a.diff_on) dcell = diffusion_cell<CURV>(P, b, k, j, i);
    if (FLUID == 0 && a.cool_on) {
      cool_omdt = cooling_omdt(make_coords(P, b, k, j, i), a.cool.gm, a.bdt);
      cool_T0 = a.cool.tref[b][c], cool_beta = a.cool.beta[b][c];
    }
  }

  for (int n = 0; n < ns; ++n) {
    const double *qd = a.in[b * nv + n], *q1 = a.in[b * nv + ns + 3 * n + 0];
    const double *q2 = a.in[b * nv + ns + 3 * n + 1], *q3 = a.in[b * nv + ns + 3 * n + 2];
    const double *qe = (FLUID == 0) ? a.in[b * nv + 5 * ns + n] : nullptr;
    // ---- fluxes of the 2*ndim faces; divergence accumulated in ApplyUpdate's order -----------
    double divf[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}; // d, m1, m2, m3, e, eg
    double plo[3] = {0, 0, 0}, pup[3] = {0, 0, 0}, vlo[3] = {0, 0, 0}, vup[3] = {0, 0, 0};
    double mlo[3] = {0, 0, 0}, mup[3] = {0, 0, 0}; // mass fluxes (EXTRA: curvilinear rotating frame)
    FluidPrim w; // this cell's stage-input primitives
    w.rho = qd[c], w.v1 = q1[c], w.v2 = q2[c], w.v3 = q3[c], w.sie = (FLUID == 0) ? qe[c] : 0.0;
    for (int dir = 1; dir <= P.ndim; ++dir) {
      const long st = (dir == 1) ? 1 : ((dir == 2) ? P.sj : P.sk);
      FaceFlux lo, up;
      if constexpr (STORED) {
        const int dd = dir - 1;
        const long cu = c + st;
        double *const *fx = f.flux[dd];
        const int m0 = b * nv + ns + 3 * n;
        lo.fd = fx[b * nv + n][c], up.fd = fx[b * nv + n][cu];
        lo.fmx = fx[m0 + dd][c], up.fmx = fx[m0 + dd][cu];
        lo.fmy = fx[m0 + (dd + 1) % 3][c], up.fmy = fx[m0 + (dd + 1) % 3][cu];
        lo.fmz = fx[m0 + (dd + 2) % 3][c], up.fmz = fx[m0 + (dd + 2) % 3][cu];
        if constexpr (FLUID == 0) {
          lo.fe = fx[b * nv + 4 * ns + n][c], up.fe = fx[b * nv + 4 * ns + n][cu];
          lo.feg = fx[b * nv + 5 * ns + n][c], up.feg = fx[b * nv + 5 * ns + n][cu];
          lo.pf = f.pflux[dd][b * ns + n][c], up.pf = f.pflux[dd][b * ns + n][cu];
          lo.vf = f.vface[dd][b * ns + n][c], up.vf = f.vface[dd][b * ns + n][cu];
        }
      } else {
      double wd[7], w1[7], w2[7], w3[7], wp[7], we[7];
      load_stencil<RECON>(qd, c, st, wd), load_stencil<RECON>(q1, c, st, w1);
      load_stencil<RECON>(q2, c, st, w2), load_stencil<RECON>(q3, c, st, w3);
      if constexpr (FLUID == 0) {
        load_stencil<RECON>(qe, c, st, we);
        constexpr int R = (RECON == 2) ? 3 : ((RECON == 1) ? 2 : 1);
#pragma unroll
        for (int m = -R; m <= R; ++m) wp[3 + m] = amax(0.0, P.gm1 * wd[3 + m] * we[3 + m]);
      }
      if constexpr (RECON == 1 && !CURV) {
        solve_pair_fast<FLUID, RIEMANN>(P, dir - 1, wd, w1, w2, w3, wp, we, lo, up);
      } else {
        lo = face_of_cell<FLUID, RIEMANN, RECON, CURV>(P, f, a.in, b, n, dir, 0, k, j, i, wd, w1, w2, w3,
                                                       wp, we);
        up = face_of_cell<FLUID, RIEMANN, RECON, CURV>(P, f, a.in, b, n, dir, 1, k, j, i, wd, w1, w2, w3,
                                                       wp, we);
      }
      if constexpr (FIX) {
        if ((fixed_faces >> (2 * (dir - 1))) & 1u) load_corrected<FLUID>(f, b, nv, ns, n, dir - 1, c, lo);
        if ((fixed_faces >> (2 * (dir - 1) + 1)) & 1u) load_corrected<FLUID>(f, b, nv, ns, n, dir - 1, c + st, up);
      }
      }
      const double *ax = (dir == 1) ? g.ax1 : ((dir == 2) ? g.ax2 : g.ax3);
      const int d = dir - 1;
      // momentum components in global order: component (d+q)%3 carries the sweep's q-th flux
      double lom[3], upm[3];
      lom[d] = lo.fmx, lom[(d + 1) % 3] = lo.fmy, lom[(d + 2) % 3] = lo.fmz;
      upm[d] = up.fmx, upm[(d + 1) % 3] = up.fmy, upm[(d + 2) % 3] = up.fmz;
      if (dir == 1) {
        divf[0] = (ax[0] * lo.fd - ax[1] * up.fd);
        divf[1] = (ax[0] * lom[0] - ax[1] * upm[0]);
        divf[2] = (ax[0] * lom[1] - ax[1] * upm[1]);
        divf[3] = (ax[0] * lom[2] - ax[1] * upm[2]);
        if constexpr (FLUID == 0) {
          divf[4] = (ax[0] * lo.fe - ax[1] * up.fe);
          divf[5] = (ax[0] * lo.feg - ax[1] * up.feg);
        }
      } else {
        divf[0] += (ax[0] * lo.fd - ax[1] * up.fd);
        divf[1] += (ax[0] * lom[0] - ax[1] * upm[0]);
        divf[2] += (ax[0] * lom[1] - ax[1] * upm[1]);
        divf[3] += (ax[0] * lom[2] - ax[1] * upm[2]);
        if constexpr (FLUID == 0) {
          divf[4] += (ax[0] * lo.fe - ax[1] * up.fe);
          divf[5] += (ax[0] * lo.feg - ax[1] * up.feg);
        }
      }
      if constexpr (FLUID == 0) plo[d] = lo.pf, pup[d] = up.pf, vlo[d] = lo.vf, vup[d] = up.vf;
      if constexpr (EXTRA) mlo[d] = lo.fd, mup[d] = up.fd;
    }
    // ---- ApplyUpdate (artemis_integrator.hpp:104-106) on u0 = PrimToCons(in), u1 = PrimToCons(u1)
    const double *rd = a.u1[b * nv + n], *r1 = a.u1[b * nv + ns + 3 * n + 0];
    const double *r2 = a.u1[b * nv + ns + 3 * n + 1], *r3 = a.u1[b * nv + ns + 3 * n + 2];
    if constexpr (FLUID == 0) {
      const double *re = a.u1[b * nv + 5 * ns + n];
      GasCons u0, u1;
      if constexpr (STORED) {
        auto ld = [&](double *const *t, GasCons &u) {
          u.d = t[b * nv + n][c], u.m1 = t[b * nv + ns + 3 * n + 0][c], u.m2 = t[b * nv + ns + 3 * n + 1][c];
          u.m3 = t[b * nv + ns + 3 * n + 2][c], u.e = t[b * nv + 4 * ns + n][c], u.eg = t[b * nv + 5 * ns + n][c];
        };
        ld(f.cons0, u0), ld(f.cons1, u1);
      } else {
        u0 = prim_to_cons_gas(f, w.rho, w.v1, w.v2, w.v3, w.sie, hx);
        u1 = prim_to_cons_gas(f, rd[c], r1[c], r2[c], r3[c], re[c], hx);
      }
      u0.d = a.gam0 * u0.d + a.gam1 * u1.d + divf[0] * a.beta_dt / g.vol;
      u0.m1 = a.gam0 * u0.m1 + a.gam1 * u1.m1 + divf[1] * a.beta_dt / g.vol;
      u0.m2 = a.gam0 * u0.m2 + a.gam1 * u1.m2 + divf[2] * a.beta_dt / g.vol;
      u0.m3 = a.gam0 * u0.m3 + a.gam1 * u1.m3 + divf[3] * a.beta_dt / g.vol;
      u0.e = a.gam0 * u0.e + a.gam1 * u1.e + divf[4] * a.beta_dt / g.vol;
      u0.eg = a.gam0 * u0.eg + a.gam1 * u1.eg + divf[5] * a.beta_dt / g.vol;
      // ---- FluxSource (fluid_fluxes.hpp:361-415)
      const double dt = a.bdt;
      u0.m1 += dt / g.dx[0] * (plo[0] - pup[0]);
      u0.eg -= dt / g.vol * 0.5 * (plo[0] + pup[0]) * (g.ax1[1] * vup[0] - g.ax1[0] * vlo[0]);
      if (multi_d) {
        u0.m2 += dt / g.dx[1] * (plo[1] - pup[1]);
        u0.eg -= dt / g.vol * 0.5 * (plo[1] + pup[1]) * (g.ax2[1] * vup[1] - g.ax2[0] * vlo[1]);
      }
      if (three_d) {
        u0.m3 += dt / g.dx[2] * (plo[2] - pup[2]);
        u0.eg -= dt / g.vol * 0.5 * (plo[2] + pup[2]) * (g.ax3[1] * vup[2] - g.ax3[0] * vlo[2]);
      }
      if constexpr (CURV) {
        const double rdt = w.rho * dt;
        double vf[3];
        rotation_velocity(co, P.omf, vf);
        if (co.x1dep())
          u0.m1 += rdt * (0.0 * sqr(w.v1 + vf[0]) + co.dh2dx1() * sqr(w.v2 + vf[1]) + co.dh3dx1() * sqr(w.v3 + vf[2]));
        if (co.x2dep() && multi_d)
          u0.m2 += rdt * (0.0 * sqr(w.v1 + vf[0]) + 0.0 * sqr(w.v2 + vf[1]) + co.dh3dx2() * sqr(w.v3 + vf[2]));
      }
      if constexpr (EXTRA) {
        if (a.diff_on) { // Gas::DiffusionUpdate (artemis_driver.cpp:218-221)
          const double v[3] = {w.v1, w.v2, w.v3};
          double dm[3], de, deg;
          if (a.dsum) { // the five sums, formed by artemis_hip_viscous_source for this stage's input primitives
            dm[0] = a.dsum[b * 5 + 0][c], dm[1] = a.dsum[b * 5 + 1][c], dm[2] = a.dsum[b * 5 + 2][c];
            de = a.dsum[b * 5 + 3][c], deg = a.dsum[b * 5 + 4][c];
          } else {
            diffusion_update_cell(P, dcell, b, n, c, a.do_viscosity, dt, v, dm, de, deg);
          }
          u0.m1 -= dm[0], u0.m2 -= dm[1], u0.m3 -= dm[2];
          u0.e -= de;
          u0.eg -= deg;
        }
      }
      if (a.grav_on) gravity_gas(ga, dt, hx, w, u0);
      if (a.nb_n) { // nbody_device.hpp: every coupled particle in order on the registers
        const double wv[4] = {w.rho, w.v1, w.v2, w.v3};
        double u[6] = {u0.d, u0.m1, u0.m2, u0.m3, u0.e, u0.eg};
        nb_apply<true>(a.nb_pl, a.nb_n, make_coords(P, b, k, j, i), a.nb_omf, dt, wv, u);
        u0.d = u[0], u0.m1 = u[1], u0.m2 = u[2], u0.m3 = u[3], u0.e = u[4], u0.eg = u[5];
      }
      if (a.rf_on) shear_gas(sa, dt, w, u0);
      if constexpr (EXTRA) {
        if (a.rfc_on) {
          const double ax2[2] = {multi_d ? g.ax2[0] : 0.0, multi_d ? g.ax2[1] : 0.0};
          const double ax3[2] = {three_d ? g.ax3[0] : 0.0, three_d ? g.ax3[1] : 0.0};
          rotating_frame_gas(rfc, multi_d, three_d, mlo, mup, g.ax1, ax2, ax3, g.vol, u0);
        }
      }
      if (a.to_cons) {
        f.cons0[b * nv + n][c] = u0.d;
        f.cons0[b * nv + ns + 3 * n + 0][c] = u0.m1, f.cons0[b * nv + ns + 3 * n + 1][c] = u0.m2;
        f.cons0[b * nv + ns + 3 * n + 2][c] = u0.m3;
        f.cons0[b * nv + 4 * ns + n][c] = u0.e, f.cons0[b * nv + 5 * ns + n][c] = u0.eg;
        continue;
      }
      if constexpr (EXTRA) {
        if (a.cool_on) cooling_gas(f, a.cool.cv, cool_omdt, cool_T0, cool_beta, hx, u0); // :243-248
      }
      // ---- SetAuxillaryFields (fill_derived.cpp:58-71) + ConsToPrim (:132-146)
      const double u_d = (u0.d > f.dfloor) ? u0.d : f.dfloor;
      const double u_d2 = amax(u0.d, f.dfloor);
      const double rv1 = u0.m1 / hx[0], rv2 = u0.m2 / hx[1], rv3 = u0.m3 / hx[2];
      const double ke = 0.5 * (sqr(rv1) + sqr(rv2) + sqr(rv3)) / u_d2;
      const double ue_cons = u0.e - ke;
      double sie = (ue_cons > f.de_switch * u0.e) ? ue_cons / u_d2 : u0.eg / u_d2;
      sie = amax(sie, f.siefloor);
      double u_u = sie * u_d;
      const double uflr = f.siefloor * u_d;
      u_u = (u_u > uflr) ? u_u : uflr;
      const double w_d = (u0.d > f.dfloor) ? u0.d : f.dfloor;
      a.out[b * nv + n][c] = w_d;
      a.out[b * nv + ns + 3 * n + 0][c] = u0.m1 / (w_d * hx[0]);
      a.out[b * nv + ns + 3 * n + 1][c] = u0.m2 / (w_d * hx[1]);
      a.out[b * nv + ns + 3 * n + 2][c] = u0.m3 / (w_d * hx[2]);
      const double w_s = u_u / w_d;
      a.out[b * nv + 5 * ns + n][c] = (w_s > f.siefloor) ? w_s : f.siefloor;
    } else {
      DustCons u0, u1;
      if constexpr (STORED) {
        auto ld = [&](double *const *t, DustCons &u) {
          u.d = t[b * nv + n][c], u.m1 = t[b * nv + ns + 3 * n + 0][c], u.m2 = t[b * nv + ns + 3 * n + 1][c];
          u.m3 = t[b * nv + ns + 3 * n + 2][c];
        };
        ld(f.cons0, u0), ld(f.cons1, u1);
      } else {
        u0 = prim_to_cons_dust(f, w.rho, w.v1, w.v2, w.v3, hx);
        u1 = prim_to_cons_dust(f, rd[c], r1[c], r2[c], r3[c], hx);
      }
      u0.d = a.gam0 * u0.d + a.gam1 * u1.d + divf[0] * a.beta_dt / g.vol;
      u0.m1 = a.gam0 * u0.m1 + a.gam1 * u1.m1 + divf[1] * a.beta_dt / g.vol;
      u0.m2 = a.gam0 * u0.m2 + a.gam1 * u1.m2 + divf[2] * a.beta_dt / g.vol;
      u0.m3 = a.gam0 * u0.m3 + a.gam1 * u1.m3 + divf[3] * a.beta_dt / g.vol;
      const double dt = a.bdt;
      if constexpr (CURV) { // Dust::FluxSource (dust.cpp:303-326): coordinate source only
        const double rdt = w.rho * dt;
        double vf[3];
        rotation_velocity(co, P.omf, vf);
        if (co.x1dep())
          u0.m1 += rdt * (0.0 * sqr(w.v1 + vf[0]) + co.dh2dx1() * sqr(w.v2 + vf[1]) + co.dh3dx1() * sqr(w.v3 + vf[2]));
        if (co.x2dep() && multi_d)
          u0.m2 += rdt * (0.0 * sqr(w.v1 + vf[0]) + 0.0 * sqr(w.v2 + vf[1]) + co.dh3dx2() * sqr(w.v3 + vf[2]));
      }
      if (a.grav_on) gravity_dust(ga, dt, hx, w, u0);
      if (a.nb_n) {
        const double wv[4] = {w.rho, w.v1, w.v2, w.v3};
        double u[4] = {u0.d, u0.m1, u0.m2, u0.m3};
        nb_apply<false>(a.nb_pl, a.nb_n, make_coords(P, b, k, j, i), a.nb_omf, dt, wv, u);
        u0.d = u[0], u0.m1 = u[1], u0.m2 = u[2], u0.m3 = u[3];
      }
      if (a.rf_on) shear_dust(sa, dt, w, u0);
      if constexpr (EXTRA) {
        if (a.rfc_on) {
          const double ax2[2] = {multi_d ? g.ax2[0] : 0.0, multi_d ? g.ax2[1] : 0.0};
          const double ax3[2] = {three_d ? g.ax3[0] : 0.0, three_d ? g.ax3[1] : 0.0};
          rotating_frame_dust(rfc, multi_d, three_d, mlo, mup, g.ax1, ax2, ax3, g.vol, u0);
        }
      }
      if (a.to_cons) {
        f.cons0[b * nv + n][c] = u0.d;
        f.cons0[b * nv + ns + 3 * n + 0][c] = u0.m1, f.cons0[b * nv + ns + 3 * n + 1][c] = u0.m2;
        f.cons0[b * nv + ns + 3 * n + 2][c] = u0.m3;
        continue;
      }
      const double w_d = (u0.d > f.dfloor) ? u0.d : f.dfloor; // ConsToPrim (fill_derived.cpp:155-164)
      a.out[b * nv + n][c] = w_d;
      a.out[b * nv + ns + 3 * n + 0][c] = u0.m1 / (w_d * hx[0]);
      a.out[b * nv + ns + 3 * n + 1][c] = u0.m2 / (w_d * hx[1]);
      a.out[b * nv + ns + 3 * n + 2][c] = u0.m3 / (w_d * hx[2]);
    }
  }
}

template <int FLUID, int RIEMANN, int RECON>
void launch_geom(const PackView &P, const CellStageArgs &a, hipStream_t s) {
  const dim3 grid = interior_grid(P);
  const bool extra = a.diff_on || a.rfc_on || a.cool_on;
  if (P.coords == ARTEMIS_CARTESIAN) {
    if (extra) hipLaunchKernelGGL((stage_cell_kernel<FLUID, RIEMANN, RECON, false, true>), grid, tile_threads(P.ie - P.is + 1), 0, s, P, a);
    else hipLaunchKernelGGL((stage_cell_kernel<FLUID, RIEMANN, RECON, false, false>), grid, tile_threads(P.ie - P.is + 1), 0, s, P, a);
  } else {
    if (extra) hipLaunchKernelGGL((stage_cell_kernel<FLUID, RIEMANN, RECON, true, true>), grid, tile_threads(P.ie - P.is + 1), 0, s, P, a);
    else hipLaunchKernelGGL((stage_cell_kernel<FLUID, RIEMANN, RECON, true, false>), grid, tile_threads(P.ie - P.is + 1), 0, s, P, a);
  }
}
template <int FLUID, int RIEMANN>
void launch_recon(const PackView &P, int recon, const CellStageArgs &a, hipStream_t s) {
  if (recon == ARTEMIS_PCM) launch_geom<FLUID, RIEMANN, 0>(P, a, s);
  else if (recon == ARTEMIS_PLM) launch_geom<FLUID, RIEMANN, 1>(P, a, s);
  else launch_geom<FLUID, RIEMANN, 2>(P, a, s);
}
// The arguments every form of the cell stage derives from the C ABI's (to_cons = 0, no zone list)
CellStageArgs cell_args(const PackView &P, const artemis_stage_general_args_t &g) {
  CellStageArgs a;
  a.gam0 = g.gam0, a.gam1 = g.gam1, a.beta_dt = g.beta_dt, a.bdt = g.bdt;
  a.bdt_ptr = g.beta_dt_dev;
  a.in = a.u1 = a.out = nullptr;
  a.to_cons = 0;
  a.grav_on = (g.gravity && (g.time >= g.gravity->tstart) && (g.time < g.gravity->tstop)) ? 1 : 0;
  if (a.grav_on) a.grav = *g.gravity;
  const bool cart = (P.coords == ARTEMIS_CARTESIAN);
  a.rf_on = (g.rf_omega != 0.0) && cart, a.rf_omega = g.rf_omega, a.rf_qshear = g.rf_qshear;
  a.rfc_on = (g.rf_omega != 0.0) && !cart;
  a.diff_on = (g.diffusion != nullptr) && P.gas.ns > 0;
  a.do_viscosity = (g.diffusion && g.diffusion->visc.type != ARTEMIS_DIFF_OFF) ? 1 : 0;
  a.dsum = (a.diff_on && P.gas.ns == 1) ? g.diffusion_sums : nullptr;
  a.cool_on = (g.cooling != nullptr) && P.gas.ns > 0;
  if (a.cool_on) a.cool = *g.cooling;
  a.fix = nullptr, a.nfix = 0;
  a.nb_pl = g.nbody_dev, a.nb_n = g.nbody_n, a.nb_omf = g.nbody_omf;
  return a;
}

// ---- refined meshes: the fine side's faces on coarse-fine boundaries (artemis_hip_ml_face_fluxes) -----------------
// One workgroup per box; a thread solves the lower face of direction bx.dir of a zone of the box from register
// stencils of the `prim` tables (pressure recomputed like the stage kernels: the pressure slot of ghost zones is
// not maintained on the one-kernel paths) with face_of_cell -- the per-task flux kernel's expression trees -- and
// stores the task's outputs for that face.
template <int FLUID, int RIEMANN, int RECON, bool CURV>
__global__ __launch_bounds__(256) void ml_face_flux_kernel(const PackView P, double *const *prim,
                                                           const artemis_ml_face_box_t *__restrict__ boxes) {
  const artemis_ml_face_box_t bx = boxes[blockIdx.x];
  const FluidView &f = (FLUID == 0) ? P.gas : P.dust;
  const int ns = f.ns, nv = (FLUID == 0 ? 6 : 4) * ns;
  const int b = bx.block, dir = bx.dir + 1, d = bx.dir;
  const long st = (dir == 1) ? 1 : ((dir == 2) ? P.sj : P.sk);
  const long ncell = static_cast<long>(bx.n[0]) * bx.n[1] * bx.n[2];
  constexpr int R = (RECON == 2) ? 3 : ((RECON == 1) ? 2 : 1);
  for (long t = threadIdx.x; t < ncell; t += blockDim.x) {
    const int i = bx.lo[0] + static_cast<int>(t % bx.n[0]), j = bx.lo[1] + static_cast<int>((t / bx.n[0]) % bx.n[1]);
    const int k = bx.lo[2] + static_cast<int>(t / (static_cast<long>(bx.n[0]) * bx.n[1]));
    const long c = (static_cast<long>(k) * P.nj + j) * P.ni + i;
    for (int n = 0; n < ns; ++n) {
      const double *qd = prim[b * nv + n], *q1 = prim[b * nv + ns + 3 * n + 0];
      const double *q2 = prim[b * nv + ns + 3 * n + 1], *q3 = prim[b * nv + ns + 3 * n + 2];
      const double *qe = (FLUID == 0) ? prim[b * nv + 5 * ns + n] : nullptr;
      double wd[7], w1[7], w2[7], w3[7], wp[7], we[7];
#pragma unroll
      for (int m = 0; m < 7; ++m) wd[m] = w1[m] = w2[m] = w3[m] = wp[m] = we[m] = 0.0;
#pragma unroll
      for (int m = -R; m < R; ++m) { // the R zones on either side of the face (zone c is the one above it)
        wd[3 + m] = qd[c + m * st], w1[3 + m] = q1[c + m * st], w2[3 + m] = q2[c + m * st], w3[3 + m] = q3[c + m * st];
        if constexpr (FLUID == 0) {
          we[3 + m] = qe[c + m * st];
          wp[3 + m] = amax(0.0, P.gm1 * wd[3 + m] * we[3 + m]);
        }
      }
      const FaceFlux F = face_of_cell<FLUID, RIEMANN, RECON, CURV>(P, f, prim, b, n, dir, 0, k, j, i, wd, w1, w2, w3, wp, we);
      const int m0 = b * nv + ns + 3 * n;
      f.flux[d][b * nv + n][c] = F.fd;
      f.flux[d][m0 + d][c] = F.fmx, f.flux[d][m0 + (d + 1) % 3][c] = F.fmy, f.flux[d][m0 + (d + 2) % 3][c] = F.fmz;
      if constexpr (FLUID == 0) {
        f.flux[d][b * nv + 4 * ns + n][c] = F.fe, f.flux[d][b * nv + 5 * ns + n][c] = F.feg;
        f.pflux[d][b * ns + n][c] = F.pf;
        f.vface[d][b * ns + n][c] = F.vf;
      }
    }
  }
}

template <int FLUID, int RIEMANN, int RECON>
void launch_faces_geom(const PackView &P, double *const *prim, const artemis_ml_face_box_t *boxes, int nboxes, hipStream_t s) {
  if (P.coords == ARTEMIS_CARTESIAN)
    hipLaunchKernelGGL((ml_face_flux_kernel<FLUID, RIEMANN, RECON, false>), dim3(nboxes), dim3(256), 0, s, P, prim, boxes);
  else
    hipLaunchKernelGGL((ml_face_flux_kernel<FLUID, RIEMANN, RECON, true>), dim3(nboxes), dim3(256), 0, s, P, prim, boxes);
}
template <int FLUID, int RIEMANN>
void launch_faces_recon(const PackView &P, int recon, double *const *prim, const artemis_ml_face_box_t *boxes, int nboxes,
                        hipStream_t s) {
  if (recon == ARTEMIS_PCM) launch_faces_geom<FLUID, RIEMANN, 0>(P, prim, boxes, nboxes, s);
  else if (recon == ARTEMIS_PLM) launch_faces_geom<FLUID, RIEMANN, 1>(P, prim, boxes, nboxes, s);
  else launch_faces_geom<FLUID, RIEMANN, 2>(P, prim, boxes, nboxes, s);
}

template <int FLUID, int RIEMANN, int RECON>
void launch_fix_geom(const PackView &P, const CellStageArgs &a, hipStream_t s) {
  const dim3 grid((a.nfix + TX * TY - 1) / (TX * TY)), block(TX * TY);
  if (P.coords == ARTEMIS_CARTESIAN)
    hipLaunchKernelGGL((stage_cell_kernel<FLUID, RIEMANN, RECON, false, true, false, true>), grid, block, 0, s, P, a);
  else
    hipLaunchKernelGGL((stage_cell_kernel<FLUID, RIEMANN, RECON, true, true, false, true>), grid, block, 0, s, P, a);
}
template <int FLUID, int RIEMANN>
void launch_fix_recon(const PackView &P, int recon, const CellStageArgs &a, hipStream_t s) {
  if (recon == ARTEMIS_PCM) launch_fix_geom<FLUID, RIEMANN, 0>(P, a, s);
  else if (recon == ARTEMIS_PLM) launch_fix_geom<FLUID, RIEMANN, 1>(P, a, s);
  else launch_fix_geom<FLUID, RIEMANN, 2>(P, a, s);
}
} // namespace

void launch_ml_face_fluxes(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas, int riemann_gas,
                           int recon_dust, int riemann_dust, const artemis_ml_face_box_t *boxes, int nboxes, hipStream_t s) {
  if (nboxes <= 0) return;
  if (P.gas.ns) {
    const int recon = g.pcm ? ARTEMIS_PCM : recon_gas;
    if (riemann_gas == ARTEMIS_HLLC) launch_faces_recon<0, 0>(P, recon, g.gas_in, boxes, nboxes, s);
    else if (riemann_gas == ARTEMIS_HLLE) launch_faces_recon<0, 1>(P, recon, g.gas_in, boxes, nboxes, s);
    else launch_faces_recon<0, 2>(P, recon, g.gas_in, boxes, nboxes, s);
  }
  if (P.dust.ns) {
    const int recon = g.pcm ? ARTEMIS_PCM : recon_dust;
    if (riemann_dust == ARTEMIS_HLLE) launch_faces_recon<1, 1>(P, recon, g.dust_in, boxes, nboxes, s);
    else launch_faces_recon<1, 2>(P, recon, g.dust_in, boxes, nboxes, s);
  }
}

// The listed zones once more, flagged faces from the corrected flux arrays (FIX above): one launch per fluid
void launch_ml_stage_fixup(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas, int riemann_gas,
                           int recon_dust, int riemann_dust, const artemis_ml_fix_cell_t *cells, int ncells, hipStream_t s) {
  if (ncells <= 0) return;
  CellStageArgs a = cell_args(P, g);
  a.fix = cells, a.nfix = ncells;
  a.to_cons = g.defer_finish ? 1 : 0;
  if (P.gas.ns) {
    a.in = g.gas_in, a.u1 = g.gas_u1, a.out = g.gas_out;
    const int recon = g.pcm ? ARTEMIS_PCM : recon_gas;
    if (riemann_gas == ARTEMIS_HLLC) launch_fix_recon<0, 0>(P, recon, a, s);
    else if (riemann_gas == ARTEMIS_HLLE) launch_fix_recon<0, 1>(P, recon, a, s);
    else launch_fix_recon<0, 2>(P, recon, a, s);
  }
  if (P.dust.ns) {
    a.in = g.dust_in, a.u1 = g.dust_u1, a.out = g.dust_out;
    const int recon = g.pcm ? ARTEMIS_PCM : recon_dust;
    if (riemann_dust == ARTEMIS_HLLE) launch_fix_recon<1, 1>(P, recon, a, s);
    else launch_fix_recon<1, 2>(P, recon, a, s);
  }
}

// The cell-local remainder of a stage over stored fluxes (see STORED above): one kernel per fluid
void launch_stage_epilogue(const PackView &P, const artemis_stage_general_args_t &g, hipStream_t s, bool to_cons) {
  CellStageArgs a = cell_args(P, g);
  a.to_cons = to_cons ? 1 : 0; // artemis_hip_stage_epilogue_cons: stop after the sources, state left in cons0
  const bool cart = (P.coords == ARTEMIS_CARTESIAN);
  const dim3 grid = interior_grid(P);
  if (P.gas.ns) {
    a.in = a.u1 = a.out = P.gas.prim;
    if (cart) hipLaunchKernelGGL((stage_cell_kernel<0, 0, 0, false, true, true>), grid, tile_threads(P.ie - P.is + 1), 0, s, P, a);
    else hipLaunchKernelGGL((stage_cell_kernel<0, 0, 0, true, true, true>), grid, tile_threads(P.ie - P.is + 1), 0, s, P, a);
  }
  if (P.dust.ns) {
    a.in = a.u1 = a.out = P.dust.prim;
    if (cart) hipLaunchKernelGGL((stage_cell_kernel<1, 1, 0, false, true, true>), grid, tile_threads(P.ie - P.is + 1), 0, s, P, a);
    else hipLaunchKernelGGL((stage_cell_kernel<1, 1, 0, true, true, true>), grid, tile_threads(P.ie - P.is + 1), 0, s, P, a);
  }
}

int stage_general_variant(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas, int riemann_gas,
                          int recon_dust, int riemann_dust) {
  if (!opt(OPT_NO_STAGE2D) && stage2d_covers(P, g, recon_gas, riemann_gas, recon_dust, riemann_dust)) return 1;
  if (!opt(OPT_NO_FUSED_CURV)) {
    if (curv_march_covers(P, g, recon_gas)) return 3;
    if (fused_curv_covers(P, g, recon_gas)) return 2;
  }
  return 0;
}

void launch_stage_cell(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas,
                       int riemann_gas, int recon_dust, int riemann_dust, hipStream_t s) {
  // 2-D Cartesian gas (+ <= 2 dust species) with the pointwise sources: the row-march kernel does the whole
  // stage of both fluids, drag, aux, c2p and dt in one pass (kernels_stage2d.hip; same bits)
  const int variant = stage_general_variant(P, g, recon_gas, riemann_gas, recon_dust, riemann_dust);
  if (variant == 1) {
    launch_stage2d(P, g, recon_gas, riemann_gas, riemann_dust, s);
    return;
  }
  if (variant == 2) { // curvilinear gas with diffusion from stored flux arrays: the older march, geometry in registers
    launch_stage_fused_curv(P, g, recon_gas, riemann_gas, s);
    return;
  }
  // defer_finish: 1 = stop at the conserved state (the caller finishes every zone); 2 = finish every zone here, the caller
  // re-finishes its listed fix-up zones (artemis_hip_stage_finish_cells); 0 = no fix-up follows
  const bool defer = g.defer_finish == 1;
  const bool to_cons = g.drag || defer;
  CellStageArgs a = cell_args(P, g);
  a.to_cons = to_cons ? 1 : 0;
  if (variant == 3) { // curvilinear gas: the streaming tile march with its geometry in LDS tables (kernels_curv.hip)
    launch_stage_curv(P, g, 0, recon_gas, riemann_gas, s);
  } else if (P.gas.ns) {
    a.in = g.gas_in, a.u1 = g.gas_u1, a.out = g.gas_out;
    const int recon = g.pcm ? ARTEMIS_PCM : recon_gas;
    if (riemann_gas == ARTEMIS_HLLC) launch_recon<0, 0>(P, recon, a, s);
    else if (riemann_gas == ARTEMIS_HLLE) launch_recon<0, 1>(P, recon, a, s);
    else launch_recon<0, 2>(P, recon, a, s);
  }
  const bool dust_march = variant == 3 && curv_march_covers_dust(P, g, recon_dust, riemann_dust);
  // one dust species coupled by simple_dust drag: the dust march does the coupled update, SetAuxillaryFields and
  // ConsToPrim of both fluids on its registers (no conserved round trip of the dust, no finish launch)
  const bool finish_in_march = dust_march && g.drag && !defer && !opt(OPT_NO_DRAG_IN_MARCH) && drag_finish_in_march(P, *g.drag);
  if (dust_march) { // the dust species on the same march (kernels_curv.hip, DUST instantiations)
    launch_stage_curv(P, g, 1, recon_dust, riemann_dust, s, finish_in_march);
  } else if (P.dust.ns) {
    a.in = g.dust_in, a.u1 = g.dust_u1, a.out = g.dust_out;
    const int recon = g.pcm ? ARTEMIS_PCM : recon_dust;
    if (riemann_dust == ARTEMIS_HLLE) launch_recon<1, 1>(P, recon, a, s);
    else launch_recon<1, 2>(P, recon, a, s);
  }
  if (defer) return; // the caller finishes (artemis_hip_stage_finish) once its fix-up has run
  if (finish_in_march) return; // (both fluids' primitives and timestep limits are done)
  PackView Q = P; // the new state: prim tables are the out tables
  Q.gas.prim = g.gas_out, Q.dust.prim = g.dust_out;
  if (g.drag) { // coupled update on cons0, then SetAuxillaryFields and ConsToPrim into the out tables
    if (!launch_drag_finish(Q, *g.drag, g.bdt, g.beta_dt_dev, s)) { // one pass when one gas species is coupled
      launch_drag_source(Q, *g.drag, g.bdt, g.beta_dt_dev, s);
      if (Q.gas.ns) launch_set_aux(Q, s);
      launch_cons_to_prim(Q, s);
    }
  }
  if (g.dt_dev) { // EstimateTimestepMesh of the new state (gas.cpp:411-433, dust.cpp:256-272)
    if (Q.gas.ns && !(variant == 3 && !to_cons)) launch_estimate_dt(Q, ARTEMIS_GAS, g.cfl_gas, g.dt_dev, s); // (the march has its own)
    if (Q.dust.ns && !(dust_march && !to_cons)) launch_estimate_dt(Q, ARTEMIS_DUST, g.cfl_dust, g.dt_dev, s);
  }
}

} // namespace artemis
