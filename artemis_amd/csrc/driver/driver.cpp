// Host driver: the stand-in for Parthenon + ArtemisDriver<GEOM> when the hydro path runs
// outside Artemis.  Plain C++ (no HIP here): device work goes through include/artemis_hip.h
// and include/artemis_rt.h only.  Structure and names follow the reference:
//   ProcessPackages      artemis.cpp:37-164   (physics switches, Gas/Dust::Initialize params)
//   ProblemGenerator     pgen/pgen.hpp:38-64  (blast, linear_wave, advection)
//   PostInitialization   derived/fill_derived.cpp:284-287
//   Step / StepTasks     artemis_driver.cpp:102-273
//   PostStepTasks        artemis_driver.cpp:279-297 (EstimateTimestep)
//   Execute loop / SetGlobalTimeStep   parthenon EvolutionDriver (upstream, recalled)
#include <algorithm>
#include <array>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "artemis_driver.h"
#include "../options.hpp"
#include "artemis_hip.h"
#include "artemis_rt.h"
#include "../geometry_core.hpp"
#include "parameter_input.hpp"
#include "block_tree.hpp"

#define SQR(x) ((x) * (x))
typedef double Real;

namespace {
thread_local std::string g_sim_err;

struct HipFail : std::runtime_error {
  using std::runtime_error::runtime_error;
};
void CK(int rc, const char *what) {
  if (rc != 0) throw HipFail(std::string(what) + ": " + artemis_hip_last_error());
}

// Device buffer of doubles owned by the driver
struct DevBuf {
  double *p = nullptr;
  size_t n = 0;
  void alloc(size_t count) {
    release();
    n = count;
    p = static_cast<double *>(artemis_rt_malloc(std::max<size_t>(count, 1) * sizeof(double)));
    if (!p) throw HipFail(std::string("device allocation failed: ") + artemis_hip_last_error());
    CK(artemis_rt_memset(p, 0, std::max<size_t>(count, 1) * sizeof(double), nullptr), "memset");
    // (buffers come back from artemis_rt's cache: the zeroing, on the null stream, is complete before any of the
    //  driver's non-blocking streams touches the buffer)
    CK(artemis_rt_stream_sync(nullptr), "sync");
  }
  void release() {
    if (p) artemis_rt_free(p);
    p = nullptr, n = 0;
  }
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
};

// A field group: [nb][nvar][N] doubles + the device table of [nb*nvar] pointers into it.
struct Field {
  DevBuf data;
  void *table = nullptr; // device array of double*
  int nb = 0, nvar = 0;
  size_t N = 0;
  bool ok() const { return data.p != nullptr; }
  // refined meshes: the ghost zones behind `ic` faces hold the initial state (they are constants of the run and nothing
  // but the boundary fill writes them: the block-graph exchange only fills zones that face a neighbour)
  bool ic_filled = false;
  void alloc(int nb_, int nvar_, size_t N_) {
    nb = nb_, nvar = nvar_, N = N_;
    ic_filled = false;
    if (nvar == 0) return;
    data.alloc(static_cast<size_t>(nb) * nvar * N);
    std::vector<double *> h(static_cast<size_t>(nb) * nvar);
    for (size_t q = 0; q < h.size(); ++q) h[q] = data.p + q * N;
    table = artemis_rt_malloc(h.size() * sizeof(double *));
    if (!table) throw HipFail("table allocation failed");
    CK(artemis_rt_memcpy_h2d(table, h.data(), h.size() * sizeof(double *), nullptr), "h2d");
    CK(artemis_rt_device_sync(), "sync");
  }
  // Rows for the blocks of `need` only; every other block's table entries point at one shared row (row 0) that nothing
  // reads: flux arrays of a refined mesh on the one-kernel stages, which are touched on coarse-fine faces alone.
  std::vector<int> slot;
  void alloc_sparse(int nb_, int nvar_, size_t N_, const std::vector<char> &need) {
    nb = nb_, nvar = nvar_, N = N_;
    if (nvar == 0) return;
    slot.assign(nb, 0);
    int rows = 1;
    for (int b = 0; b < nb; ++b)
      if (need[b]) slot[b] = rows++;
    data.alloc(static_cast<size_t>(rows) * nvar * N);
    std::vector<double *> h(static_cast<size_t>(nb) * nvar);
    for (int b = 0; b < nb; ++b)
      for (int v = 0; v < nvar; ++v) h[static_cast<size_t>(b) * nvar + v] = var(b, v);
    table = artemis_rt_malloc(h.size() * sizeof(double *));
    if (!table) throw HipFail("table allocation failed");
    CK(artemis_rt_memcpy_h2d(table, h.data(), h.size() * sizeof(double *), nullptr), "h2d");
    CK(artemis_rt_device_sync(), "sync");
  }
  double *const *tab() const { return static_cast<double *const *>(table); }
  double *var(int b, int v) const {
    const size_t row = slot.empty() ? static_cast<size_t>(b) : static_cast<size_t>(slot[b]);
    return data.p + (row * nvar + v) * N;
  }
  void release() {
    data.release();
    if (table) artemis_rt_free(table);
    table = nullptr;
    slot.clear();
  }
  ~Field() {
    if (table) artemis_rt_free(table);
  }
};

struct Block {
  int level = 0;        // refinement level (0 on a uniform mesh)
  int lx[3];            // logical location in the block grid of its level
  Real xmin[3], xmax[3];
  int bc[6];            // artemis_bc per face (NONE where a neighbour block exists)
  int nbr_rank[6], nbr_block[6];
  long gid;             // global block id
};

struct Link { // one directed ghost-slab transfer out of local block b through face f
  int b, face;
  int nbr_rank, nbr_block; // destination (its face is face^1)
  long count;
  DevBuf sbuf, rbuf;       // rbuf used for remote links only
  int tag_send, tag_recv;
};

constexpr size_t kMaxTimedLaunches = 8192; // HIP event pairs kept per timed evolve()

enum { PG_BLAST, PG_LINWAVE, PG_ADVECTION, PG_CONSTANT, PG_STRAT, PG_BUMP, PG_COND, PG_DISK };

} // namespace

struct artemis_sim_impl {
  artemis_host::ParameterInput pin;
  artemis_comm_t comm;
  bool has_comm = false;
  int rank = 0, nranks = 1;

  // multilevel (static mesh refinement): block tree, operation lists, coarse buffers (block_tree.hpp,
  // include/artemis_hip.h "multilevel block-graph data path")
  bool multilevel = false;
  struct Region {
    int level;
    double lo[3], hi[3];
  };
  std::vector<Region> regions;
  long nblocks_global = 0;
  // adaptive refinement (<parthenon/mesh> refinement = adaptive, numlevel, derefine_count; <gas> refine_field,
  // refine_type, refine_thr, deref_thr: gas.cpp:305-380).  The block tree of an adaptive run changes between cycles:
  // the C handle then builds a new state on the new leaves (`forced_leaves`) and adopts the old one's data.
  bool adaptive = false;
  int amr_max_level = 0, derefine_count = 10;
  int refine_field = 0; // 0 none, 1 gas density, 2 gas pressure
  int refine_type = 0;  // 1 gradient (ScalarFirstDerivative), 2 magnitude (ScalarMagnitude)
  double refine_thr = 0.0, deref_thr = 0.0;
  bool have_forced = false;
  std::vector<artemis_host::Leaf> forced_leaves, tree_leaves; // tree_leaves: the leaves this state was built on
  std::vector<int> split_rank, split_local; // owner rank / local index of every leaf (global id) under this state's Z-order split
  std::vector<int> amr_tags();                    // AmrTag of every local block (-1 derefine, 0 same, +1 refine)
  void adopt_state_from(artemis_sim_impl &old);   // copy / prolongate / restrict the conserved state, then re-derive
  // A remesh during the run builds the new state with `adopting` set: the problem generator then runs only on the blocks
  // whose `ic` conditions keep reading its output (adopt_state_from overwrites everything else), and the old state
  // gives up everything but its conserved variables and geometry before the new one allocates (release_for_adoption)
  bool adopting = false;
  void release_for_adoption();
  // ... and takes the tables that are pure functions of a block's position (the `ic` states on the block and on its
  // coarse buffer, the radial factor of the viscosity law) from the old state for every block both meshes hold on this
  // rank -- device-to-device row copies instead of a host libm pass per zone
  const artemis_sim_impl *reuse_from = nullptr;
  // rows [b] <- src rows [ob(b)] of a [nb][nvar][N] field for every block with ob(b) >= 0: leaves are Z-ordered in both
  // meshes, so unchanged blocks come in long runs with a constant offset -- one copy per run, not per block
  template <class OB>
  void copy_rows(Field &dst, const Field &src, const OB &ob) {
    const size_t row = static_cast<size_t>(dst.nvar) * dst.N;
    int b = 0;
    while (b < nb) {
      const int o0 = ob(b);
      if (o0 < 0) {
        ++b;
        continue;
      }
      int len = 1;
      while (b + len < nb && ob(b + len) == o0 + len) ++len;
      CK(artemis_rt_memcpy_d2d(dst.data.p + static_cast<size_t>(b) * row, src.data.p + static_cast<size_t>(o0) * row,
                               sizeof(Real) * row * len, stream), "d2d");
      b += len;
    }
  }
  std::vector<char> ic_coarse_valid; // per block: its coarse-buffer `ic` state has been generated (or copied)
  int reuse_block(int b) const {     // the old state's local block with this block's level and logical location, or -1
    if (!reuse_from) return -1;
    const Block &B = blocks[b];
    auto it = reuse_lookup.find(std::make_tuple(B.level, B.lx[0], B.lx[1], B.lx[2]));
    return it == reuse_lookup.end() ? -1 : it->second;
  }
  std::map<std::tuple<int, int, int, int>, int> reuse_lookup;
  struct DevArr { // raw device array owned by the driver
    void *p = nullptr;
    int n = 0;
    template <class T>
    void upload(const std::vector<T> &h) {
      release();
      n = static_cast<int>(h.size());
      if (h.empty()) return;
      p = artemis_rt_malloc(h.size() * sizeof(T));
      if (!p) throw HipFail("device allocation failed");
      CK(artemis_rt_memcpy_h2d(p, h.data(), h.size() * sizeof(T), nullptr), "h2d");
    }
    void release() {
      if (p) artemis_rt_free(p);
      p = nullptr, n = 0;
    }
    ~DevArr() { release(); }
  };
  struct {
    Field gcoarse, dcoarse;       // coarse buffers, laid out like the prim tables
    DevBuf cgeom, cmetric;
    std::vector<Real> cgeom_h, cmetric_h; // host copies (the problem generator evaluates `ic` states on them)
    DevArr ops_a, ops_u, ops_b;   // ghost ops: packs + direct same/finer | unpacks same/finer | from-coarser
    DevArr ops_fx, ops_fxu;       // flux correction: packs + direct | unpacks
    DevArr fine_boxes, fix_cells; // one-kernel stages: fine-side faces to solve | coarse zones to redo (artemis_hip.h)
    DevArr restrict_blocks, boxes;
    DevArr floor_blocks;          // blocks whose ghost zones took restricted or prolongated values (artemis_hip_ml_floor_ghosts)
    DevBuf gsend, grecv, fsend, frecv;
    std::vector<artemis_msg_t> gmsgs, fmsgs;
    std::vector<int> bc_coarse;
    int max_level = 0;
  } ml;
  struct PeerMsg {
    int peer, tag;
    bool is_send;
    long offset, count;
  };
  struct {
    std::vector<PeerMsg> gmsgs, fmsgs;
    std::vector<artemis_ml_op_t> a, u, b, fx, fxu;
    std::vector<int> restrict_blocks, floor_blocks;
    std::vector<artemis_ml_box_t> boxes;
    std::vector<artemis_ml_face_box_t> fine_boxes;
    std::vector<artemis_ml_fix_cell_t> fix_cells;
    long gsend_n = 0, grecv_n = 0, fsend_n = 0, frecv_n = 0;
  } ml_host;
  artemis_ml_pack_t make_ml_pack() const {
    artemis_ml_pack_t m;
    m.gas_coarse = ml.gcoarse.tab(), m.dust_coarse = ml.dcoarse.tab();
    m.cgeom = ml.cgeom.p, m.cmetric = ml.cmetric.p;
    return m;
  }
  void build_mesh_multilevel();
  void allocate_multilevel();
  void fill_ghosts_multilevel(int prim_idx);
  void flux_correction_multilevel(const artemis_pack_t &p);
  void exchange_messages(std::vector<artemis_msg_t> &msgs);

  // mesh
  int nx[3], mbnx[3], nblk[3], rgrid[3], rcoord[3], lblk[3];
  Real xmin[3], xmax[3];
  int mesh_bc[6];
  int ng, ndim, ni, nj, nk, is, ie, js, je, ks, ke;
  size_t N;
  std::vector<Block> blocks;
  int nb = 0;

  // physics (artemis.cpp:63-72, gas.cpp:55-208, dust.cpp:45-110)
  bool do_gas = true, do_dust = false;
  int ns_gas = 0, ns_dust = 0;
  int recon_gas = ARTEMIS_PLM, riemann_gas = ARTEMIS_HLLC, recon_dust = ARTEMIS_PLM,
      riemann_dust = ARTEMIS_HLLE;
  Real gamma = 1.66666666667, dfloor_gas = 1e-20, siefloor_gas = 1e-20, de_switch = 0.0;
  Real cv_gas = 1.5; // IdealGas specific heat: <gas> cv, or kB / ((gamma-1) amu mu) (gas.cpp:105-116)
  Real dfloor_dust = 1e-20, cfl_gas = 0.8, cfl_dust = 0.8;
  // optional source packages (artemis.cpp:65-72): gravity, rotating_frame, drag
  bool do_gravity = false, do_rframe = false, do_drag = false, do_cooling = false;
  struct { // Orbit of <gravity/binary> (gravity.hpp:30-94) and the pair's centre of mass
    Real a = 1, e = 0, n = 0, coso = 1, sino = 0, cosO = 1, sinO = 0, cosI = 1, sinI = 0, cosf0 = -1, sinf0 = 0;
    Real com[3] = {0, 0, 0};
  } orb;
  // positions of the two bodies at `time` in the frame rotating with omf (binary_mass.cpp:56-70)
  void place_binary() {
    if (!do_gravity || grav.type != ARTEMIS_GRAVITY_BINARY) return;
    const Real omf = do_rframe ? rf_omega : 0.0;
    const Real sint = std::sin(time * (orb.n - omf));
    const Real cost = std::cos(time * (orb.n - omf));
    Real cosf = orb.cosf0 * cost - orb.sinf0 * sint;
    Real sinf = orb.cosf0 * sint + orb.sinf0 * cost;
    const Real rbm = orb.a * (1.0 - SQR(orb.e)) / (1.0 + orb.e * cosf);
    const Real xb = rbm * cosf;
    const Real yb = rbm * sinf;
    cosf = xb * orb.coso - orb.sino * yb;
    sinf = xb * orb.sino + orb.coso * yb;
    const Real rb[3] = {(orb.cosO * cosf - orb.sinO * sinf * orb.cosI), (orb.sinO * cosf + orb.cosO * sinf * orb.cosI),
                        sinf * orb.sinI};
    const Real mu1 = 1. / (1.0 + grav.q), mu2 = grav.q / (1.0 + grav.q);
    for (int n = 0; n < 3; n++) {
      grav.pos[n] = orb.com[n] - mu2 * rb[n];
      grav.pos2[n] = orb.com[n] + mu1 * rb[n];
    }
  }
  artemis_cooling_t cool = {};
  Field cool_tref, cool_beta; // host-filled Tref / beta of every cell (cooling.hpp:47-58, beta_cooling.cpp:98-99)
  artemis_gravity_t grav;
  // <gravity/nbody> + the nbody package (nbody.cpp:48-130) with <nbody> integrator = none: the particles stay
  // where the deck puts them; force rows accumulate like the reference's particle_force (nbody_gravity.hpp:210)
  bool grav_nbody = false, nbody_frame_correction = true;
  std::vector<artemis_nbody_particle_t> particles;
  std::vector<double> particle_force; // [npart][7]
  // the one-kernel stages take the particles from the device and leave the seven sums per particle in device
  // accumulators (artemis_hip_nbody_force_sums): no synchronisation inside the stage loop
  DevBuf nb_dev, nb_force_dev, nb_scratch, amr_maxima, redo_scratch;
  std::vector<artemis_nbody_particle_t> nb_uploaded;
  bool nbody_in_stage = false; // N-body gravity can run inside artemis_hip_stage_general's kernels
  // refined meshes: the cost model of the Z-order split (see build_mesh_multilevel) and what it achieved
  std::vector<double> lb_level_cost;
  double lb_flux_face_cost = 0.0, lb_max_over_mean = 1.0;
  void nbody_stage_args(const artemis_pack_t &p, artemis_stage_general_args_t &a, Real bdt);
  void ensure_redo_scratch() { // the tuned kernel's detect-and-redo lists of THIS state (artemis_stage_args_t.redo_scratch)
    if (redo_scratch.p) return;
    const artemis_pack_t p = make_pack(base);
    redo_scratch.alloc((artemis_hip_redo_scratch_bytes(&p) + sizeof(double) - 1) / sizeof(double));
  }
  void flush_nbody_force();
  Real rf_omega = 0.0, rf_qshear = 0.0;
  artemis_drag_t drag;
  bool damp_to_visc = false; // <gas/damping> damp_to_visc: drag.damp_visc = &diff.visc at the call sites
  artemis_bc_params_t bcpar = {};
  // gas diffusion (gas.cpp:180-197): viscosity and / or heat conduction
  bool do_viscosity = false, do_conduction = false;
  artemis_diffusion_t diff;
  Field gdflux[3];
  Field gdsum;        // artemis_hip_viscous_source's five sums per zone (the one-kernel stages on uniform meshes)
  Field visc_radial;  // per-cell radial factor of the viscosity law (host libm), diff.visc.radial
  DevBuf diff_dist;   // Coords::Distance table of the diffusion flux tasks (static geometry), diff.dist
  Field ic_gas, ic_dust; // disk `ic` condition: the initial primitives as generated (disk.hpp:597-632)
  Field ic_gas_c, ic_dust_c; // ... evaluated on the coarse buffers of a refined mesh (their own zone centres)
  bool edge_ghosts = false; // sequential x1, x2, x3 exchange with extended slabs (viscosity)
  std::string integrator = "rk2";
  int nstages = 2;
  Real gam0[3], gam1[3], beta[3];
  int pgen = PG_BLAST;

  // state
  Field gprim[3];               // primitive ping-pong buffers (fused path); [0..2]
  int base = 0;                 // index of the buffer holding the current state
  Field gu0, gu1, gflux[3], gpflux[3], gvface[3];
  Field dprim[3], du0, du1, dflux[3]; // dust primitives ping-pong with the same index as gprim
  DevBuf geom, dt_dev, metric;
  DevBuf plmtab; // PLM_G geometry table (artemis_hip_plm_table_fill), curvilinear meshes
  std::vector<Real> hgeom, hmetric; // host copies of the edge and x2-trig tables
  int coords = ARTEMIS_CARTESIAN;   // geometry::CoordSelect(artemis/coordinates, ndim)
  // geometry::Coords<GEOM> of cell (k,j,i) of local block b (host side: pgens, history)
  artemis::DCoords cell_coords(int b, int k, int j, int i) const {
    const Real *m = hmetric.empty() ? nullptr : hmetric.data() + b * artemis::metric_block_stride(nj, nk);
    return artemis::coords_of(coords, hgeom.data() + 6 * b, m, nj, nk, k, j, i);
  }
  DevBuf tstate; // device-resident {time, dt, dt_est, beta_dt[3]} for the synchronisation-free loop
  double *dt_host = nullptr;    // pinned
  bool unfused_ready = false;
  // step_fused with outflow faces: the tuned kernel stages the edge zone instead of reading the ghost zones behind them
  // (artemis_stage_args_t.outflow_faces); the per-stage boundary fill leaves the x1 ghost columns alone
  // (artemis_bc_params_t.x1_interior_done) or is skipped altogether when every physical face of the rank is such a face,
  // and evolve() fills the ghost zones once before it returns
  int x1_done_hint = 0;
  std::vector<unsigned char> outflow_by_block; // artemis_stage_args_t.outflow_faces_by_block of the current stage loop
  bool skip_bc_hint = false;
  bool ghosts_stale = false;
  void fill_stale_ghosts();
  bool use_fused = false, fused_possible = false;
  // refined meshes: the one-kernel stages on every block + the fix-up of the zones on coarse-fine faces
  // (step_ml_fused; include/artemis_hip.h "flux correction as a thin fix-up"); ml_tuned: the tuned gas kernel
  bool ml_fused = false, ml_fused_possible = false, ml_tuned = false;
  bool tuned = false; // the hand-tuned gas kernel covers this deck; otherwise the general cell-centred stage
  int general_variant = -1; // what artemis_hip_stage_general ran last (artemis_hip_stage_general_variant)
  int overlap = 0; // 0 off, 1 shell launch + bulk launch, 2 one launch with in-kernel shell signalling
  DevBuf signal; // [0] shell-done counter, [1] wait-kernel timeout flag (as 32-bit words)
  // artemis_stage_args_t.tiny_in / tiny_out / tiny_clear: one 32-bit word per stage of a step through which a tuned stage
  // tells the next one whether ANY velocity below 2^-200 exists in the state it wrote; without one the next stage
  // skips the per-zone detection.  Valid while this driver is the only producer of the primitives: one rank (halo
  // slabs from other ranks are not scanned), copy-type physical conditions, consecutive tuned stages.
  DevBuf tiny_words;
  bool tiny_valid = false;
  bool shell_wait_used = false; // an overlap-2 stage ran since the flag was last cleared
  // "drop-in" accounting mode of the tuned path (bench.py): the last stage also writes the conserved state and
  // every stage ends with the whole-block PrimToCons a Parthenon host runs as FillDerived (artemis_driver.cpp:261)
  int dropin = 0; // 0 off; 1 cons on the last stage + whole-block PrimToCons per stage; 2 cons every stage + ghost-zone PrimToCons
  // test hook (ARTEMIS_LOOPBACK_COMM=1): route same-rank ghost slabs through the communicator
  // as messages to self, so one GPU exercises the RCCL send/recv path end to end
  bool loopback = false;
  bool remote(const Link &L) const { return L.nbr_rank != rank || loopback; }
  bool cons_valid = false;

  std::vector<std::unique_ptr<Link>> links;
  std::vector<int> bc_flat;

  void *stream = nullptr, *comm_stream = nullptr;
  void *ev0 = nullptr, *ev1 = nullptr;
  double kernel_ms_sum = 0.0;
  long kernel_launches = 0;
  bool time_kernels = false;
  std::vector<std::pair<void *, void *>> kev; // event pairs of the timed region

  Real time = 0.0, dt = DBL_MAX, tlim = -1.0;
  long ncycle = 0, nlim = -1;
  double last_wall = 0.0;

  // linear_wave / advection parameters (linear_wave.hpp:44-53, advection.hpp:43-51)
  struct {
    int wave_flag = 0;
    Real amp = 0, vflow = 0, lambda = 0, d0 = 1, p0 = 0, v1_0 = 0, k_par = 0;
    Real cos_a2 = 1, cos_a3 = 1, sin_a2 = 0, sin_a3 = 0, rem[5][5], ev[5], gamma = 0, gm1 = 0;
  } lw;

  // ---------------------------------------------------------------------------------------
  artemis_pack_t make_pack(int prim_idx) const {
    artemis_pack_t p;
    std::memset(&p, 0, sizeof p);
    p.nblocks = nb, p.nghost = ng, p.nx1 = mbnx[0], p.nx2 = mbnx[1], p.nx3 = mbnx[2];
    p.coords = coords;
    p.gm1 = gamma - 1.0;
    p.geom = geom.p;
    p.metric = metric.p;
    p.omega_frame = do_rframe ? rf_omega : 0.0; // fluid_fluxes.hpp:433-437
    p.plm_table = plmtab.p;
    p.gas.nspecies = ns_gas, p.gas.recon = recon_gas, p.gas.riemann = riemann_gas;
    p.gas.dfloor = dfloor_gas, p.gas.siefloor = siefloor_gas, p.gas.de_switch = de_switch;
    p.gas.prim = gprim[prim_idx].tab(), p.gas.cons0 = gu0.tab(), p.gas.cons1 = gu1.tab();
    p.dust.nspecies = ns_dust, p.dust.recon = recon_dust, p.dust.riemann = riemann_dust;
    p.dust.dfloor = dfloor_dust;
    p.dust.prim = dprim[prim_idx].tab(), p.dust.cons0 = du0.tab(), p.dust.cons1 = du1.tab();
    for (int d = 0; d < 3; ++d) {
      p.gas.flux[d] = gflux[d].tab(), p.gas.pflux[d] = gpflux[d].tab();
      p.gas.vface[d] = gvface[d].tab(), p.dust.flux[d] = dflux[d].tab();
      p.gas.diff_flux[d] = gdflux[d].tab();
    }
    return p;
  }

  void setup(const char *deck, int nover, const char *const *over, const artemis_comm_t *c);
  void build_mesh();
  void allocate();
  void ensure_unfused();
  void ensure_ml_flux_arrays();
  void ensure_flux_arrays(bool sparse_ok = false);
  bool flux_ready = false, flux_sparse = false;
  void problem_generator();
  void fill_ghosts(int prim_idx);
  void step_general(bool want_dt, bool device_dt);
  void fill_ghosts_start(int prim_idx, void *hs, int dim = -1);
  void fill_ghosts_finish(int prim_idx, void *hs, int dim = -1, bool apply_bcs = true);
  void materialise_cons();
  Real new_dt_unfused();
  void step_fused(bool want_dt, bool device_dt);
  void step_ml_fused();
  void step_unfused();
  long evolve(long max_cycles);
  void upload_block(Field &f, int b, const std::vector<Real> &h);
  std::vector<Real> download(const Field &f, int b);
  void lw_setup(bool eigen);
  int history(double *out);
  int errors(double *out);
};

namespace {

int parse_bc(const std::string &s, int pgen, int dir) {
  if (s == "periodic") return ARTEMIS_BC_PERIODIC;
  if (s == "outflow") return ARTEMIS_BC_OUTFLOW;
  if (s == "reflecting" || s == "reflect") return ARTEMIS_BC_REFLECT;
  // user conditions are registered per problem (problem_modifier.hpp:114-128)
  if (pgen == PG_STRAT && s == "extrap" && dir != 1) return ARTEMIS_BC_STRAT_EXTRAP;
  if (pgen == PG_STRAT && s == "inflow" && dir == 1) return ARTEMIS_BC_STRAT_INFLOW;
  if (pgen == PG_COND && s == "conductive") return ARTEMIS_BC_CONDUCTIVE; // problem_modifier.hpp:95-108
  if (pgen == PG_DISK && s == "ic") return ARTEMIS_BC_IC;                 // problem_modifier.hpp:67-96
  if (pgen == PG_DISK && s == "extrap") return ARTEMIS_BC_DISK_EXTRAP;
  if (pgen == PG_DISK && s == "viscous") { // problem_modifier.hpp:98-104, disk.hpp:452-455
    if (dir != 0)
      throw std::runtime_error("Viscous boundary conditions only work for the inner or outer radial boundary");
    return ARTEMIS_BC_DISK_VISC;
  }
  throw std::runtime_error("boundary flag '" + s + "' is not built for this problem "
                           "(periodic|outflow|reflecting; strat: extrap on x1/x3, inflow on x2)");
}

// Factor nranks into a rank grid that divides the block grid, preferring to cut the slowest
// dimension first (keeps x1 rows long).
void choose_rank_grid(int nranks, const int nblk[3], int rg[3]) {
  rg[0] = rg[1] = rg[2] = 1;
  int rem = nranks;
  for (int p = 2; rem > 1;) {
    if (rem % p != 0) {
      ++p;
      continue;
    }
    int best = -1;
    for (int d = 2; d >= 0; --d) {
      if ((nblk[d] / rg[d]) % p == 0 && (best < 0 || nblk[d] / rg[d] > nblk[best] / rg[best])) best = d;
    }
    if (best < 0)
      throw std::runtime_error("cannot split the mesh-block grid over " + std::to_string(nranks) +
                               " ranks");
    rg[best] *= p;
    rem /= p;
  }
}

} // namespace

// ---------------------------------------------------------------------------------------
void artemis_sim_impl::setup(const char *deck, int nover, const char *const *over,
                        const artemis_comm_t *c) {
  pin.LoadFromString(deck);
  for (int q = 0; q < nover; ++q) pin.ApplyOverride(over[q]);
  if (c) {
    comm = *c, has_comm = true, rank = c->rank, nranks = c->nranks;
    loopback = artemis::opt(artemis::OPT_LOOPBACK_COMM) == 1;
  }
  // <artemis> (artemis.cpp:48-53,93-97)
  const std::string problem = pin.GetString("artemis", "problem");
  const std::string sys = pin.GetOrAddString("artemis", "coordinates", "cartesian");
  if (problem == "blast") pgen = PG_BLAST;
  else if (problem == "linear_wave") pgen = PG_LINWAVE;
  else if (problem == "advection") pgen = PG_ADVECTION;
  else if (problem == "constant") pgen = PG_CONSTANT;
  else if (problem == "strat") pgen = PG_STRAT;
  else if (problem == "gaussian_bump") pgen = PG_BUMP;
  else if (problem == "conduction") pgen = PG_COND;
  else if (problem == "disk") pgen = PG_DISK;
  else throw std::runtime_error("problem generator '" + problem + "' is not built");
  // <physics> (artemis.cpp:63-72); everything but gas/dust must stay off
  do_gas = pin.GetOrAddBoolean("physics", "gas", true);
  do_dust = pin.GetOrAddBoolean("physics", "dust", false);
  do_gravity = pin.GetOrAddBoolean("physics", "gravity", false);
  do_rframe = pin.GetOrAddBoolean("physics", "rotating_frame", false);
  do_drag = pin.GetOrAddBoolean("physics", "drag", false);
  do_viscosity = pin.GetOrAddBoolean("physics", "viscosity", false);
  do_conduction = pin.GetOrAddBoolean("physics", "conduction", false);
  do_cooling = pin.GetOrAddBoolean("physics", "cooling", false);
  const bool do_nbody = pin.GetOrAddBoolean("physics", "nbody", false);
  if (pin.GetOrAddBoolean("physics", "radiation", false))
    throw std::runtime_error("physics/radiation is out of scope of this build");
  // <parthenon/mesh>
  ng = pin.GetOrAddInteger("parthenon/mesh", "nghost", 2);
  const char *xn[3] = {"x1", "x2", "x3"};
  for (int d = 0; d < 3; ++d) {
    nx[d] = pin.GetOrAddInteger("parthenon/mesh", std::string("n") + xn[d], 1);
    xmin[d] = pin.GetOrAddReal("parthenon/mesh", std::string(xn[d]) + "min", -0.5);
    xmax[d] = pin.GetOrAddReal("parthenon/mesh", std::string(xn[d]) + "max", 0.5);
    mesh_bc[2 * d] = parse_bc(pin.GetOrAddString("parthenon/mesh", std::string("i") + xn[d] + "_bc", "outflow"), pgen, d);
    mesh_bc[2 * d + 1] = parse_bc(pin.GetOrAddString("parthenon/mesh", std::string("o") + xn[d] + "_bc", "outflow"), pgen, d);
    mbnx[d] = pin.GetOrAddInteger("parthenon/meshblock", std::string("n") + xn[d], nx[d]);
    if (mbnx[d] < 1 || nx[d] % mbnx[d] != 0)
      throw std::runtime_error("mesh size must be a multiple of the meshblock size");
    nblk[d] = nx[d] / mbnx[d];
  }
  ndim = (nx[2] > 1) ? 3 : ((nx[1] > 1) ? 2 : 1);
  // <parthenon/mesh> refinement = none | static (| adaptive: the remeshing framework is not built);
  // <parthenon/static_refinementN> level, x?min, x?max (inputs/disk/disk_cart.in:42,68-75)
  {
    const std::string ref = pin.GetOrAddString("parthenon/mesh", "refinement", "none");
    if (ref != "none" && ref != "static" && ref != "adaptive") throw std::runtime_error("parthenon/mesh/refinement must be none|static|adaptive");
    if (ref == "adaptive") {
      adaptive = true;
      amr_max_level = std::max(0, pin.GetOrAddInteger("parthenon/mesh", "numlevel", 1) - 1);
      derefine_count = pin.GetOrAddInteger("parthenon/mesh", "derefine_count", 10);
    }
    if (ref == "static" || ref == "adaptive") {
      // the rank split's cost model (this driver's own block, not a reference parameter): see build_mesh_multilevel
      if (pin.DoesParameterExist("artemis_amd/loadbalance", "level_cost")) lb_level_cost = pin.GetVector("artemis_amd/loadbalance", "level_cost");
      lb_flux_face_cost = pin.GetOrAddReal("artemis_amd/loadbalance", "flux_face_cost", 0.0);
      for (int q = 0; q < 64; ++q) {
        const std::string blk = "parthenon/static_refinement" + std::to_string(q);
        if (!pin.DoesBlockExist(blk)) continue;
        Region r;
        r.level = pin.GetInteger(blk, "level");
        if (r.level < 1) throw std::runtime_error(blk + "/level must be >= 1");
        for (int d = 0; d < 3; ++d) {
          r.lo[d] = pin.GetOrAddReal(blk, std::string(xn[d]) + "min", xmin[d]);
          r.hi[d] = pin.GetOrAddReal(blk, std::string(xn[d]) + "max", xmax[d]);
          if (d < ndim && !(r.lo[d] >= xmin[d] && r.hi[d] <= xmax[d] && r.lo[d] <= r.hi[d]))
            throw std::runtime_error("Refinement region must be smaller than the whole mesh.");
        }
        regions.push_back(r);
      }
      multilevel = !regions.empty() || adaptive;
    }
    if (multilevel) {
      if (ng % 2 != 0) throw std::runtime_error("multilevel meshes need an even number of ghost zones");
      for (int d = 0; d < ndim; ++d)
        if (mbnx[d] % 2 != 0 || mbnx[d] / 2 < (ng + 1) / 2 + 1)
          throw std::runtime_error("multilevel meshes need an even meshblock size of at least nghost + 4 zones");
      // (the conductive condition of the conduction problem, conduction.hpp:125-255, runs on the coarse buffers like on
      //  the fine arrays: it is local to the boundary zone and takes its geometry from the pack it is called on)
    }
  }
  // geometry::CoordSelect (geometry.hpp:38-56, artemis.cpp:94-97)
  if (sys == "cartesian") coords = ARTEMIS_CARTESIAN;
  else if (sys == "spherical")
    coords = (ndim == 1) ? ARTEMIS_SPHERICAL1D : ((ndim == 2) ? ARTEMIS_SPHERICAL2D : ARTEMIS_SPHERICAL3D);
  else if (sys == "cylindrical") coords = ARTEMIS_CYLINDRICAL;
  else if (sys == "axisymmetric") coords = ARTEMIS_AXISYMMETRIC;
  else throw std::runtime_error("Coordinate type not recognized!");
  if (coords != ARTEMIS_CARTESIAN && pgen != PG_BLAST && pgen != PG_COND && pgen != PG_DISK)
    throw std::runtime_error("problem generator '" + problem + "' is Cartesian-only");
  // <gravity> (gravity.cpp:25-118); G = 1 in scale-free units (units.cpp:68-76)
  if (do_gravity) {
    std::memset(&grav, 0, sizeof grav);
    grav.tstart = pin.GetOrAddReal("gravity", "tstart", -DBL_MAX);
    grav.tstop = pin.GetOrAddReal("gravity", "tstop", DBL_MAX);
    int count = 0;
    if (pin.DoesBlockExist("gravity/uniform")) {
      count++, grav.type = ARTEMIS_GRAVITY_UNIFORM;
      grav.g[0] = pin.GetReal("gravity/uniform", "gx1"), grav.g[1] = pin.GetReal("gravity/uniform", "gx2");
      grav.g[2] = pin.GetReal("gravity/uniform", "gx3");
    }
    if (pin.DoesBlockExist("gravity/point")) {
      count++, grav.type = ARTEMIS_GRAVITY_POINT;
      grav.gm = 1.0 * pin.GetReal("gravity/point", "mass");
      grav.soft = pin.GetOrAddReal("gravity/point", "soft", 0.0);
      grav.sink = pin.GetOrAddReal("gravity/point", "sink", 0.0);
      grav.sink_rate = pin.GetOrAddReal("gravity/point", "sink_rate", 0.0);
      grav.pos[0] = pin.GetOrAddReal("gravity/point", "x", 0.0);
      grav.pos[1] = pin.GetOrAddReal("gravity/point", "y", 0.0);
      grav.pos[2] = pin.GetOrAddReal("gravity/point", "z", 0.0);
    }
    if (pin.DoesBlockExist("gravity/binary")) { // gravity.cpp:77-111
      count++, grav.type = ARTEMIS_GRAVITY_BINARY;
      const char *bn = "gravity/binary";
      if (coords == ARTEMIS_AXISYMMETRIC || coords == ARTEMIS_SPHERICAL1D || coords == ARTEMIS_SPHERICAL2D)
        throw std::runtime_error("Binary gravity is not compatable with axisymmetric coordinates!");
      grav.gm = 1.0 * pin.GetReal(bn, "mass");
      grav.soft = pin.GetOrAddReal(bn, "soft1", 0.0), grav.soft2 = pin.GetOrAddReal(bn, "soft2", 0.0);
      grav.sink = pin.GetOrAddReal(bn, "sink1", 0.0), grav.sink2 = pin.GetOrAddReal(bn, "sink2", 0.0);
      grav.sink_rate = pin.GetOrAddReal(bn, "sink_rate1", 0.0), grav.sink_rate2 = pin.GetOrAddReal(bn, "sink_rate2", 0.0);
      orb.com[0] = pin.GetOrAddReal(bn, "x", 0.0), orb.com[1] = pin.GetOrAddReal(bn, "y", 0.0);
      orb.com[2] = pin.GetOrAddReal(bn, "z", 0.0);
      grav.q = pin.GetReal(bn, "q");
      orb.a = pin.GetReal(bn, "a"), orb.e = pin.GetOrAddReal(bn, "e", 0.0);
      const Real ibin = pin.GetOrAddReal(bn, "i", 0.0) * M_PI / 180.;
      const Real obin = pin.GetOrAddReal(bn, "omega", 0.0) * M_PI / 180.;
      const Real Obin = pin.GetOrAddReal(bn, "Omega", 0.0) * M_PI / 180.;
      const Real fbin = pin.GetOrAddReal(bn, "f", 180.0) * M_PI / 180.;
      orb.n = std::sqrt(grav.gm / (orb.a * orb.a * orb.a)); // Orbit::Orbit, gravity.hpp:48-64
      orb.coso = std::cos(obin), orb.sino = std::sin(obin), orb.cosI = std::cos(ibin), orb.sinI = std::sin(ibin);
      orb.cosO = std::cos(Obin), orb.sinO = std::sin(Obin), orb.cosf0 = std::cos(fbin), orb.sinf0 = std::sin(fbin);
    }
    if (pin.DoesBlockExist("gravity/nbody")) { // gravity.cpp:110-117
      count++, grav.type = 0, grav_nbody = true;
      if (!do_nbody) throw std::runtime_error("You have <gravity/nbody> but not physics/nbody = true!");
    }
    if (count == 0) throw std::runtime_error("Unknown gravity node!");
    if (count != 1) throw std::runtime_error("artemis only supports 1 gravity type at this time");
  }
  // <rotating_frame> (rotating_frame.cpp:24-50)
  if (do_rframe) {
    rf_omega = pin.GetReal("rotating_frame", "omega");
    rf_qshear = pin.GetOrAddReal("rotating_frame", "qshear", 0.0);
    if (rf_omega == 0.0) throw std::runtime_error("rotating_frame/omega cannot be zero!");
    if (coords != ARTEMIS_CARTESIAN && rf_qshear != 0.0) // rotating_frame.cpp:34-38
      throw std::runtime_error("rotating_frame/qshear must be zero for non-Cartesian coordinate systems!");
  }
  // <nbody> (nbody/nbody.cpp:48-130, nbody/nbody_setup.cpp:160-722): particle blocks only, no integration
  if (do_nbody) {
    if (coords == ARTEMIS_AXISYMMETRIC || coords == ARTEMIS_SPHERICAL1D || coords == ARTEMIS_SPHERICAL2D)
      throw std::runtime_error("NBody does not work with axisymmetric coordinates!");
    if (pin.GetOrAddString("nbody", "integrator", "ias15") != "none")
      throw std::runtime_error("nbody/integrator: only `none` (static particles) is built; the REBOUND integration "
                               "is outside this build -- a host that integrates the particles calls "
                               "artemis_hip_nbody_gravity with their current state");
    for (const char *pre : {"nbody/binary", "nbody/triple", "nbody/system", "nbody/planet"})
      if (!pin.BlocksWithPrefix(pre).empty())
        throw std::runtime_error(std::string("<") + pre + "*> blocks are not built (use <nbody/particleN>)");
    struct PP {
      Real m = 0, rs = 0, racc = 0, gamma = 0, beta = 0, x = 0, y = 0, z = 0, vx = 0, vy = 0, vz = 0;
      int couple = 1, spline = 0;
    };
    std::map<int, PP> parts;
    for (const std::string &blk : pin.BlocksWithPrefix("nbody/particle")) {
      const size_t s1 = blk.find('/', 6);
      const std::string idstr = blk.substr(14, (s1 == std::string::npos ? blk.size() : s1) - 14);
      const int id = std::stoi(idstr);
      PP &q = parts[id];
      if (s1 == std::string::npos) {
        q.m = pin.GetReal(blk, "mass");
        q.couple = pin.GetOrAddInteger(blk, "couple", 1);
      } else {
        const std::string sub = blk.substr(s1 + 1);
        if (sub == "soft") {
          const std::string t = pin.GetString(blk, "type");
          if (t == "none") q.rs = 0.0, q.spline = 0;
          else if (t == "plummer") q.rs = pin.GetReal(blk, "radius"), q.spline = 0;
          else if (t == "spline") q.rs = pin.GetReal(blk, "radius"), q.spline = 1;
          else throw std::runtime_error("Unknown particle softening type " + t);
        } else if (sub == "sink") {
          q.racc = pin.GetReal(blk, "radius"), q.gamma = pin.GetReal(blk, "gamma");
          q.beta = pin.GetOrAddReal(blk, "beta", 0.0);
        } else if (sub == "initialize") {
          q.x = pin.GetOrAddReal(blk, "x", 0.0), q.y = pin.GetOrAddReal(blk, "y", 0.0), q.z = pin.GetOrAddReal(blk, "z", 0.0);
          q.vx = pin.GetOrAddReal(blk, "vx", 0.0), q.vy = pin.GetOrAddReal(blk, "vy", 0.0), q.vz = pin.GetOrAddReal(blk, "vz", 0.0);
        }
      }
    }
    if (parts.empty()) throw std::runtime_error("physics/nbody = true but no <nbody/particleN> block");
    // nbody_setup.cpp:690-714: total mass rescaled to nbody/mtot, positions / velocities shifted by the
    // mass-weighted sums (not divided by the total mass: kept as the reference has it)
    Real mtot = 0.0, R[3] = {0, 0, 0}, V[3] = {0, 0, 0};
    for (auto &kv : parts) {
      const PP &q = kv.second;
      mtot += q.m;
      R[0] += q.m * q.x, R[1] += q.m * q.y, R[2] += q.m * q.z;
      V[0] += q.m * q.vx, V[1] += q.m * q.vy, V[2] += q.m * q.vz;
    }
    Real mresc = pin.GetOrAddReal("nbody", "mtot", -DBL_MAX);
    if (mresc == -DBL_MAX) mresc = mtot;
    // frame: nbody.cpp:94-109
    nbody_frame_correction = (pin.GetOrAddString("nbody", "frame", "global") == "global");
    const Real Omf = pin.GetOrAddReal("rotating_frame", "omega", 0.0), qsh = pin.GetOrAddReal("rotating_frame", "qshear", 0.0);
    Real Rf[3] = {0, 0, 0}, Vf[3] = {0, 0, 0};
    if (nbody_frame_correction && Omf != 0.0 && qsh != 0.0) {
      const Real R0 = std::pow(SQR(Omf) / (1.0 * mresc), 1.0 / 3.0);
      Rf[0] = R0, Vf[1] = R0 * Omf;
    }
    for (auto &kv : parts) {
      const PP &q = kv.second;
      artemis_nbody_particle_t a;
      std::memset(&a, 0, sizeof a);
      a.gm = 1.0 * (q.m * mresc / mtot);
      a.pos[0] = q.x - R[0], a.pos[1] = q.y - R[1], a.pos[2] = q.z - R[2];
      a.vel[0] = q.vx - V[0], a.vel[1] = q.vy - V[1], a.vel[2] = q.vz - V[2];
      for (int d = 0; d < 3; ++d) a.xf[d] = Rf[d], a.vf[d] = Vf[d];
      a.rs = q.rs, a.racc = q.racc, a.gamma = q.gamma, a.beta = q.beta, a.spline = q.spline, a.couple = q.couple;
      particles.push_back(a);
    }
    particle_force.assign(7 * particles.size(), 0.0);
    if (grav_nbody) grav.gm = 1.0 * mresc; // gravity.cpp:117: gm of the nbody package
  }
  if (grav_nbody && !do_nbody) throw std::runtime_error("You have <gravity/nbody> but not physics/nbody = true!");
  // <cooling> (gas/cooling/cooling.cpp:34-88); the tables are filled once the mesh exists
  if (do_cooling) {
    if (pin.GetString("cooling", "type") != "beta") throw std::runtime_error("Unknown cooling type");
    cool.beta0 = pin.GetReal("cooling", "beta0");
    cool.beta_min = pin.GetOrAddReal("cooling", "beta_min", 1e-12);
    cool.exp_scale = pin.GetOrAddReal("cooling", "exp_scale", 0.0);
    cool.tfloor = pin.GetOrAddReal("cooling", "tfloor", 0.0);
    const std::string tref = pin.GetString("cooling", "tref");
    if (tref == "nbody") throw std::runtime_error("cooling/tref = nbody needs the n-body package (out of scope of this build)");
    if (tref != "powerlaw") throw std::runtime_error("Unknown cooling reference temperature");
    cool.tcyl = pin.GetOrAddReal("cooling", "tcyl", 0.0), cool.cyl_plaw = pin.GetOrAddReal("cooling", "cyl_plaw", 0.0);
    cool.tsph = pin.GetOrAddReal("cooling", "tsph", 0.0), cool.sph_plaw = pin.GetOrAddReal("cooling", "sph_plaw", 0.0);
    cool.gm = do_gravity ? grav.gm : std::nan(""); // Null<Real>() when the gravity package is off
  }
  if (pgen == PG_STRAT) { // strat.hpp:55-70 InitStratParams reads the rotating_frame package
    if (!do_rframe) throw std::runtime_error("problem = strat requires physics/rotating_frame");
    bcpar.qshear = rf_qshear, bcpar.omega = rf_omega;
  }
  // <parthenon/time>
  tlim = pin.GetOrAddReal("parthenon/time", "tlim", -1.0);
  nlim = pin.GetOrAddInteger("parthenon/time", "nlim", -1);
  integrator = pin.GetOrAddString("parthenon/time", "integrator", "rk2");
  // parthenon LowStorageIntegrator (upstream, recalled)
  if (integrator == "rk1") {
    nstages = 1, gam0[0] = 0.0, gam1[0] = 1.0, beta[0] = 1.0;
  } else if (integrator == "rk2") {
    nstages = 2, gam0[0] = 0.0, gam1[0] = 1.0, beta[0] = 1.0;
    gam0[1] = 0.5, gam1[1] = 0.5, beta[1] = 0.5;
  } else if (integrator == "vl2") {
    nstages = 2, gam0[0] = 0.0, gam1[0] = 1.0, beta[0] = 0.5;
    gam0[1] = 0.0, gam1[1] = 1.0, beta[1] = 1.0;
  } else if (integrator == "rk3") {
    nstages = 3, gam0[0] = 0.0, gam1[0] = 1.0, beta[0] = 1.0;
    gam0[1] = 0.25, gam1[1] = 0.75, beta[1] = 0.25;
    gam0[2] = 2.0 / 3.0, gam1[2] = 1.0 / 3.0, beta[2] = 2.0 / 3.0;
  } else {
    throw std::runtime_error("integrator '" + integrator + "' not recognized");
  }
  auto recon_of = [&](const std::string &s) {
    if (s == "pcm") return (int)ARTEMIS_PCM;
    if (s == "plm") return (int)ARTEMIS_PLM;
    if (s == "ppm") return (int)ARTEMIS_PPM;
    throw std::runtime_error("Reconstruction method not recognized.");
  };
  // <gas> (gas.cpp:59-208)
  if (do_gas) {
    recon_gas = recon_of(pin.GetOrAddString("gas", "reconstruct", "plm"));
    const std::string r = pin.GetOrAddString("gas", "riemann", "hllc");
    if (r == "hllc") riemann_gas = ARTEMIS_HLLC;
    else if (r == "hlle") riemann_gas = ARTEMIS_HLLE;
    else if (r == "llf") riemann_gas = ARTEMIS_LLF;
    else throw std::runtime_error("Riemann solver (gas) not recognized.");
    cfl_gas = pin.GetOrAddReal("gas", "cfl", 0.8);
    { // gas.cpp:305-380: the refinement criterion of the gas package
      const std::string rf = pin.GetOrAddString("gas", "refine_field", "none");
      if (rf != "none") {
        if (rf != "density" && rf != "pressure") throw std::runtime_error("Only density or pressure based criterion currently supported!");
        refine_field = (rf == "density") ? 1 : 2;
        const std::string rt = pin.GetString("gas", "refine_type");
        if (rt != "gradient" && rt != "magnitude") throw std::runtime_error("Only gradient or magnitude based criterion currently supported!");
        refine_type = (rt == "gradient") ? 1 : 2;
        refine_thr = pin.GetReal("gas", "refine_thr");
        if (refine_type == 2) deref_thr = pin.GetReal("gas", "deref_thr");
      }
    }
    if (pin.GetOrAddString("gas", "eos", "ideal") != "ideal")
      throw std::runtime_error("only the ideal-gas EOS exists in the reference");
    gamma = pin.GetOrAddReal("gas", "gamma", 1.66666666667);
    dfloor_gas = pin.GetOrAddReal("gas", "dfloor", 1.0e-20);
    siefloor_gas = pin.GetOrAddReal("gas", "siefloor", 1.0e-20);
    de_switch = pin.GetOrAddReal("gas", "de_switch", 0.0);
    ns_gas = pin.GetOrAddInteger("gas", "nspecies", 1);
    // gas.cpp:105-116; kB = amu = 1 in scale-free units (units.cpp:68-76)
    if (pin.DoesParameterExist("gas", "cv")) {
      if (pin.DoesParameterExist("gas", "mmw")) throw std::runtime_error("Cannot specify both cv and mmw");
      cv_gas = pin.GetReal("gas", "cv");
      if (!(cv_gas > 0)) throw std::runtime_error("Only positive cv allowed!");
    } else {
      const Real mu = pin.GetOrAddReal("gas", "mu", 1.);
      if (!(mu > 0)) throw std::runtime_error("Only positive mean molecular weight allowed!");
      cv_gas = 1.0 / ((gamma - 1.) * 1.0 * mu);
    }
  }
  // <gas/viscosity>, <gas/conductivity> (gas.cpp:189-197, diffusion_coeff.hpp:84-136)
  std::memset(&diff, 0, sizeof diff);
  if (do_viscosity || do_conduction) {
    if (!do_gas) throw std::runtime_error("Viscosity / conduction requires the gas package");
    diff.cv = cv_gas;
    auto averaging = [&](const std::string &blk) {
      const std::string a = pin.GetOrAddString(blk, "averaging", "arithmetic");
      if (a == "arithmetic") return 0;
      if (a == "harmonic") return 1;
      throw std::runtime_error(a + " in " + blk + " is not supported");
    };
    if (do_viscosity) {
      artemis_diffcoeff_t &c = diff.visc;
      const std::string t = pin.GetString("gas/viscosity", "type");
      c.avg = averaging("gas/viscosity");
      c.r0 = pin.GetOrAddReal("problem", "r0", 1.0), c.rho_ref = 1.0, c.T_ref = 1.0;
      if (t == "constant" || t == "powerlaw") {
        c.type = ARTEMIS_VISCOSITY_PLAW;
        c.coeff = pin.GetReal("gas/viscosity", "nu");
        c.eta = pin.GetOrAddReal("gas/viscosity", "eta_bulk", 0.0);
        c.r_exp = pin.GetOrAddReal("gas/viscosity", "r_exp", 0.0);
      } else if (t == "alpha") { // diffusion_coeff.hpp:113-119
        if (!do_gravity || (grav.type == ARTEMIS_GRAVITY_UNIFORM && !grav_nbody))
          throw std::runtime_error("gas/viscosity/type = alpha reads gm of the gravity package: gravity/point or gravity/binary is required");
        c.type = ARTEMIS_VISCOSITY_ALPHA;
        c.coeff = pin.GetReal("gas/viscosity", "alpha");
        c.eta = pin.GetOrAddReal("gas/viscosity", "eta_bulk", 0.0);
        c.omega0 = std::sqrt(grav.gm / (c.r0 * c.r0 * c.r0));
      } else {
        throw std::runtime_error(t + " in gas/viscosity is not supported");
      }
    }
    if (do_conduction) {
      artemis_diffcoeff_t &c = diff.cond;
      const std::string t = pin.GetString("gas/conductivity", "type");
      c.avg = averaging("gas/conductivity");
      c.r0 = 1.0;
      if (t == "conductivity") c.type = ARTEMIS_CONDUCTIVITY_PLAW, c.coeff = pin.GetReal("gas/conductivity", "cond");
      else if (t == "diffusivity") c.type = ARTEMIS_THERMALDIFF_PLAW, c.coeff = pin.GetReal("gas/conductivity", "kappa");
      else throw std::runtime_error(t + " in gas/conductivity is not supported");
      c.temp_exp = pin.GetOrAddReal("gas/conductivity", "temp_exp", 0.0);
      c.rho_exp = pin.GetOrAddReal("gas/conductivity", "rho_exp", 0.0);
      c.rho_ref = pin.GetOrAddReal("gas/conductivity", "rho_ref", 1.0);
      c.T_ref = pin.GetOrAddReal("gas/conductivity", "T_ref", 1.0);
    }
    if (ng < 2) throw std::runtime_error("gas diffusion needs nghost >= 2");
    edge_ghosts = do_viscosity; // the strain tensor reads edge / corner ghost zones
  }
  if (pgen == PG_COND) { // conduction.hpp:39-54 InitCondParams + what CondBoundaryImpl reads
    if (!do_conduction) throw std::runtime_error("problem = conduction requires physics/conduction");
    bcpar.cond_temp = pin.GetOrAddReal("problem", "gas_temp", 1.0);
    bcpar.cond_flux = pin.GetOrAddReal("problem", "flux", 0.0);
    for (int d = 0; d < 3; ++d) bcpar.cond_g[d] = (do_gravity && grav.type == ARTEMIS_GRAVITY_UNIFORM) ? grav.g[d] : 0.0;
    bcpar.cond_coeff = diff.cond.coeff, bcpar.cond_cv = diff.cv, bcpar.cond_type = diff.cond.type;
    bcpar.cond_temp_exp = diff.cond.temp_exp, bcpar.cond_rho_exp = diff.cond.rho_exp;
    bcpar.cond_T_ref = diff.cond.T_ref, bcpar.cond_rho_ref = diff.cond.rho_ref;
  }
  // <dust> (dust.cpp:45-110)
  if (do_dust) {
    recon_dust = recon_of(pin.GetOrAddString("dust", "reconstruct", "plm"));
    const std::string r = pin.GetOrAddString("dust", "riemann", "hlle");
    if (r == "hlle") riemann_dust = ARTEMIS_HLLE;
    else if (r == "llf") riemann_dust = ARTEMIS_LLF;
    else throw std::runtime_error("Riemann solver (dust) not recognized.");
    cfl_dust = pin.GetOrAddReal("dust", "cfl", 0.8);
    dfloor_dust = pin.GetOrAddReal("dust", "dfloor", 1.0e-20);
    ns_dust = pin.GetOrAddInteger("dust", "nspecies", 1);
  }
  // <drag> (drag.cpp:25-84, drag.hpp:68-153), <dust> sizes / grain_density (dust.cpp:102-176)
  if (do_drag) {
    std::memset(&drag, 0, sizeof drag);
    const std::string t = pin.GetString("drag", "type");
    if (t == "self") drag.type = ARTEMIS_DRAG_SELF;
    else if (t == "simple_dust") drag.type = ARTEMIS_DRAG_SIMPLE_DUST;
    else throw std::runtime_error("Bad choice of drag type");
    for (int d = 0; d < 3; ++d) drag.xmin[d] = xmin[d], drag.xmax[d] = xmax[d];
    auto damping = [&](const std::string &blk, artemis_damping_t &o, bool present) {
      const char *ax[3] = {"x1", "x2", "x3"};
      for (int d = 0; d < 3; ++d) {
        o.ix[d] = -DBL_MAX, o.ox[d] = DBL_MAX, o.irate[d] = 0.0, o.orate[d] = 0.0;
        if (!present) continue;
        o.ix[d] = pin.GetOrAddReal(blk, std::string("inner_") + ax[d], -DBL_MAX);
        o.irate[d] = pin.GetOrAddReal(blk, std::string("inner_") + ax[d] + "_rate", 0.0);
        o.ox[d] = pin.GetOrAddReal(blk, std::string("outer_") + ax[d], DBL_MAX);
        o.orate[d] = pin.GetOrAddReal(blk, std::string("outer_") + ax[d] + "_rate", 0.0);
      }
      // drag.hpp:101; only the gas block's flag is read by DragSource (drag.cpp:109,135)
      if (present && pin.GetOrAddBoolean(blk, "damp_to_visc", false) && blk == "gas/damping") {
        if (!do_viscosity) throw std::runtime_error("damp_to_visc needs <physics> viscosity = true (the gas package's visc_params)");
        damp_to_visc = true; // viscosity_plaw or viscosity_alpha: the only kinds <gas/viscosity> produces
      }
    };
    const bool gd = do_gas && pin.DoesBlockExist("gas/damping"), dd = do_dust && pin.DoesBlockExist("dust/damping");
    if (drag.type == ARTEMIS_DRAG_SELF && ((do_gas && !gd) || (do_dust && !dd)))
      throw std::runtime_error("With do_drag = true you need a gas/damping (dust/damping) node");
    damping("gas/damping", drag.gas, gd);
    damping("dust/damping", drag.dust, dd);
    if (drag.type == ARTEMIS_DRAG_SIMPLE_DUST) {
      if (!(do_gas && do_dust)) throw std::runtime_error("drag type simple_dust requires do_gas = do_dust = true");
      if (!pin.DoesBlockExist("dust/stopping_time"))
        throw std::runtime_error("drag type simple_dust requires a dust/stopping_time node");
      if (ns_dust > ARTEMIS_MAX_DUST_SPECIES) throw std::runtime_error("too many dust species for simple_dust drag");
      const std::string m = pin.GetString("dust/stopping_time", "type");
      drag.scale = pin.GetOrAddReal("dust/stopping_time", "scale", 1.0);
      if (m == "constant") {
        drag.model = ARTEMIS_DRAG_CONSTANT;
        const std::vector<double> taus = pin.GetVector("dust/stopping_time", "tau");
        if (static_cast<int>(taus.size()) < ns_dust) throw std::runtime_error("dust/stopping_time/tau is too short");
        for (int n = 0; n < ns_dust; ++n) drag.tau[n] = drag.scale * taus[n];
      } else if (m == "stokes") {
        drag.model = ARTEMIS_DRAG_STOKES;
        for (int n = 0; n < ns_dust; ++n) drag.tau[n] = drag.scale;
        if (pin.GetOrAddString("dust", "size_input", "direct") != "direct")
          throw std::runtime_error("dust/size_input: only `direct` is built");
        const std::vector<double> sz = pin.GetVector("dust", "sizes");
        if (static_cast<int>(sz.size()) < ns_dust) throw std::runtime_error("dust/sizes is too short");
        for (int n = 0; n < ns_dust; ++n) drag.sizes[n] = 1.0 * sz[n];
      } else {
        throw std::runtime_error("bad type for stopping time model");
      }
      drag.grain_density = 1.0 * pin.GetOrAddReal("dust", "grain_density", 1.0);
    }
  }
  auto need = [&](int recon) { return recon == ARTEMIS_PCM ? 1 : (recon == ARTEMIS_PLM ? 2 : 3); };
  if ((do_gas && ng < need(recon_gas)) || (do_dust && ng < need(recon_dust)))
    throw std::runtime_error("reconstruction requires more ghost cells (gas.cpp:61-76)");

  const bool setup_timing = artemis::opt(artemis::OPT_SETUP_TIMING) != 0; // host-side phases of the constructor
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto lap = [&](const char *what, std::chrono::steady_clock::time_point t0) {
    if (setup_timing)
      std::fprintf(stderr, "[artemis setup] %-20s %8.3f s\n", what, std::chrono::duration<double>(now() - t0).count());
  };
  auto t_setup = now();
  build_mesh();
  lap("build_mesh", t_setup), t_setup = now();
  allocate();
  lap("allocate", t_setup);
  // one kernel per stage: the hand-tuned gas kernel where it applies, the general cell-centred
  // stage (artemis_hip_stage_general) for everything else the per-task path can do
  tuned = do_gas && !do_dust && ns_gas == 1 && recon_gas != ARTEMIS_PPM && ng >= 2 &&
          coords == ARTEMIS_CARTESIAN && !do_gravity && !do_rframe && !do_drag;
  // the general stage folds DiffusionUpdate (after the diffusion-flux tasks), the curvilinear rotating
  // frame and beta cooling into its kernel; cooling together with drag runs on the per-task chain
  // a refined mesh needs the stage's face fluxes for flux correction (artemis_driver.cpp:196-202): per-task chain
  // N-body gravity rides inside the stage kernels (artemis_stage_general_args_t.nbody_dev) for at most one species per
  // fluid; its seven sums per particle come from artemis_hip_nbody_force_sums, accumulated on the device
  // (artemis_hip_nbody_force_sums keeps one LDS slot set per particle: at most 128; larger sets take the gravity task)
  nbody_in_stage = grav_nbody && ns_gas <= 1 && ns_dust <= 1 && !do_cooling && !artemis::opt(artemis::OPT_NBODY_TASK) &&
                   particles.size() <= 128 &&
                   (coords == ARTEMIS_CARTESIAN || coords == ARTEMIS_CYLINDRICAL || coords == ARTEMIS_SPHERICAL3D);
  fused_possible = !(do_cooling && do_drag) && !multilevel && (!grav_nbody || nbody_in_stage);
  tuned = tuned && fused_possible && !(do_viscosity || do_conduction || do_cooling);
  if (artemis::opt(artemis::OPT_NO_TUNED)) tuned = false; // experiments: route everything through artemis_hip_stage_general
  // Large 2-D gas meshes on one rank: the row-march kernel behind artemis_hip_stage_general (kernels_stage2d.hip, x2
  // march in registers) beats the tile kernel's one-plane form -- 1.00e10 vs 7.3e9 zone-cycles/s at 4096^2, a tie at
  // 1024^2, and the tile kernel wins on small meshes (scripts/tuned2d_timing.py) -- so it takes blocks of >= 2^21 zones.
  if (tuned && ndim == 2 && nranks == 1 && static_cast<long>(mbnx[0]) * mbnx[1] >= (1L << 21) && !artemis::opt(artemis::OPT_TUNED_2D))
    tuned = false;
  // Default path by measurement (scripts/path_timing.py, one MI355X): the cell-centred general stage wins
  // on Cartesian meshes (2048^2 viscous 1.73e9 vs 1.50e9 zone-cycles/s, SURVEY config 3 2.9e9 vs 1.1e9); in
  // curvilinear coordinates every face carries PLM_G / scale-factor geometry and solving each face from
  // both of its cells costs more than the flux arrays save (spherical 3-D blast 1.16e9 vs 1.29e9, disk
  // decks 5.7e8 vs 6.8e8), so those default to the per-task chain.  artemis_sim_set_path overrides.
  // One exception since round 2: a single gas species without dust / drag / cooling runs the streaming tile
  // kernel's curvilinear instantiation (artemis_hip_stage_general variant 2: every face solved once, geometry
  // in registers), which beats the per-task chain (scripts/path_timing.py).
  // Since round 4 a dust species beside it, drag and N-body gravity come along: the gas march leaves the conserved
  // state for the drag finish, the dust runs on its cell-centred kernel (kernels_stage_cell.hip launch_stage_cell).
  const bool curv_tile = do_gas && ns_gas == 1 && ns_dust <= 1 && recon_gas != ARTEMIS_PPM && ng >= 2 &&
                         (!do_dust || !artemis::opt(artemis::OPT_NO_CURV_DUST)) &&
                         !do_cooling && !artemis::opt(artemis::OPT_NO_FUSED_CURV);
  use_fused = fused_possible && (coords == ARTEMIS_CARTESIAN || curv_tile);
  // Refined meshes: the same stage kernels on every block, then the coarse zones on coarse-fine faces redone with the
  // corrected fluxes (step_ml_fused).  Drag couples the fluids after the update: the stage and the fix-up stop at the
  // conserved state (defer_finish) and artemis_hip_stage_finish runs once over every zone after the fix-up.
  ml_fused_possible = multilevel && !(do_cooling && do_drag) && (!grav_nbody || nbody_in_stage) &&
                      (coords == ARTEMIS_CARTESIAN || curv_tile);
  ml_tuned = ml_fused_possible && do_gas && !do_dust && ns_gas == 1 && recon_gas != ARTEMIS_PPM && ng >= 2 &&
             coords == ARTEMIS_CARTESIAN && !do_gravity && !do_rframe && !do_viscosity && !do_conduction && !do_cooling &&
             !artemis::opt(artemis::OPT_NO_TUNED);
  ml_fused = ml_fused_possible && !artemis::opt(artemis::OPT_NO_ML_FUSED);
  if (multilevel) edge_ghosts = false; // the block-graph exchange fills all 3^ndim - 1 directions itself
  t_setup = now();
  // (the per-task chain's start-of-step copies and dense flux arrays: not on the one-kernel paths -- a refined mesh on
  //  step_ml_fused allocates flux rows for the blocks that own a coarse-fine face when its first step asks for them, and
  //  artemis_sim_set_path("unfused") / step_unfused allocate the rest on demand.  On the configs[4] mesh that is a sixth
  //  of the footprint.)
  if (!use_fused && !ml_fused) ensure_unfused();
  else if (ml_fused) ensure_ml_flux_arrays(); // (here, not in the first step: a remesh returns its cache to the device after the build)
  lap("ensure_unfused", t_setup), t_setup = now();
  problem_generator();
  lap("problem_generator", t_setup);
}

void artemis_sim_impl::build_mesh() {
  if (multilevel) {
    build_mesh_multilevel();
    return;
  }
  nblocks_global = static_cast<long>(nblk[0]) * nblk[1] * nblk[2];
  choose_rank_grid(nranks, nblk, rgrid);
  // rank -> coordinates in the rank grid (x1 fastest)
  rcoord[0] = rank % rgrid[0];
  rcoord[1] = (rank / rgrid[0]) % rgrid[1];
  rcoord[2] = rank / (rgrid[0] * rgrid[1]);
  for (int d = 0; d < 3; ++d) lblk[d] = nblk[d] / rgrid[d];
  const int g[3] = {ng, ndim > 1 ? ng : 0, ndim > 2 ? ng : 0};
  ni = mbnx[0] + 2 * g[0], nj = mbnx[1] + 2 * g[1], nk = mbnx[2] + 2 * g[2];
  is = g[0], ie = g[0] + mbnx[0] - 1, js = g[1], je = g[1] + mbnx[1] - 1;
  ks = g[2], ke = g[2] + mbnx[2] - 1;
  N = static_cast<size_t>(ni) * nj * nk;
  nb = lblk[0] * lblk[1] * lblk[2];
  auto rank_of = [&](const int lx[3], int &lb) {
    int rc[3], l[3];
    for (int d = 0; d < 3; ++d) rc[d] = lx[d] / lblk[d], l[d] = lx[d] % lblk[d];
    lb = (l[2] * lblk[1] + l[1]) * lblk[0] + l[0];
    return (rc[2] * rgrid[1] + rc[1]) * rgrid[0] + rc[0];
  };
  blocks.resize(nb);
  for (int b = 0; b < nb; ++b) {
    Block &B = blocks[b];
    int l[3] = {b % lblk[0], (b / lblk[0]) % lblk[1], b / (lblk[0] * lblk[1])};
    for (int d = 0; d < 3; ++d) {
      B.lx[d] = rcoord[d] * lblk[d] + l[d];
      // parthenon default (uniform) mesh generator on the logical location (upstream,
      // recalled): x(r) = xmin*(1-r) + xmax*r with r = lx/nblk
      const Real rl = static_cast<Real>(B.lx[d]) / nblk[d];
      const Real rr = static_cast<Real>(B.lx[d] + 1) / nblk[d];
      B.xmin[d] = (B.lx[d] == 0) ? xmin[d] : xmin[d] * (1.0 - rl) + xmax[d] * rl;
      B.xmax[d] = (B.lx[d] + 1 == nblk[d]) ? xmax[d] : xmin[d] * (1.0 - rr) + xmax[d] * rr;
    }
    B.gid = (static_cast<long>(B.lx[2]) * nblk[1] + B.lx[1]) * nblk[0] + B.lx[0];
    for (int f = 0; f < 6; ++f) {
      const int d = f / 2, side = f % 2;
      B.nbr_rank[f] = -1, B.nbr_block[f] = -1, B.bc[f] = ARTEMIS_BC_NONE;
      if (d >= ndim) {
        B.bc[f] = ARTEMIS_BC_OUTFLOW; // never used: inactive direction
        continue;
      }
      int nl[3] = {B.lx[0], B.lx[1], B.lx[2]};
      nl[d] += side ? 1 : -1;
      const bool outside = nl[d] < 0 || nl[d] >= nblk[d];
      if (outside) {
        if (mesh_bc[f] != ARTEMIS_BC_PERIODIC) {
          B.bc[f] = mesh_bc[f];
          continue;
        }
        if (nblk[d] == 1) { // my own periodic image: plain periodic copy on the device
          B.bc[f] = ARTEMIS_BC_PERIODIC;
          continue;
        }
        nl[d] = (nl[d] + nblk[d]) % nblk[d];
      }
      B.nbr_rank[f] = rank_of(nl, B.nbr_block[f]);
    }
  }
  bc_flat.resize(6 * nb);
  for (int b = 0; b < nb; ++b)
    for (int f = 0; f < 6; ++f) bc_flat[6 * b + f] = blocks[b].bc[f];
}

// Statically refined mesh: leaves of the block tree in Z-order, dealt to the ranks in contiguous runs of
// (nearly) equal length -- every block costs the same, which is Parthenon's default load balance.
void artemis_sim_impl::build_mesh_multilevel() {
  const bool timing = artemis::opt(artemis::OPT_SETUP_TIMING) != 0;
  auto t_last = std::chrono::steady_clock::now();
  auto mlap = [&](const char *what) {
    if (!timing) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[artemis setup]   mesh: %-28s %.3f s\n", what, std::chrono::duration<double>(t - t_last).count());
    t_last = t;
  };
  artemis_host::BlockTree tree;
  tree.ndim = ndim;
  for (int d = 0; d < 3; ++d) tree.nrb[d] = nblk[d], tree.periodic[d] = (d < ndim) && mesh_bc[2 * d] == ARTEMIS_BC_PERIODIC;
  if (have_forced) // an adaptive run's current tree, handed over by the C handle
    for (const artemis_host::Leaf &lf : forced_leaves) tree.ensure(lf.level, lf.lx);
  for (const Region &r : regions) tree.add_region(r.level, r.lo, r.hi, xmin, xmax);
  const std::vector<artemis_host::Leaf> leaves = tree.leaves();
  tree_leaves = leaves;
  ml.max_level = tree.max_level();
  nblocks_global = static_cast<long>(leaves.size());
  if (nblocks_global < nranks) throw std::runtime_error("fewer mesh blocks than ranks");
  // operation lists of the whole mesh (global ids): the split below may weigh blocks by what they take part in
  mlap("tree + leaves");
  const artemis_host::MeshOps M = artemis_host::build_mesh_ops(tree, leaves, mbnx, ng);
  mlap("operation lists");
  std::vector<int> rank_of(leaves.size()), local_of(leaves.size());
  {
    // Cost of a block = zones (the same for every block) x the stage cost of its level (<artemis_amd/loadbalance>
    // level_cost = c0, c1, ...; default 1: every block costs the same, which is Parthenon's default and what measurements
    // of the one-kernel stages give -- the stage kernels do not know levels) + flux_face_cost per coarse-fine face
    // operation the block takes part in (fine-side face solves, restriction, the coarse side's fix-up zones; default 0).
    // Contiguous Z-order runs whose cumulative cost is as even as the block granularity allows.
    std::vector<double> cost(leaves.size(), 1.0);
    bool uniform = true;
    for (size_t gb = 0; gb < leaves.size(); ++gb) {
      const int lv = leaves[gb].level;
      if (lv < static_cast<int>(lb_level_cost.size())) cost[gb] = lb_level_cost[lv];
    }
    if (lb_flux_face_cost != 0.0)
      for (const artemis_host::GlobalOp &o : M.flux) {
        if (o.op.dst_block >= 0) cost[o.op.dst_block] += lb_flux_face_cost;
        if (o.op.src_block >= 0) cost[o.op.src_block] += lb_flux_face_cost;
      }
    for (double c : cost) uniform = uniform && c == cost[0];
    if (uniform) { // equal counts (the first `extra` ranks take one more)
      const long base_n = nblocks_global / nranks, extra = nblocks_global % nranks;
      long g = 0;
      for (int r = 0; r < nranks; ++r) {
        const long cnt = base_n + (r < extra ? 1 : 0);
        for (long q = 0; q < cnt; ++q, ++g) rank_of[g] = r, local_of[g] = static_cast<int>(q);
      }
    } else {
      double total = 0.0;
      for (double c : cost) total += c;
      long g = 0;
      double done = 0.0;
      for (int r = 0; r < nranks; ++r) {
        // rank r's run ends where the cumulative cost is nearest to (r + 1) / nranks of the total, leaving at least one
        // block for every rank still to come
        const double target = total * (r + 1) / nranks;
        long q = 0;
        const long must_leave = nranks - 1 - r;
        while (g < nblocks_global - must_leave &&
               (q == 0 || r == nranks - 1 || std::fabs(done + cost[g] - target) <= std::fabs(done - target))) {
          rank_of[g] = r, local_of[g] = static_cast<int>(q);
          done += cost[g], ++g, ++q;
        }
      }
    }
    std::vector<double> per_rank(nranks, 0.0);
    for (size_t gb = 0; gb < leaves.size(); ++gb) per_rank[rank_of[gb]] += cost[gb];
    double mx = 0.0, sum = 0.0;
    for (double c : per_rank) mx = std::max(mx, c), sum += c;
    lb_max_over_mean = mx / (sum / nranks);
  }
  split_rank = rank_of, split_local = local_of; // (adopt_state_from: a remesh asks both states who owns a leaf)
  const int g[3] = {ng, ndim > 1 ? ng : 0, ndim > 2 ? ng : 0};
  ni = mbnx[0] + 2 * g[0], nj = mbnx[1] + 2 * g[1], nk = mbnx[2] + 2 * g[2];
  is = g[0], ie = g[0] + mbnx[0] - 1, js = g[1], je = g[1] + mbnx[1] - 1;
  ks = g[2], ke = g[2] + mbnx[2] - 1;
  N = static_cast<size_t>(ni) * nj * nk;
  for (int d = 0; d < 3; ++d) rgrid[d] = 1, rcoord[d] = 0, lblk[d] = nblk[d];
  blocks.clear();
  for (size_t gb = 0; gb < leaves.size(); ++gb) {
    if (rank_of[gb] != rank) continue;
    Block B;
    B.level = leaves[gb].level, B.gid = static_cast<long>(gb);
    for (int d = 0; d < 3; ++d) {
      B.lx[d] = leaves[gb].lx[d];
      const int n = tree.extent(B.level, d);
      // parthenon default (uniform) mesh generator on the logical location of the block's level
      const Real rl = static_cast<Real>(B.lx[d]) / n, rr = static_cast<Real>(B.lx[d] + 1) / n;
      B.xmin[d] = (B.lx[d] == 0) ? xmin[d] : xmin[d] * (1.0 - rl) + xmax[d] * rl;
      B.xmax[d] = (B.lx[d] + 1 == n) ? xmax[d] : xmin[d] * (1.0 - rr) + xmax[d] * rr;
    }
    for (int f = 0; f < 6; ++f) {
      const int d = f / 2, side = f % 2;
      B.nbr_rank[f] = -1, B.nbr_block[f] = -1, B.bc[f] = ARTEMIS_BC_NONE;
      if (d >= ndim) {
        B.bc[f] = ARTEMIS_BC_OUTFLOW;
        continue;
      }
      const bool outside = side ? (B.lx[d] + 1 == tree.extent(B.level, d)) : (B.lx[d] == 0);
      if (outside && mesh_bc[f] != ARTEMIS_BC_PERIODIC) B.bc[f] = mesh_bc[f];
    }
    blocks.push_back(B);
  }
  nb = static_cast<int>(blocks.size());
  bc_flat.resize(6 * nb);
  for (int b = 0; b < nb; ++b)
    for (int f = 0; f < 6; ++f) bc_flat[6 * b + f] = blocks[b].bc[f];

  // operation lists: keep what touches this rank, global ids -> local indices / message slots
  const int nfill = 5 * ns_gas + 4 * ns_dust;
  const int nflux = 7 * ns_gas + ((do_viscosity || do_conduction) ? 4 * ns_gas : 0) + 4 * ns_dust;
  auto split = [&](const std::vector<artemis_host::GlobalOp> &all, int nvar, std::vector<artemis_ml_op_t> &packs_direct,
                   std::vector<artemis_ml_op_t> &unpacks, std::vector<artemis_ml_op_t> *late, long &send_total,
                   long &recv_total, std::vector<PeerMsg> &msgs, int tag) {
    std::vector<long> send_n(nranks, 0), recv_n(nranks, 0);
    auto slot = [&](const artemis_ml_op_t &o) { return static_cast<long>(nvar) * o.n[0] * o.n[1] * o.n[2]; };
    // LOOPBACK_COMM on one rank (a transport test: ncclSend / ncclRecv to self): the Z-ordered leaf list is cut into two
    // virtual halves and every operation between them travels as it would between two ranks -- packed into the send
    // buffer, sent to this rank itself, unpacked from the receive buffer (restriction on the fly, coarse-buffer
    // destinations and flux corrections included)
    const bool lb = loopback && nranks == 1;
    auto half = [&](int g) { return (2L * g >= nblocks_global) ? 1 : 0; };
    auto cross = [&](const artemis_host::GlobalOp &G) { return lb && half(G.op.dst_block) != half(G.op.src_block); };
    for (const auto &G : all) {
      const int rd = rank_of[G.op.dst_block], rs = rank_of[G.op.src_block];
      if (rs == rank && (rd != rank || cross(G))) send_n[rd] += slot(G.op);
      if (rd == rank && (rs != rank || cross(G))) recv_n[rs] += slot(G.op);
    }
    std::vector<long> send_at(nranks, 0), recv_at(nranks, 0);
    send_total = recv_total = 0;
    for (int r = 0; r < nranks; ++r) send_at[r] = send_total, send_total += send_n[r], recv_at[r] = recv_total, recv_total += recv_n[r];
    std::vector<long> sa = send_at, ra = recv_at;
    for (const auto &G : all) {
      const int rd = rank_of[G.op.dst_block], rs = rank_of[G.op.src_block];
      if (rd != rank && rs != rank) continue;
      artemis_ml_op_t o = G.op;
      o.dst_block = (rd == rank) ? local_of[G.op.dst_block] : -1;
      o.src_block = (rs == rank) ? local_of[G.op.src_block] : -1;
      const bool coarse_dst = (o.kind == ARTEMIS_ML_FROM_COARSER);
      if (cross(G)) { // (both ends are mine: the pack and the unpack of the same slot)
        artemis_ml_op_t pk = o, un = o;
        pk.dst_block = -1, pk.buf = sa[rd], sa[rd] += slot(o);
        un.src_block = -1, un.buf = ra[rs], ra[rs] += slot(o);
        packs_direct.push_back(pk);
        ((late && coarse_dst) ? *late : unpacks).push_back(un);
      } else if (o.dst_block < 0) {
        o.buf = sa[rd], sa[rd] += slot(o);
        packs_direct.push_back(o);
      } else if (o.src_block < 0) {
        o.buf = ra[rs], ra[rs] += slot(o);
        ((late && coarse_dst) ? *late : unpacks).push_back(o);
      } else {
        ((late && coarse_dst) ? *late : packs_direct).push_back(o);
      }
    }
    msgs.clear(); // one message per peer and direction; buffer addresses are filled in by allocate_multilevel
    for (int r = 0; r < nranks; ++r) {
      if (send_n[r]) msgs.push_back(PeerMsg{r, tag, true, send_at[r], send_n[r]});
      if (recv_n[r]) msgs.push_back(PeerMsg{r, tag, false, recv_at[r], recv_n[r]});
    }
  };
  mlap("split + blocks");
  ml_host.a.clear(), ml_host.u.clear(), ml_host.b.clear(), ml_host.fx.clear(), ml_host.fxu.clear();
  split(M.ghost, nfill, ml_host.a, ml_host.u, &ml_host.b, ml_host.gsend_n, ml_host.grecv_n, ml_host.gmsgs, 7001);
  split(M.flux, nflux, ml_host.fx, ml_host.fxu, nullptr, ml_host.fsend_n, ml_host.frecv_n, ml_host.fmsgs, 7002);
  // The one-kernel stages keep no flux arrays: the fine side of every coarse-fine face is solved on its own
  // (fine_boxes: the faces a FLUX op restricts, artemis_hip_ml_flux_kernel's index map) and the coarse zones that
  // touch a corrected face are redone with it (fix_cells; a zone on a block edge can touch several).
  mlap("rank lists");
  ml_host.fine_boxes.clear(), ml_host.fix_cells.clear();
  {
    // (fine boxes in list order; the coarse zones grouped by block through a per-block mask -- a std::map keyed by zone
    //  cost 12-15 ms per remesh on the configs[4] mesh, a third of the mesh build)
    std::vector<std::vector<const artemis_ml_op_t *>> by_block(nb);
    auto visit = [&](const artemis_ml_op_t &o) {
      const int d = o.dir;
      if (o.src_block >= 0) {
        artemis_ml_face_box_t fb;
        fb.block = o.src_block, fb.dir = d;
        for (int q = 0; q < 3; ++q) {
          if (q == d) fb.lo[q] = o.off[q], fb.n[q] = 1;
          else if (q < ndim) fb.lo[q] = 2 * o.lo[q] + o.off[q], fb.n[q] = 2 * o.n[q];
          else fb.lo[q] = o.lo[q], fb.n[q] = o.n[q];
        }
        ml_host.fine_boxes.push_back(fb);
      }
      if (o.dst_block >= 0) by_block[o.dst_block].push_back(&o);
    };
    for (const auto &o : ml_host.fx) visit(o);
    for (const auto &o : ml_host.fxu) visit(o);
    std::vector<unsigned char> mask(N, 0);
    std::vector<int> hit;
    const int s0[3] = {is, js, ks};
    for (int blk = 0; blk < nb; ++blk) {
      if (by_block[blk].empty()) continue;
      hit.clear();
      for (const artemis_ml_op_t *op : by_block[blk]) {
        const artemis_ml_op_t &o = *op;
        const int d = o.dir;
        const bool lower = (o.lo[d] == s0[d]); // my lower boundary face; otherwise the upper one, stored one zone up
        for (int k = o.lo[2]; k < o.lo[2] + o.n[2]; ++k)
          for (int j = o.lo[1]; j < o.lo[1] + o.n[1]; ++j)
            for (int i = o.lo[0]; i < o.lo[0] + o.n[0]; ++i) {
              int q[3] = {i, j, k};
              if (!lower) q[d] -= 1;
              const int c = (q[2] * nj + q[1]) * ni + q[0];
              if (!mask[c]) hit.push_back(c);
              mask[c] |= static_cast<unsigned char>(1u << (2 * d + (lower ? 0 : 1)));
            }
      }
      std::sort(hit.begin(), hit.end()); // (k, j, i) order within the block, as the keyed map gave it
      for (int c : hit) {
        artemis_ml_fix_cell_t fc;
        fc.block = blk, fc.k = c / (nj * ni), fc.j = (c / ni) % nj, fc.i = c % ni, fc.faces = mask[c];
        ml_host.fix_cells.push_back(fc);
        mask[c] = 0;
      }
    }
  }
  mlap("fix-up zones");
  ml_host.restrict_blocks.clear(), ml_host.boxes.clear();
  for (size_t gb = 0; gb < leaves.size(); ++gb)
    if (rank_of[gb] == rank && M.has_coarser[gb]) ml_host.restrict_blocks.push_back(local_of[gb]);
  for (artemis_ml_box_t bx : M.prolong)
    if (rank_of[bx.block] == rank) {
      bx.block = local_of[bx.block];
      ml_host.boxes.push_back(bx);
    }
  // blocks next to a level boundary: a finer neighbour's restricted zones (direct or unpacked) or a prolongation landed
  // in their ghost zones -- the one-kernel path floors those like the reference's PrimToCons (fill_ghosts_multilevel)
  {
    std::vector<unsigned char> at_level_boundary(nb, 0);
    for (int b : ml_host.restrict_blocks) at_level_boundary[b] = 1;
    for (const auto *list : {&ml_host.a, &ml_host.u})
      for (const artemis_ml_op_t &o : *list)
        if (o.kind == ARTEMIS_ML_FROM_FINER && o.dst_block >= 0) at_level_boundary[o.dst_block] = 1;
    ml_host.floor_blocks.clear();
    for (int b = 0; b < nb; ++b)
      if (at_level_boundary[b]) ml_host.floor_blocks.push_back(b);
  }
  // physical conditions on the coarse buffers: only where a prolongation stencil can reach them
  ml.bc_coarse.assign(6 * nb, ARTEMIS_BC_NONE);
  for (int b : ml_host.restrict_blocks)
    for (int f = 0; f < 6; ++f) ml.bc_coarse[6 * b + f] = blocks[b].bc[f];
  for (int b = 0; b < nb; ++b)
    for (int f = 2 * ndim; f < 6; ++f) ml.bc_coarse[6 * b + f] = ARTEMIS_BC_OUTFLOW; // inactive directions
}

void artemis_sim_impl::allocate_multilevel() {
  const int cg[3] = {ng, ndim > 1 ? ng : 0, ndim > 2 ? ng : 0};
  const int cnx[3] = {mbnx[0] / 2, ndim > 1 ? mbnx[1] / 2 : 1, ndim > 2 ? mbnx[2] / 2 : 1};
  const size_t cN = static_cast<size_t>(cnx[0] + 2 * cg[0]) * (cnx[1] + 2 * cg[1]) * (cnx[2] + 2 * cg[2]);
  ml.gcoarse.alloc(nb, 6 * ns_gas, cN), ml.dcoarse.alloc(nb, 4 * ns_dust, cN);
  std::vector<Real> hc(6 * nb);
  for (int b = 0; b < nb; ++b)
    for (int d = 0; d < 3; ++d) {
      const Real dx = (blocks[b].xmax[d] - blocks[b].xmin[d]) / cnx[d];
      hc[6 * b + 2 * d] = blocks[b].xmin[d] - cg[d] * dx, hc[6 * b + 2 * d + 1] = dx;
    }
  ml.cgeom.alloc(hc.size());
  ml.cgeom_h = hc;
  CK(artemis_rt_memcpy_h2d(ml.cgeom.p, hc.data(), hc.size() * sizeof(Real), nullptr), "h2d cgeom");
  {
    artemis_pack_t p0;
    std::memset(&p0, 0, sizeof p0);
    p0.nblocks = nb, p0.nghost = ng, p0.nx1 = cnx[0], p0.nx2 = cnx[1], p0.nx3 = cnx[2], p0.coords = coords;
    const long nm = artemis_hip_metric_count(&p0);
    if (nm > 0) {
      std::vector<Real> hm(nm, 0.0);
      CK(artemis_hip_metric_fill(&p0, hc.data(), hm.data()), "coarse metric tables");
      ml.cmetric.alloc(nm);
      ml.cmetric_h = hm;
      CK(artemis_rt_memcpy_h2d(ml.cmetric.p, hm.data(), nm * sizeof(Real), nullptr), "h2d cmetric");
    }
  }
  ml.ops_a.upload(ml_host.a), ml.ops_u.upload(ml_host.u), ml.ops_b.upload(ml_host.b);
  ml.ops_fx.upload(ml_host.fx), ml.ops_fxu.upload(ml_host.fxu);
  ml.fine_boxes.upload(ml_host.fine_boxes), ml.fix_cells.upload(ml_host.fix_cells);
  ml.restrict_blocks.upload(ml_host.restrict_blocks), ml.boxes.upload(ml_host.boxes);
  ml.floor_blocks.upload(ml_host.floor_blocks);
  ml.gsend.alloc(ml_host.gsend_n), ml.grecv.alloc(ml_host.grecv_n), ml.fsend.alloc(ml_host.fsend_n), ml.frecv.alloc(ml_host.frecv_n);
  auto address = [](const std::vector<PeerMsg> &in, double *sbase, double *rbase, std::vector<artemis_msg_t> &out) {
    out.clear();
    for (const PeerMsg &q : in) {
      artemis_msg_t m;
      m.peer = q.peer, m.tag = q.tag, m.count = q.count;
      m.send = q.is_send ? sbase + q.offset : nullptr, m.recv = q.is_send ? nullptr : rbase + q.offset;
      out.push_back(m);
    }
  };
  address(ml_host.gmsgs, ml.gsend.p, ml.grecv.p, ml.gmsgs);
  address(ml_host.fmsgs, ml.fsend.p, ml.frecv.p, ml.fmsgs);
  CK(artemis_rt_device_sync(), "sync");
}

// Post the per-peer messages of one phase on the comm stream, ordered behind the compute stream's packs
// and ahead of its unpacks.
void artemis_sim_impl::exchange_messages(std::vector<artemis_msg_t> &msgs) {
  if (msgs.empty()) return;
  if (!has_comm) throw std::runtime_error("remote neighbours but no communicator");
  CK(artemis_rt_event_record(ev0, stream), "event");
  CK(artemis_rt_stream_wait_event(comm_stream, ev0), "wait");
  if (comm.exchange_start(comm.ctx, static_cast<int>(msgs.size()), msgs.data(), comm_stream))
    throw std::runtime_error("exchange_start failed");
  if (comm.exchange_finish(comm.ctx, comm_stream)) throw std::runtime_error("exchange_finish failed");
  CK(artemis_rt_event_record(ev1, comm_stream), "event");
  CK(artemis_rt_stream_wait_event(stream, ev1), "wait");
}

// Ghost fill on a refined mesh (AddBoundaryExchangeTasks with pmesh->multilevel, artemis_driver.cpp:258):
// same-level copies and restricted data from finer neighbours land in the fine ghost zones, coarser
// neighbours' interiors in the coarse buffers; then ghost-halo restriction, physical conditions on the coarse
// buffers, prolongation, physical conditions on the fine arrays.  Seven launches whatever the block count.
void artemis_sim_impl::fill_ghosts_multilevel(int prim_idx) {
  const artemis_pack_t p = make_pack(prim_idx);
  const artemis_ml_pack_t m = make_ml_pack();
  CK(artemis_hip_ml_exchange(&p, &m, static_cast<const artemis_ml_op_t *>(ml.ops_a.p), ml.ops_a.n, ml.gsend.p, ml.grecv.p, stream),
     "ml exchange (pack / same level / from finer)");
  exchange_messages(ml.gmsgs);
  CK(artemis_hip_ml_exchange(&p, &m, static_cast<const artemis_ml_op_t *>(ml.ops_u.p), ml.ops_u.n, ml.gsend.p, ml.grecv.p, stream),
     "ml exchange (unpack)");
  CK(artemis_hip_ml_restrict_halos(&p, &m, static_cast<const int *>(ml.restrict_blocks.p), ml.restrict_blocks.n, stream),
     "ml restrict halos");
  CK(artemis_hip_ml_exchange(&p, &m, static_cast<const artemis_ml_op_t *>(ml.ops_b.p), ml.ops_b.n, ml.gsend.p, ml.grecv.p, stream),
     "ml exchange (from coarser)");
  if (ml.restrict_blocks.n > 0) {
    artemis_pack_t pc = p; // the coarse buffers as a pack of their own: same tables layout, half the zones
    pc.nx1 = mbnx[0] / 2, pc.nx2 = (ndim > 1) ? mbnx[1] / 2 : 1, pc.nx3 = (ndim > 2) ? mbnx[2] / 2 : 1;
    pc.geom = ml.cgeom.p, pc.metric = ml.cmetric.p;
    pc.gas.prim = ml.gcoarse.tab(), pc.dust.prim = ml.dcoarse.tab();
    pc.plm_table = nullptr; // (the fine blocks' table; nothing on the coarse buffers reconstructs)
    artemis_bc_params_t bp = bcpar;
    bp.floor_ghosts = 0;
    if (ic_gas_c.ok() || ic_dust_c.ok()) // `ic`: the profile at the coarse buffers' own zone centres
      bp.ic_gas = ic_gas_c.ok() ? ic_gas_c.tab() : nullptr, bp.ic_dust = ic_dust_c.ok() ? ic_dust_c.tab() : nullptr;
    CK(artemis_hip_apply_bc(&pc, ml.bc_coarse.data(), &bp, stream), "apply_bc (coarse buffers)");
    CK(artemis_hip_ml_prolongate(&p, &m, static_cast<const artemis_ml_box_t *>(ml.boxes.p), ml.boxes.n, stream), "ml prolongate");
  }
  artemis_bc_params_t bp = bcpar;
  bp.floor_ghosts = 0;
  // `ic` faces copy the initial state into ghost zones nobody else writes on a refined mesh (the exchange, the
  // restriction and the prolongation touch zones that face a neighbour only; the stage kernels store active zones): once a
  // buffer holds them they stay -- the fill is skipped for them (on the configs[4] mesh: every physical condition, 19 + 13
  // launches per stage)
  Field &owner = do_gas ? gprim[prim_idx] : dprim[prim_idx];
  const bool skip_ic = ml_fused && owner.ic_filled && !artemis::opt(artemis::OPT_NO_IC_SKIP);
  bool has_ic = false;
  std::vector<int> bc_eff(bc_flat.begin(), bc_flat.end());
  for (int &f : bc_eff) {
    has_ic = has_ic || f == ARTEMIS_BC_IC;
    if (skip_ic && f == ARTEMIS_BC_IC) f = ARTEMIS_BC_NONE;
  }
  if (ml_fused) { // no PrimToCons follows on the one-kernel path: the conditions that compute values floor them here
    bool value_bc = false;
    for (size_t q = 0; q < bc_eff.size(); ++q)
      value_bc = value_bc || bc_eff[q] >= ARTEMIS_BC_CONDUCTIVE || (bc_eff[q] == ARTEMIS_BC_STRAT_EXTRAP && (q % 6) / 2 == 2);
    bp.floor_ghosts = value_bc ? 1 : 0;
  }
  CK(artemis_hip_apply_bc(&p, bc_eff.data(), &bp, stream), "apply_bc");
  if (ml_fused && has_ic) owner.ic_filled = true;
  // the reference's PrimToCons floors every ghost zone behind the fill (fill_derived.cpp:227-262).  Same-level copies of
  // floored zones are floored; restricted averages and prolongations are not always (a rounding below a floor all eight
  // zones sit on, three limited slopes adding up): the blocks that took either, one launch
  if (ml_fused && !artemis::opt(artemis::OPT_NO_ML_FLOOR))
    CK(artemis_hip_ml_floor_ghosts(&p, static_cast<const int *>(ml.floor_blocks.p), ml.floor_blocks.n, stream), "ml floor ghosts");
}

// SendBoundBufs<flxcor_send> / ReceiveFluxCorrections / SetFluxCorrections (artemis_driver.cpp:196-202)
void artemis_sim_impl::flux_correction_multilevel(const artemis_pack_t &p) {
  CK(artemis_hip_ml_flux_correction(&p, static_cast<const artemis_ml_op_t *>(ml.ops_fx.p), ml.ops_fx.n, ml.fsend.p, ml.frecv.p, stream),
     "flux correction");
  exchange_messages(ml.fmsgs);
  CK(artemis_hip_ml_flux_correction(&p, static_cast<const artemis_ml_op_t *>(ml.ops_fxu.p), ml.ops_fxu.n, ml.fsend.p, ml.frecv.p, stream),
     "flux correction (unpack)");
}

void artemis_sim_impl::allocate() {
  // the launcher selects the GPU (artemis_rt_set_device) before creating the simulation
  stream = artemis_rt_stream_create();
  comm_stream = artemis_rt_stream_create();
  ev0 = artemis_rt_event_create(), ev1 = artemis_rt_event_create();
  if (!stream || !comm_stream || !ev0 || !ev1) throw HipFail("stream/event creation failed");
  std::vector<Real> hg(6 * nb);
  for (int b = 0; b < nb; ++b)
    for (int d = 0; d < 3; ++d) {
      // parthenon UniformCartesian (upstream, recalled): dx = (xmax-xmin)/nx,
      // Xf(idx) = (xmin - istart*dx) + idx*dx
      const Real dx = (blocks[b].xmax[d] - blocks[b].xmin[d]) / mbnx[d];
      const int g = (d < ndim) ? ng : 0;
      hg[6 * b + 2 * d] = blocks[b].xmin[d] - g * dx;
      hg[6 * b + 2 * d + 1] = dx;
    }
  geom.alloc(hg.size());
  CK(artemis_rt_memcpy_h2d(geom.p, hg.data(), hg.size() * sizeof(Real), nullptr), "h2d geom");
  hgeom = hg;
  {
    // x2 trigonometry tables of spherical2D/3D, filled with the host libm (artemis_hip.h)
    artemis_pack_t p0;
    std::memset(&p0, 0, sizeof p0);
    p0.nblocks = nb, p0.nghost = ng, p0.nx1 = mbnx[0], p0.nx2 = mbnx[1], p0.nx3 = mbnx[2];
    p0.coords = coords;
    const long nm = artemis_hip_metric_count(&p0);
    if (nm > 0) {
      hmetric.assign(nm, 0.0);
      CK(artemis_hip_metric_fill(&p0, hgeom.data(), hmetric.data()), "metric tables");
      metric.alloc(nm);
      CK(artemis_rt_memcpy_h2d(metric.p, hmetric.data(), nm * sizeof(Real), nullptr), "h2d metric");
    }
  }
  if (coords != ARTEMIS_CARTESIAN && !artemis::opt(artemis::OPT_NO_PLM_TABLE)) {
    // PLM_G's geometric weights, once per mesh (the per-task and cell-centred kernels read them instead of forming
    // ~40 quotients per zone; the tile kernels keep theirs in registers)
    artemis_pack_t p0;
    std::memset(&p0, 0, sizeof p0);
    p0.nblocks = nb, p0.nghost = ng, p0.nx1 = mbnx[0], p0.nx2 = mbnx[1], p0.nx3 = mbnx[2];
    p0.coords = coords, p0.geom = geom.p, p0.metric = metric.p;
    plmtab.alloc(artemis_hip_plm_table_count(&p0));
    CK(artemis_hip_plm_table_fill(&p0, plmtab.p, nullptr), "PLM_G table");
  }
  dt_dev.alloc(1);
  tstate.alloc(6);
  signal.alloc(2);
  tiny_words.alloc(2); // (up to four 32-bit words: nstages <= 3)
  dt_host = static_cast<double *>(artemis_rt_malloc_host(sizeof(double)));
  if (!dt_host) throw HipFail("pinned allocation failed");
  gprim[0].alloc(nb, 6 * ns_gas, N);
  gu0.alloc(nb, 6 * ns_gas, N);
  dprim[0].alloc(nb, 4 * ns_dust, N);
  du0.alloc(nb, 4 * ns_dust, N);
  // ghost-slab links
  for (int b = 0; b < nb; ++b)
    for (int f = 0; f < 2 * ndim; ++f) {
      if (blocks[b].nbr_rank[f] < 0) continue;
      auto L = std::make_unique<Link>();
      L->b = b, L->face = f, L->nbr_rank = blocks[b].nbr_rank[f], L->nbr_block = blocks[b].nbr_block[f];
      const artemis_pack_t p = make_pack(0);
      L->count = artemis_hip_halo_count_ext(&p, f, edge_ghosts ? 1 : 0);
      L->sbuf.alloc(L->count);
      if (remote(*L)) L->rbuf.alloc(L->count);
      // tag = (destination global block id, destination face)
      int nl[3] = {blocks[b].lx[0], blocks[b].lx[1], blocks[b].lx[2]};
      const int d = f / 2;
      nl[d] = (nl[d] + ((f % 2) ? 1 : -1) + nblk[d]) % nblk[d];
      const long ngid = (static_cast<long>(nl[2]) * nblk[1] + nl[1]) * nblk[0] + nl[0];
      L->tag_send = static_cast<int>(ngid * 6 + (f ^ 1));
      L->tag_recv = static_cast<int>(blocks[b].gid * 6 + f);
      links.push_back(std::move(L));
    }
  if (multilevel) allocate_multilevel();
  CK(artemis_rt_device_sync(), "sync");
}

// the flux arrays alone: what the one-kernel stages of a refined mesh need next to their primitive buffers (the
// fine-side faces of coarse-fine boundaries and their restrictions live in them); the per-task chain also needs u1
// sparse_ok: the caller touches flux arrays only on coarse-fine faces (the one-kernel stages of a refined mesh: the fine
// side's faces, their restrictions, the faces of the fix-up zones) -- rows then exist for the blocks that own such a
// face alone (48 arrays per zone otherwise: a third of a refined mesh's footprint).  A later caller that needs every
// block's rows (the per-task chain) gets them allocated afresh.
void artemis_sim_impl::ensure_flux_arrays(bool sparse_ok) {
  if (flux_ready && (!flux_sparse || sparse_ok)) return;
  const bool sparse = sparse_ok && multilevel && !artemis::opt(artemis::OPT_DENSE_FLUX);
  std::vector<char> need;
  if (sparse) {
    need.assign(nb, 0);
    auto mark = [&](int b) {
      if (b >= 0 && b < nb) need[b] = 1;
    };
    for (const auto &o : ml_host.fx) mark(o.dst_block), mark(o.src_block);
    for (const auto &o : ml_host.fxu) mark(o.dst_block), mark(o.src_block);
    for (const auto &fb : ml_host.fine_boxes) mark(fb.block);
    for (const auto &c : ml_host.fix_cells) mark(c.block);
  }
  auto make = [&](Field &f, int nvar) {
    f.release();
    if (sparse) f.alloc_sparse(nb, nvar, N, need);
    else f.alloc(nb, nvar, N);
  };
  for (int d = 0; d < 3; ++d) {
    make(gflux[d], 6 * ns_gas), make(gpflux[d], ns_gas), make(gvface[d], ns_gas), make(dflux[d], 4 * ns_dust);
    if (do_viscosity || do_conduction) make(gdflux[d], 4 * ns_gas);
  }
  flux_ready = true, flux_sparse = sparse;
}
// step_ml_fused's flux arrays: rows for the blocks that own a coarse-fine face, unless the diffusion fluxes are written for
// the whole pack (heat conduction, or a pack the viscous-source march does not cover)
void artemis_sim_impl::ensure_ml_flux_arrays() {
  const artemis_pack_t p0 = make_pack(base);
  const bool diffuse0 = do_gas && (do_viscosity || do_conduction);
  const bool vs0 = diffuse0 && do_viscosity && !do_conduction && ns_gas == 1 && !artemis::opt(artemis::OPT_NO_VISC_SOURCE) &&
                   artemis_hip_viscous_source_covers(&p0) != 0;
  ensure_flux_arrays(!diffuse0 || vs0);
  // ... and everything else a step asks for, NOW: a remesh builds the new state while the old one's released buffers
  // sit in the allocator's cache and returns what is left to the device afterwards -- a buffer first asked for by the
  // first step would be a fresh hipMalloc of GBs after every remesh (and its predecessor a hipFree)
  for (int q = 1; q < 3; ++q) {
    if (!gprim[q].ok()) gprim[q].alloc(nb, 6 * ns_gas, N);
    if (!dprim[q].ok()) dprim[q].alloc(nb, 4 * ns_dust, N);
  }
  if (vs0 && !gdsum.ok()) gdsum.alloc(nb, 5, N);
}
void artemis_sim_impl::ensure_unfused() {
  if (unfused_ready) return;
  gu1.alloc(nb, 6 * ns_gas, N);
  du1.alloc(nb, 4 * ns_dust, N);
  ensure_flux_arrays();
  unfused_ready = true;
}

void artemis_sim_impl::upload_block(Field &f, int b, const std::vector<Real> &h) {
  CK(artemis_rt_memcpy_h2d(f.var(b, 0), h.data(), h.size() * sizeof(Real), stream), "h2d");
  CK(artemis_rt_stream_sync(stream), "sync");
}
std::vector<Real> artemis_sim_impl::download(const Field &f, int b) {
  std::vector<Real> h(static_cast<size_t>(f.nvar) * N);
  CK(artemis_rt_stream_sync(stream), "sync");
  CK(artemis_rt_memcpy_d2h(h.data(), f.var(b, 0), h.size() * sizeof(Real), stream), "d2h");
  CK(artemis_rt_stream_sync(stream), "sync");
  return h;
}

// ---------------------------------------------------------------------------------------
// Ghost fill of the FillGhost primitives of buffer `prim_idx`: neighbour slabs (device copy
// within the rank, comm callbacks between ranks), then physical BCs.  This is where
// AddBoundaryExchangeTasks sits in the reference (artemis_driver.cpp:258).
// `hs` is the stream the pack/unpack kernels run on: the compute stream normally, the comm stream
// when the exchange overlaps the bulk of the stage kernel (then everything between "shell done"
// and "ghosts filled" lives on the comm stream and the compute stream keeps computing).
// dim < 0: every face at once (slabs span the interior of the other dimensions; the hydro stencil
// reads no edge or corner zone).  dim >= 0: only the faces of that dimension, with slabs extended
// over the ghost zones of the lower dimensions (edge_ghosts mode, one phase per dimension).
void artemis_sim_impl::fill_ghosts_start(int prim_idx, void *hs, int dim) {
  if (links.empty()) return;
  const artemis_pack_t p = make_pack(prim_idx);
  const int ext = (dim >= 0) ? 1 : 0;
  std::vector<artemis_msg_t> msgs;
  for (auto &L : links) {
    if (dim >= 0 && L->face / 2 != dim) continue;
    CK(artemis_hip_halo_pack_ext(&p, L->b, L->face, ext, L->sbuf.p, hs), "halo pack");
    if (remote(*L)) {
      artemis_msg_t m;
      m.peer = L->nbr_rank, m.tag = L->tag_send, m.send = L->sbuf.p, m.recv = nullptr, m.count = L->count;
      msgs.push_back(m);
      m.tag = L->tag_recv, m.send = nullptr, m.recv = L->rbuf.p;
      msgs.push_back(m);
    }
  }
  if (!msgs.empty()) {
    if (!has_comm) throw std::runtime_error("remote neighbours but no communicator");
    if (hs != comm_stream) { // sends may start once the packs are done
      CK(artemis_rt_event_record(ev0, hs), "event");
      CK(artemis_rt_stream_wait_event(comm_stream, ev0), "wait");
    }
    if (comm.exchange_start(comm.ctx, static_cast<int>(msgs.size()), msgs.data(), comm_stream))
      throw std::runtime_error("exchange_start failed");
  }
}
void artemis_sim_impl::fill_ghosts_finish(int prim_idx, void *hs, int dim, bool apply_bcs) {
  const artemis_pack_t p = make_pack(prim_idx);
  const int ext = (dim >= 0) ? 1 : 0;
  bool any_remote = false;
  for (auto &L : links) any_remote = any_remote || (remote(*L) && (dim < 0 || L->face / 2 == dim));
  if (any_remote) {
    if (comm.exchange_finish(comm.ctx, comm_stream)) throw std::runtime_error("exchange_finish failed");
    if (hs != comm_stream) {
      CK(artemis_rt_event_record(ev1, comm_stream), "event");
      CK(artemis_rt_stream_wait_event(hs, ev1), "wait");
    }
  }
  for (auto &L : links) {
    if (dim >= 0 && L->face / 2 != dim) continue;
    if (!remote(*L)) {
      CK(artemis_hip_halo_unpack_ext(&p, L->nbr_block, L->face ^ 1, ext, L->sbuf.p, hs), "halo unpack");
    } else {
      // what I received through face f came from the neighbour's opposite face
      CK(artemis_hip_halo_unpack_ext(&p, L->b, L->face, ext, L->rbuf.p, hs), "halo unpack");
    }
  }
  if (hs != stream) { // hand the filled ghosts back to the compute stream
    CK(artemis_rt_event_record(ev1, hs), "event");
    CK(artemis_rt_stream_wait_event(stream, ev1), "wait");
  }
  if (apply_bcs) {
    // user conditions that compute new ghost values (conductive, disk ic / extrap / viscous) rely on the
    // PrimToCons that follows them to apply the floors; the fused stages do not run PrimToCons
    artemis_bc_params_t bp = bcpar;
    bool value_bc = false;
    for (size_t q = 0; q < bc_flat.size(); ++q) // (the strat x3 `extrap` continues the density with pow())
      value_bc = value_bc || bc_flat[q] >= ARTEMIS_BC_CONDUCTIVE ||
                 (bc_flat[q] == ARTEMIS_BC_STRAT_EXTRAP && (q % 6) / 2 == 2);
    bp.floor_ghosts = (use_fused && value_bc) ? 1 : 0;
    bp.x1_interior_done = x1_done_hint;
    if (!skip_bc_hint) CK(artemis_hip_apply_bc(&p, bc_flat.data(), &bp, stream), "apply_bc");
  }
}
void artemis_sim_impl::fill_ghosts(int prim_idx) {
  if (multilevel) {
    fill_ghosts_multilevel(prim_idx);
    return;
  }
  if (!edge_ghosts) {
    fill_ghosts_start(prim_idx, stream, -1);
    fill_ghosts_finish(prim_idx, stream, -1, true);
    return;
  }
  // x1, then x2, then x3: each phase forwards the ghost zones the earlier ones filled, so edge and
  // corner zones arrive from the diagonal neighbours; physical conditions come last, over the
  // entire extent (parthenon order)
  for (int d = 0; d < ndim; ++d) {
    fill_ghosts_start(prim_idx, stream, d);
    fill_ghosts_finish(prim_idx, stream, d, d == ndim - 1);
  }
}
void artemis_sim_impl::materialise_cons() {
  if (cons_valid) return;
  const artemis_pack_t p = make_pack(base);
  CK(artemis_hip_prim_to_cons(&p, stream), "PrimToCons");
  cons_valid = true;
}

// ---------------------------------------------------------------------------------------
// pgen/linear_wave.hpp:117-215 and pgen/advection.hpp:62-168: wavevector set-up
void artemis_sim_impl::lw_setup(bool eigen) {
  const bool multi_d = ndim > 1, three_d = ndim > 2;
  const bool along_x1 = pin.GetOrAddBoolean("problem", "along_x1", false);
  const bool along_x2 = pin.GetOrAddBoolean("problem", "along_x2", false);
  const bool along_x3 = pin.GetOrAddBoolean("problem", "along_x3", false);
  if ((along_x1 && (along_x2 || along_x3)) || (along_x2 && along_x3))
    throw std::runtime_error("Can only specify one of along_x1/2/3 to be true");
  if ((along_x2 || along_x3) && ndim == 1)
    throw std::runtime_error("Cannot specify waves along x2 or x3 axis in 1D");
  if (along_x3 && ndim == 2) throw std::runtime_error("Cannot specify waves along x3 axis in 2D");
  const Real x1size = xmax[0] - xmin[0], x2size = xmax[1] - xmin[1], x3size = xmax[2] - xmin[2];
  lw.cos_a3 = 1.0, lw.sin_a3 = 0.0, lw.cos_a2 = 1.0, lw.sin_a2 = 0.0;
  if (multi_d && !along_x1) {
    const Real ang_3 = std::atan(x1size / x2size);
    lw.sin_a3 = std::sin(ang_3), lw.cos_a3 = std::cos(ang_3);
  }
  if (three_d && !along_x1) {
    const Real ang_2 = std::atan(0.5 * (x1size * lw.cos_a3 + x2size * lw.sin_a3) / x3size);
    lw.sin_a2 = std::sin(ang_2), lw.cos_a2 = std::cos(ang_2);
  }
  if (along_x2) lw.cos_a3 = 0.0, lw.sin_a3 = 1.0, lw.cos_a2 = 1.0, lw.sin_a2 = 0.0;
  if (along_x3) lw.cos_a3 = 0.0, lw.sin_a3 = 1.0, lw.cos_a2 = 0.0, lw.sin_a2 = 1.0;
  lw.lambda = std::numeric_limits<float>::max();
  if (lw.cos_a2 * lw.cos_a3 > 0.0) lw.lambda = std::min(lw.lambda, x1size * lw.cos_a2 * lw.cos_a3);
  if (lw.cos_a2 * lw.sin_a3 > 0.0) lw.lambda = std::min(lw.lambda, x2size * lw.cos_a2 * lw.sin_a3);
  if (lw.sin_a2 > 0.0) lw.lambda = std::min(lw.lambda, x3size * lw.sin_a2);
  lw.k_par = 2.0 * (M_PI) / lw.lambda;
  lw.d0 = 1.0, lw.v1_0 = lw.vflow;
  lw.gamma = gamma, lw.gm1 = gamma - 1.0, lw.p0 = 1.0 / gamma;
  if (eigen) { // linear_wave.hpp:58-111 HydroEigensystem(d0, v1_0, 0, 0, p0, gamma)
    const Real d = lw.d0, v1 = lw.v1_0, v2 = 0.0, v3 = 0.0, p = lw.p0;
    const Real vsq = v1 * v1 + v2 * v2 + v3 * v3;
    const Real h = (p / (gamma - 1.0) + 0.5 * d * vsq + p) / d;
    const Real a = std::sqrt(gamma * p / d);
    lw.ev[0] = v1 - a, lw.ev[1] = v1, lw.ev[2] = v1, lw.ev[3] = v1, lw.ev[4] = v1 + a;
    Real(&r)[5][5] = lw.rem;
    r[0][0] = 1.0, r[1][0] = v1 - a, r[2][0] = v2, r[3][0] = v3, r[4][0] = h - v1 * a;
    r[0][1] = 0.0, r[1][1] = 0.0, r[2][1] = 1.0, r[3][1] = 0.0, r[4][1] = v2;
    r[0][2] = 0.0, r[1][2] = 0.0, r[2][2] = 0.0, r[3][2] = 1.0, r[4][2] = v3;
    r[0][3] = 1.0, r[1][3] = v1, r[2][3] = v2, r[3][3] = v3, r[4][3] = 0.5 * vsq;
    r[0][4] = 1.0, r[1][4] = v1 + a, r[2][4] = v2, r[3][4] = v3, r[4][4] = h + v1 * a;
  }
}

// Coords<GEOM>::ConvertCoordsToCart (geometry.hpp:248, cylindrical.hpp:88-92,
// spherical.hpp:166-173 / :355-362 / :528-534, axisymmetric.hpp:77-82)
static void to_cart(int sys, const Real xi[3], Real xc[3]) {
  if (sys == ARTEMIS_SPHERICAL3D || sys == ARTEMIS_SPHERICAL2D) {
    const Real cp = (sys == ARTEMIS_SPHERICAL3D) ? std::cos(xi[2]) : 1.0;
    const Real sp = (sys == ARTEMIS_SPHERICAL3D) ? std::sin(xi[2]) : 0.0;
    const Real ct = std::cos(xi[1]), st = std::sin(xi[1]);
    xc[0] = xi[0] * st * cp, xc[1] = xi[0] * st * sp, xc[2] = xi[0] * ct;
  } else if (sys == ARTEMIS_SPHERICAL1D) {
    const Real cp = 1.0, sp = 0.0, ct = 0.0, st = 1.0;
    xc[0] = xi[0] * st * cp, xc[1] = xi[0] * st * sp, xc[2] = xi[0] * ct;
  } else if (sys == ARTEMIS_CYLINDRICAL) {
    const Real cp = std::cos(xi[1]), sp = std::sin(xi[1]);
    xc[0] = xi[0] * cp, xc[1] = xi[0] * sp, xc[2] = xi[2];
  } else if (sys == ARTEMIS_AXISYMMETRIC) {
    const Real cp = std::cos(xi[2]), sp = std::sin(xi[2]);
    xc[0] = xi[0] * cp, xc[1] = xi[0] * sp, xc[2] = xi[1];
  } else {
    xc[0] = xi[0], xc[1] = xi[1], xc[2] = xi[2];
  }
}

void artemis_sim_impl::problem_generator() {
  const bool pg_timing = artemis::opt(artemis::OPT_SETUP_TIMING) != 0;
  auto pg_t0 = std::chrono::steady_clock::now();
  auto pg_lap = [&](const char *what) {
    if (pg_timing)
      std::fprintf(stderr, "[artemis setup]   pgen: %-24s %8.3f s\n", what,
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - pg_t0).count());
    pg_t0 = std::chrono::steady_clock::now();
  };
  const Real gm1 = gamma - 1.0;
  auto xf = [&](int b, int d, int idx) { // Coordinates_t::Xf (geometry.hpp:65-72)
    const Real dx = (blocks[b].xmax[d] - blocks[b].xmin[d]) / mbnx[d];
    const int g = (d < ndim) ? ng : 0;
    return (blocks[b].xmin[d] - g * dx) + idx * dx;
  };
  if (pgen == PG_LINWAVE) {
    lw.wave_flag = pin.GetInteger("problem", "wave_flag");
    lw.amp = pin.GetReal("problem", "amp");
    lw.vflow = pin.GetOrAddReal("problem", "vflow", 0.0);
    lw_setup(true);
    const Real nperiod = pin.GetOrAddReal("problem", "nperiod", 1.0);
    tlim = nperiod * (std::abs(lw.lambda / lw.ev[lw.wave_flag])); // linear_wave.hpp:214-215
  } else if (pgen == PG_ADVECTION) {
    lw.amp = pin.GetReal("problem", "amp");
    lw.vflow = pin.GetOrAddReal("problem", "vflow", 0.0);
    if (do_gas && ns_gas != 1) throw std::runtime_error("Advection pgen requires a single gas species.");
    if (do_dust && ns_dust != 2) throw std::runtime_error("Advection pgen requires two dust species.");
    lw_setup(false);
    const Real nperiod = pin.GetOrAddReal("problem", "nperiod", 1.0);
    tlim = nperiod * (std::abs(lw.lambda / lw.v1_0)); // advection.hpp:167-168
  }
  // blast parameters (blast.hpp:137-155)
  const Real rinit = pin.GetOrAddReal("problem", "radius", 1.0);
  const Real ienergy = pin.GetOrAddReal("problem", "internal_energy", 1.0);
  const Real p0 = pin.GetOrAddReal("problem", "p0", 1.0);
  const Real d0 = pin.GetOrAddReal("problem", "d0", 1.0);
  const Real x0[3] = {pin.GetOrAddReal("problem", "x1", 0.0), pin.GetOrAddReal("problem", "x2", 0.0),
                      pin.GetOrAddReal("problem", "x3", 0.0)};
  const int samples = pin.GetOrAddInteger("problem", "samples", -1);
  const std::string sym = pin.GetOrAddString("problem", "symmetry", "spherical");
  if (pgen == PG_BLAST && sym != "spherical" && sym != "cylindrical")
    throw std::runtime_error("Bad blast wave symmetry parameter in <problem>!");
  const int btype = (sym == "spherical") ? 1 : 2;

  // constant.hpp:63-78 / strat.hpp:55-70 parameters; Cv = kB / ((gamma-1) amu mu) with
  // kB = amu = 1 in scale-free units and mu = 1 (gas.cpp:106-116, units.cpp:68-76)
  struct { Real g_rho = 1, g_v[3] = {0, 0, 0}, g_temp = 1, d_rho = 1, d_v[3] = {0, 0, 0}; } cs;
  struct { Real h = 1, rho0 = 1, dens_min = 1e-5, d2g = 0.01; } st;
  const Real cv = cv_gas;
  if (pgen == PG_CONSTANT) {
    if (do_gas && ns_gas != 1) throw std::runtime_error("Constant pgen requires a single gas species.");
    if (pin.GetString("problem", "system") != "cartesian")
      throw std::runtime_error("constant pgen: only problem/system = cartesian is built");
    cs.g_rho = pin.GetOrAddReal("problem", "gas_rho", 1.0);
    cs.g_v[0] = pin.GetOrAddReal("problem", "gas_vx1", 0.0), cs.g_v[1] = pin.GetOrAddReal("problem", "gas_vx2", 0.0);
    cs.g_v[2] = pin.GetOrAddReal("problem", "gas_vx3", 0.0);
    cs.g_temp = pin.GetOrAddReal("problem", "gas_temp", 1.0);
    cs.d_rho = pin.GetOrAddReal("problem", "dust_rho", 1.0);
    cs.d_v[0] = pin.GetOrAddReal("problem", "dust_vx1", 0.0), cs.d_v[1] = pin.GetOrAddReal("problem", "dust_vx2", 0.0);
    cs.d_v[2] = pin.GetOrAddReal("problem", "dust_vx3", 0.0);
  }
  if (pgen == PG_STRAT) {
    if (!do_gas || ns_gas != 1) throw std::runtime_error("strat pgen requires a single gas species.");
    st.h = pin.GetOrAddReal("problem", "h", 1.0);
    st.rho0 = pin.GetOrAddReal("problem", "rho0", 1.0);
    st.dens_min = pin.GetOrAddReal("problem", "dens_min", 1.0e-5);
    st.d2g = pin.GetOrAddReal("problem", "dust_to_gas", 0.01);
  }
  // disk.hpp:50-66 DiskParams, :253-323 InitDiskParams (n-body temperature: out of scope)
  struct {
    Real r0, h0, p, q, flare, rho0, dens_min, pres_min, gm, Omega0, l0, omf, dust_to_gas, rexp, rcav;
    Real Gamma, gamma_gas, alpha, nu0, nu_indx, mdot, temp_soft2;
    bool quiet_start;
  } dk = {};
  if (pgen == PG_DISK) {
    if (!do_gas || ns_gas != 1) throw std::runtime_error("disk pgen requires a single gas species.");
    if (!do_gravity) throw std::runtime_error("disk pgen reads gm of the gravity package: physics/gravity is required");
    dk.gm = grav.gm;
    dk.r0 = pin.GetOrAddReal("problem", "r0", 1.0);
    dk.Omega0 = std::sqrt(dk.gm / (dk.r0 * dk.r0 * dk.r0));
    dk.rho0 = pin.GetOrAddReal("problem", "rho0", 1.0);
    dk.p = pin.GetOrAddReal("problem", "dslope", -2.25);
    dk.h0 = pin.GetOrAddReal("problem", "h0", 0.05);
    dk.gamma_gas = gamma;
    dk.Gamma = pin.GetOrAddReal("problem", "polytropic_index", dk.gamma_gas);
    if (!(dk.Gamma >= 1)) throw std::runtime_error("problem/gamma needs to be >= 1");
    dk.dens_min = pin.GetOrAddReal("problem", "dens_min", 1.0e-5);
    dk.pres_min = pin.GetOrAddReal("problem", "pres_min", 1.0e-8);
    dk.rexp = pin.GetOrAddReal("problem", "rexp", 0.0);
    dk.rcav = pin.GetOrAddReal("problem", "rcav", 0.0);
    dk.l0 = pin.GetOrAddReal("problem", "l0", 0.0);
    dk.dust_to_gas = pin.GetOrAddReal("problem", "dust_to_gas", 0.01);
    dk.temp_soft2 = pin.GetOrAddReal("problem", "temp_soft", 0.0);
    const Real big = -DBL_MAX; // -Big<Real>()
    Real q = pin.GetOrAddReal("problem", "tslope", big);
    Real flare = pin.GetOrAddReal("problem", "flare", big);
    if (!((flare != big) || (q != big))) throw std::runtime_error("Set flare or tslope in <problem>");
    if (flare == big) flare = 0.5 * (1.0 + q);
    else if (q == big) q = 2.0 * flare - 1.;
    else throw std::runtime_error("Set either flare or tslope in <problem> not both!");
    dk.flare = flare, dk.q = q;
    dk.quiet_start = pin.GetOrAddBoolean("problem", "quiet_start", false);
    dk.omf = do_rframe ? rf_omega : 0.0;
    if (do_viscosity) {
      const std::string vtype = pin.GetString("gas/viscosity", "type");
      if (vtype == "alpha") {
        dk.alpha = pin.GetReal("gas/viscosity", "alpha");
        dk.nu0 = dk.alpha * dk.gamma_gas * SQR(dk.h0 * dk.r0 * dk.Omega0);
        dk.nu_indx = 1.5 + dk.q;
      } else if (vtype == "powerlaw") {
        dk.nu0 = pin.GetReal("gas/viscosity", "nu");
        dk.nu_indx = pin.GetOrAddReal("gas/viscosity", "r_exp", 0.0);
      } else {
        throw std::runtime_error("Disk pgen is only compatible with alpha or powerlaw viscosity");
      }
      if (pin.DoesParameterExist("problem", "mdot")) {
        dk.mdot = pin.GetReal("problem", "mdot");
        dk.rho0 = dk.mdot / (3.0 * M_PI * dk.nu0);
      } else {
        dk.mdot = 3.0 * M_PI * dk.nu0 * dk.rho0;
      }
    }
    if (pin.GetOrAddBoolean("problem", "nbody_temp", false) && pin.GetOrAddBoolean("physics", "nbody", false))
      throw std::runtime_error("problem/nbody_temp needs the n-body package (out of scope of this build)");
    bool any_ic = false;
    for (const Block &B : blocks)
      for (int f = 0; f < 6; ++f) any_ic = any_ic || B.bc[f] == ARTEMIS_BC_IC;
    if (any_ic) {
      ic_gas.alloc(nb, 6 * ns_gas, N), ic_dust.alloc(nb, 4 * ns_dust, N);
      bcpar.ic_gas = ic_gas.tab(), bcpar.ic_dust = do_dust ? ic_dust.tab() : nullptr;
    }
    bcpar.disk_omf = dk.omf;
    bcpar.disk_nu0 = dk.nu0, bcpar.disk_nu_indx = dk.nu_indx, bcpar.disk_r0 = dk.r0, bcpar.disk_mdot = dk.mdot;
    if (coords == ARTEMIS_CARTESIAN)
      for (const Block &B : blocks)
        for (int f = 0; f < 6; ++f)
          if (B.bc[f] == ARTEMIS_BC_DISK_VISC) // disk.hpp:420-424
            throw std::runtime_error("Viscous boundary conditions only work with spherical/cylindrical radial boundaries");
  }
  // DenProfile / TempProfile / PresProfile / ViscosityProfile (disk.hpp:69-135) on the host libm
  auto disk_den = [&](const Real R, const Real z) {
    const Real r = std::sqrt(R * R + z * z);
    const Real h = dk.h0 * std::pow(R / dk.r0, dk.flare);
    const Real sig0 = dk.rho0;
    const Real exp_fac = (dk.rexp == 0.) ? 1. : std::exp(-SQR(R / dk.rexp));
    const Real dmid = (sig0 * std::pow(R / dk.r0, dk.p)) * (1. - dk.l0 * std::sqrt(dk.r0 / R)) *
                      (dk.dens_min / dk.rho0 +
                       (1. - dk.dens_min / dk.rho0) * std::exp(-std::pow(dk.rcav / R, 12.0))) *
                      exp_fac;
    const Real sint = (r == 0.0) ? 1.0 : R / r;
    const Real efac = (1. - sint) / (h * h);
    if (dk.Gamma == 1.) return std::max(dk.dens_min, dmid * std::exp(-efac));
    const Real pfac = 1. - (dk.Gamma - 1) * efac;
    return std::max(dk.dens_min, dmid * std::pow(pfac + 1e-99, 1. / (dk.Gamma - 1)));
  };
  auto disk_temp = [&](const Real R, const Real z) {
    const Real rho = disk_den(R, z);
    const Real rho0 = disk_den(R, 0.0);
    const Real H = R * dk.h0 * std::pow(R / dk.r0, dk.flare);
    const Real ir1 = 1.0 / std::sqrt(R * R + dk.temp_soft2);
    const Real omk2 = SQR(dk.Omega0) * ir1 * ir1 * ir1;
    const Real T0 = omk2 * H * H / dk.Gamma;
    return T0 * std::pow(rho / rho0, dk.Gamma - 1.0);
  };
  struct { Real g_rho = 1, g_v[3] = {0, 0, 0}, g_temp = 1; } cd;
  if (pgen == PG_COND) {
    if (!do_gas || ns_gas != 1 || do_dust) throw std::runtime_error("Cond pgen requires a single gas species.");
    cd.g_rho = pin.GetOrAddReal("problem", "gas_rho", 1.0);
    cd.g_v[0] = pin.GetOrAddReal("problem", "gas_vx1", 0.0), cd.g_v[1] = pin.GetOrAddReal("problem", "gas_vx2", 0.0);
    cd.g_v[2] = pin.GetOrAddReal("problem", "gas_vx3", 0.0);
    cd.g_temp = pin.GetOrAddReal("problem", "gas_temp", 1.0);
  }
  struct { Real xc[3] = {0, 0, 0}, sig = 1, dfac = 0, tfac = 0, ufac = 0, vfac = 0, wfac = 0, g_rho = 1, g_v[3] = {0, 0, 0}, g_pres = 1; } bp;
  if (pgen == PG_BUMP) { // gaussian_bump.hpp:55-74
    if (!do_gas || ns_gas != 1 || do_dust) throw std::runtime_error("Gaussian bump pgen requires a single gas species (dust is not built).");
    if (pin.GetString("problem", "system") != "cartesian")
      throw std::runtime_error("gaussian_bump pgen: only problem/system = cartesian is built");
    bp.xc[0] = pin.GetOrAddReal("problem", "x1c", 0.0), bp.xc[1] = pin.GetOrAddReal("problem", "x2c", 0.0);
    bp.xc[2] = pin.GetOrAddReal("problem", "x3c", 0.0);
    bp.sig = pin.GetReal("problem", "sigma");
    bp.dfac = pin.GetOrAddReal("problem", "density_bump", 0.0);
    bp.tfac = pin.GetOrAddReal("problem", "temperature_bump", 0.0);
    bp.ufac = pin.GetOrAddReal("problem", "vx1_bump", 0.0), bp.vfac = pin.GetOrAddReal("problem", "vx2_bump", 0.0);
    bp.wfac = pin.GetOrAddReal("problem", "vx3_bump", 0.0);
    bp.g_rho = pin.GetOrAddReal("problem", "gas_rho", 1.0);
    bp.g_v[0] = pin.GetOrAddReal("problem", "gas_vx1", 0.0), bp.g_v[1] = pin.GetOrAddReal("problem", "gas_vx2", 0.0);
    bp.g_v[2] = pin.GetOrAddReal("problem", "gas_vx3", 0.0);
    bp.g_pres = pin.GetOrAddReal("problem", "gas_pres", 1.0);
  }
  // One block's initial primitives on the host (libm, like the reference's CPU build).  Blocks are independent and
  // nothing below writes shared state, so a refined mesh (thousands of small blocks, 20 s single-threaded for
  // inputs/disk/disk_cart.in) is generated by a few host threads; uploads stay on this thread's stream, in order.
  // only_ic_ghosts (a remesh: the state itself is handed over by adopt_state_from): the generated values are read as the
  // `ic` condition's states alone, i.e. in the ghost zones behind the block's `ic` faces -- a tenth of the block; the
  // host libm evaluations of the other zones (a quarter of a second per 270 new blocks) are skipped, their entries stay 0
  auto generate_block = [&](const int b, std::vector<Real> &hg, std::vector<Real> &hd, const bool only_ic_ghosts = false) {
    std::fill(hg.begin(), hg.end(), 0.0);
    std::fill(hd.begin(), hd.end(), 0.0);
    const int g3[3] = {ng, ndim > 1 ? ng : 0, ndim > 2 ? ng : 0};
    auto behind_ic_face = [&](int k, int j, int i) {
      const int idx[3] = {i, j, k};
      for (int d = 0; d < ndim; ++d) {
        if (idx[d] < g3[d] && blocks[b].bc[2 * d] == ARTEMIS_BC_IC) return true;
        if (idx[d] >= g3[d] + mbnx[d] && blocks[b].bc[2 * d + 1] == ARTEMIS_BC_IC) return true;
      }
      return false;
    };
    // The disk profile in cylindrical coordinates is a function of (x1, x3) alone -- ComputeDiskProfile reads R and z,
    // the x2 faces of the pressure gradient sit at the centre's R and z and enter with a zero unit-vector component --
    // and it is ~150 libm calls per zone: a zone takes the values of the zone one row down (same i, same k) when that
    // one was computed.  The same bits as evaluating it again, a twentieth of the time of a remesh's generator pass.
    const bool row_memo = (pgen == PG_DISK && coords == ARTEMIS_CYLINDRICAL);
    std::vector<long> memo_cell(row_memo ? ni : 0, -1);
    for (int k = 0; k < nk; ++k) {
      if (row_memo) std::fill(memo_cell.begin(), memo_cell.end(), -1L);
      for (int j = 0; j < nj; ++j)
        for (int i = 0; i < ni; ++i) {
          if (only_ic_ghosts && !behind_ic_face(k, j, i)) continue;
          if (row_memo) {
            const size_t c_ = (static_cast<size_t>(k) * nj + j) * ni + i;
            if (memo_cell[i] >= 0) {
              const size_t m_ = static_cast<size_t>(memo_cell[i]);
              for (int v = 0; v < 6 * ns_gas; ++v) hg[v * N + c_] = hg[v * N + m_];
              for (int v = 0; v < 4 * ns_dust; ++v) hd[v * N + c_] = hd[v * N + m_];
              continue;
            }
            memo_cell[i] = static_cast<long>(c_);
          }
          const Real b1[2] = {xf(b, 0, i), xf(b, 0, i + 1)}, b2[2] = {xf(b, 1, j), xf(b, 1, j + 1)};
          const Real b3[2] = {xf(b, 2, k), xf(b, 2, k + 1)};
          const Real xv[3] = {0.5 * (b1[0] + b1[1]), 0.5 * (b2[0] + b2[1]), 0.5 * (b3[0] + b3[1])};
          const size_t c = (static_cast<size_t>(k) * nj + j) * ni + i;
          if (pgen == PG_CONSTANT) { // constant.hpp:100-163, problem/system = cartesian
            const Real ex1[3] = {1.0, 0.0, 0.0}, ex2[3] = {0.0, 1.0, 0.0}, ex3[3] = {0.0, 0.0, 1.0};
            if (do_gas) {
              hg[0 * N + c] = cs.g_rho;
              hg[(ns_gas + 0) * N + c] = (cs.g_v[0] * ex1[0] + cs.g_v[1] * ex1[1] + cs.g_v[2] * ex1[2]);
              hg[(ns_gas + 1) * N + c] = (cs.g_v[0] * ex2[0] + cs.g_v[1] * ex2[1] + cs.g_v[2] * ex2[2]);
              hg[(ns_gas + 2) * N + c] = (cs.g_v[0] * ex3[0] + cs.g_v[1] * ex3[1] + cs.g_v[2] * ex3[2]);
              hg[(5 * ns_gas) * N + c] = std::max(0.0, cv * cs.g_temp);
            }
            for (int n = 0; n < ns_dust; ++n) {
              hd[n * N + c] = cs.d_rho;
              hd[(ns_dust + 3 * n + 0) * N + c] = (cs.d_v[0] * ex1[0] + cs.d_v[1] * ex1[1] + cs.d_v[2] * ex1[2]);
              hd[(ns_dust + 3 * n + 1) * N + c] = (cs.d_v[0] * ex2[0] + cs.d_v[1] * ex2[1] + cs.d_v[2] * ex2[2]);
              hd[(ns_dust + 3 * n + 2) * N + c] = (cs.d_v[0] * ex3[0] + cs.d_v[1] * ex3[1] + cs.d_v[2] * ex3[2]);
            }
          } else if (pgen == PG_DISK) { // disk.hpp:141-247 ComputeDiskProfile + :325-354 DiskICImpl
            const artemis::DCoords co = cell_coords(b, k, j, i);
            Real xc[3];
            co.centre(xc);
            auto frame = [&](const Real xi[3]) { // ConvertToCylWithVec of any point, host libm
              return artemis::cyl_frame(coords, xi, std::cos(xi[1]), std::sin(xi[1]));
            };
            const artemis::Frame fr = frame(xc);
            const Real *xcyl = fr.x;
            const Real gdens = disk_den(xcyl[0], xcyl[2]);
            const Real rt = xcyl[0];
            const Real gtemp = disk_temp(rt, xcyl[2]);
            // grad(P) restated literally: `(pfm = pres_min) ? pres_min : ...` ASSIGNS (disk.hpp:186, :203,
            // :220), so both face pressures collapse to pres_min whenever pres_min != 0
            const Real fpts[3][2][3] = {{{co.x1[0], xc[1], xc[2]}, {co.x1[1], xc[1], xc[2]}},
                                        {{xc[0], co.x2[0], xc[2]}, {xc[0], co.x2[1], xc[2]}},
                                        {{xc[0], xc[1], co.x3[0]}, {xc[0], xc[1], co.x3[1]}}};
            const Real widths[3] = {co.width1(), co.width2(), co.width3()};
            auto pres = [&](const Real tf, const Real R, const Real z) {
              return std::max(dk.pres_min, std::max(0.0, (gamma - 1.0) * disk_den(R, z) * cv * tf));
            };
            Real pgrad[3];
            for (int d = 0; d < 3; ++d) {
              artemis::Frame xfc = frame(fpts[d][0]);
              const Real tfm = disk_temp(xfc.x[0], xfc.x[2]);
              Real pfm = pres(tfm, xfc.x[0], xfc.x[2]);
              xfc = frame(fpts[d][1]);
              const Real tfp = disk_temp(xfc.x[0], xfc.x[2]);
              const Real pfp = (pfm = dk.pres_min) ? dk.pres_min : pres(tfp, xfc.x[0], xfc.x[2]);
              pfm = (pfp == dk.pres_min) ? dk.pres_min : pfm;
              pgrad[d] = (pfp - pfm) / widths[d];
            }
            const Real eR[3] = {fr.e1[0], fr.e2[0], fr.e3[0]};
            const Real dpdr = pgrad[0] * eR[0] + pgrad[1] * eR[1] + pgrad[2] * eR[2];
            const Real r = std::sqrt(SQR(xcyl[0]) + SQR(xcyl[2]));
            const Real omk2 = dk.gm / (r * r * r);
            const Real vk2 = omk2 * SQR(xcyl[0]);
            const Real vp = std::sqrt(vk2 + dpdr * xcyl[0] / gdens);
            const Real nu = dk.nu0 * std::pow(rt / dk.r0, dk.nu_indx);
            const Real vr = dk.quiet_start ? 0.0 : -1.5 * nu / xcyl[0];
            const Real vcyl[3] = {vr, vp - dk.omf * xcyl[0], 0.0};
            auto vdot = [](const Real a[3], const Real w[3]) { return a[0] * w[0] + a[1] * w[1] + a[2] * w[2]; };
            hg[0 * N + c] = gdens;
            hg[(ns_gas + 0) * N + c] = vdot(vcyl, fr.e1);
            hg[(ns_gas + 1) * N + c] = vdot(vcyl, fr.e2);
            hg[(ns_gas + 2) * N + c] = vdot(vcyl, fr.e3);
            hg[(5 * ns_gas) * N + c] = cv * gtemp;
            const Real ddens = dk.dust_to_gas * gdens;
            const Real vkep[3] = {0.0, std::sqrt(vk2) - dk.omf * xcyl[0], 0.0};
            for (int n = 0; n < ns_dust; ++n) {
              hd[n * N + c] = ddens;
              hd[(ns_dust + 3 * n + 0) * N + c] = vdot(vkep, fr.e1);
              hd[(ns_dust + 3 * n + 1) * N + c] = vdot(vkep, fr.e2);
              hd[(ns_dust + 3 * n + 2) * N + c] = vdot(vkep, fr.e3);
            }
          } else if (pgen == PG_COND) { // conduction.hpp:88-103
            const Real gm1c = gamma - 1.0;
            const Real gx1 = (do_gravity && grav.type == ARTEMIS_GRAVITY_UNIFORM) ? grav.g[0] : 0.0;
            const Real P0 = std::max(0.0, gm1c * cd.g_rho * cv * cd.g_temp);
            const Real Rgas = P0 / (cd.g_rho * cd.g_temp);
            const Real x1c = cell_coords(b, k, j, i).x1v(); // coords.GetCellCenter()[0]
            const Real Pz = P0 * std::exp(gx1 * cd.g_rho / P0 * (x1c - xmin[0]));
            const Real dens = Pz / (Rgas * cd.g_temp);
            hg[0 * N + c] = dens;
            hg[(ns_gas + 0) * N + c] = cd.g_v[0];
            hg[(ns_gas + 1) * N + c] = cd.g_v[1];
            hg[(ns_gas + 2) * N + c] = cd.g_v[2];
            hg[(5 * ns_gas) * N + c] = std::max(0.0, cv * cd.g_temp);
          } else if (pgen == PG_BUMP) { // gaussian_bump.hpp:112-178, problem/system = cartesian
            const Real dxs = SQR(xv[0] - bp.xc[0]) + SQR(xv[1] - bp.xc[1]) * (ndim >= 2) +
                             SQR(xv[2] - bp.xc[2]) * (ndim == 3);
            const Real bump = std::exp(-dxs / (2.0 * SQR(bp.sig)));
            const Real ex1[3] = {1.0, 0.0, 0.0}, ex2[3] = {0.0, 1.0, 0.0}, ex3[3] = {0.0, 0.0, 1.0};
            const Real vx1 = (bp.g_v[0] * ex1[0] + bp.g_v[1] * ex1[1] + bp.g_v[2] * ex1[2]);
            const Real vx2 = (bp.g_v[0] * ex2[0] + bp.g_v[1] * ex2[1] + bp.g_v[2] * ex2[2]);
            const Real vx3 = (bp.g_v[0] * ex3[0] + bp.g_v[1] * ex3[1] + bp.g_v[2] * ex3[2]);
            hg[(ns_gas + 0) * N + c] = vx1 + bp.ufac * bump;
            hg[(ns_gas + 1) * N + c] = vx2 + bp.vfac * bump;
            hg[(ns_gas + 2) * N + c] = vx3 + bp.wfac * bump;
            if (bp.tfac > 0.0) {
              const Real sie0 = bp.g_pres / (bp.g_rho * (gamma - 1.0));
              const Real sie = sie0 * (1. + bp.tfac * bump);
              hg[0 * N + c] = bp.g_pres / (sie * (gamma - 1.0));
              hg[(5 * ns_gas) * N + c] = sie;
            } else {
              const Real dens = bp.g_rho * (1. + bp.dfac * bump);
              hg[0 * N + c] = dens;
              hg[(5 * ns_gas) * N + c] = bp.g_pres / ((gamma - 1.0) * dens);
            }
          } else if (pgen == PG_STRAT) { // strat.hpp:116-148
            const Real x = xv[0];
            const Real z = xv[2];
            const Real vx1 = 0.0;
            const Real vx2 = -rf_qshear * rf_omega * x;
            const Real vx3 = 0.0;
            const Real temp = SQR(st.h * rf_omega);
            const Real efac = (ndim == 3) ? std::exp(-SQR(z) / (2.0 * SQR(st.h))) : 1.0;
            const Real dens = std::max(st.dens_min, efac * st.rho0);
            const Real sie = std::max(0.0, cv * temp);
            hg[0 * N + c] = dens;
            hg[(ns_gas + 0) * N + c] = vx1;
            hg[(ns_gas + 1) * N + c] = vx2;
            hg[(ns_gas + 2) * N + c] = vx3;
            hg[(5 * ns_gas) * N + c] = sie;
            const Real ddens = dens * st.d2g;
            for (int n = 0; n < ns_dust; ++n) {
              hd[n * N + c] = ddens;
              hd[(ns_dust + 3 * n + 0) * N + c] = vx1;
              hd[(ns_dust + 3 * n + 1) * N + c] = vx2;
              hd[(ns_dust + 3 * n + 2) * N + c] = vx3;
            }
          } else if (pgen == PG_BLAST) { // blast.hpp:168-228
            const artemis::DCoords co = cell_coords(b, k, j, i);
            const Real total_vol = co.volume();
            const Real e0 = p0 / gm1;
            // cell centroid and blast centre in Cartesian coordinates (blast.hpp:181-185)
            const Real xcen[3] = {co.x1v(), co.x2v(), co.x3v()};
            Real xc[3], xb[3];
            to_cart(coords, xcen, xc);
            to_cart(coords, x0, xb);
            for (int n = 0; n < 3; n++) xc[n] -= xb[n];
            Real vol;
            if (samples > 0 && coords == ARTEMIS_AXISYMMETRIC && btype == 1) {
              // compute_overlap_sph, axisymmetric branch (blast.hpp:107-121): r-weighted samples
              const Real dxf = (b1[1] - b1[0]) / (Real)samples, dyf = (b2[1] - b2[0]) / (Real)samples;
              const Real dV = dxf * dyf;
              Real tot = 0.0;
              for (int ii = 0; ii < samples; ii++) {
                const Real xs = b1[0] + (ii + 0.5) * dxf;
                for (int jj = 0; jj < samples; jj++) {
                  const Real ys = b2[0] + (jj + 0.5) * dyf;
                  if (SQR(xs) + SQR(ys) <= SQR(rinit)) tot += xs * dV;
                }
              }
              vol = tot;
            } else if (samples > 0 && coords != ARTEMIS_CARTESIAN) {
              vol = 0.0; // the other overlap helpers have a Cartesian branch only (blast.hpp:68,122)
            } else if (samples > 0) {
              // compute_overlap_{sph,cyl} (blast.hpp:65-106): sub-sample the zone.  Zones whose
              // sample points are all outside (inside) the sphere are counted directly; the
              // counts, and so the products below, are the same as the loops would give.
              const bool sph = (btype == 1);
              const Real dxf = (b1[1] - b1[0]) / (Real)samples, dyf = (b2[1] - b2[0]) / (Real)samples;
              const Real dzf = (b3[1] - b3[0]) / (Real)samples;
              auto nearest = [](const Real *bb) { return (bb[0] > 0) ? bb[0] : ((bb[1] < 0) ? -bb[1] : 0.0); };
              auto farthest = [](const Real *bb) { return std::max(std::abs(bb[0]), std::abs(bb[1])); };
              const Real nz = sph ? nearest(b3) : 0.0, fz = sph ? farthest(b3) : 0.0;
              const Real dmin2 = SQR(nearest(b1)) + SQR(nearest(b2)) + SQR(nz);
              const Real dmax2 = SQR(farthest(b1)) + SQR(farthest(b2)) + SQR(fz);
              long tot = 0;
              const long full = sph ? (long)samples * samples * samples : (long)samples * samples;
              if (dmin2 > SQR(rinit) * (1.0 + 1e-12)) {
                tot = 0;
              } else if (dmax2 < SQR(rinit) * (1.0 - 1e-12)) {
                tot = full;
              } else {
                for (int ii = 0; ii < samples; ii++) {
                  const Real xs = b1[0] + (ii + 0.5) * dxf;
                  for (int jj = 0; jj < samples; jj++) {
                    const Real ys = b2[0] + (jj + 0.5) * dyf;
                    if (sph) {
                      for (int kk = 0; kk < samples; kk++) {
                        const Real zs = b3[0] + (kk + 0.5) * dzf;
                        if (SQR(xs) + SQR(ys) + SQR(zs) <= SQR(rinit)) tot++;
                      }
                    } else if (SQR(xs) + SQR(ys) <= SQR(rinit)) {
                      tot++;
                    }
                  }
                }
              }
              vol = sph ? tot * dxf * dyf * dzf : tot * dxf * dyf;
            } else {
              vol = ((SQR(xc[0]) + SQR(xc[1]) + SQR(xc[2]) < rinit * rinit) ? total_vol : 0.0);
            }
            Real ie_;
            if (btype == 1)
              ie_ = e0 * (1.0 - vol / total_vol) +
                    ienergy * vol / total_vol / (4.0 * M_PI / 3.0 * rinit * rinit * rinit);
            else
              ie_ = e0 * (1.0 - vol / total_vol) + ienergy * vol / total_vol / (M_PI * rinit * rinit);
            hg[0 * N + c] = d0;
            hg[(5 * ns_gas) * N + c] = ie_ / d0;
          } else { // linear_wave.hpp:231-258 / advection.hpp:185-216
            const Real x = lw.cos_a2 * (xv[0] * lw.cos_a3 + xv[1] * lw.sin_a3) + xv[2] * lw.sin_a2;
            const Real sn = std::sin(lw.k_par * x);
            Real cd, cm1, cm2, cm3, ce;
            if (pgen == PG_LINWAVE) {
              const int wf = lw.wave_flag;
              const Real mx = lw.d0 * lw.vflow + lw.amp * sn * lw.rem[1][wf];
              const Real my = lw.amp * sn * lw.rem[2][wf];
              const Real mz = lw.amp * sn * lw.rem[3][wf];
              cd = lw.d0 + lw.amp * sn * lw.rem[0][wf];
              cm1 = mx * lw.cos_a2 * lw.cos_a3 - my * lw.sin_a3 - mz * lw.sin_a2 * lw.cos_a3;
              cm2 = mx * lw.cos_a2 * lw.sin_a3 + my * lw.cos_a3 - mz * lw.sin_a2 * lw.sin_a3;
              cm3 = mx * lw.sin_a2 + mz * lw.cos_a2;
              ce = lw.p0 / lw.gm1 + 0.5 * lw.d0 * (lw.v1_0) * (lw.v1_0) + lw.amp * sn * lw.rem[4][wf];
            } else {
              const Real mx = lw.d0 * lw.vflow + lw.amp * sn * lw.v1_0;
              cd = lw.d0 + lw.amp * sn;
              cm1 = mx * lw.cos_a2 * lw.cos_a3;
              cm2 = mx * lw.cos_a2 * lw.sin_a3;
              cm3 = mx * lw.sin_a2;
              ce = lw.p0 / lw.gm1 + 0.5 * lw.d0 * SQR(lw.v1_0) + 0.5 * lw.d0 * lw.amp * sn * SQR(lw.v1_0);
            }
            const Real cu = ce - 0.5 * (SQR(cm1) + SQR(cm2) + SQR(cm3)) / cd;
            if (do_gas) {
              hg[0 * N + c] = cd;
              hg[(ns_gas + 0) * N + c] = cm1 / cd;
              hg[(ns_gas + 1) * N + c] = cm2 / cd;
              hg[(ns_gas + 2) * N + c] = cm3 / cd;
              hg[(5 * ns_gas) * N + c] = cu / cd;
            }
            if (do_dust && pgen == PG_ADVECTION) {
              hd[0 * N + c] = cd;
              hd[(ns_dust + 0) * N + c] = cm1 / cd;
              hd[(ns_dust + 1) * N + c] = cm2 / cd;
              hd[(ns_dust + 2) * N + c] = cm3 / cd;
              hd[1 * N + c] = cd;
              hd[(ns_dust + 3) * N + c] = -cm1 / cd;
              hd[(ns_dust + 4) * N + c] = -cm2 / cd;
              hd[(ns_dust + 5) * N + c] = -cm3 / cd;
            }
          }
        }
    }
  };
  {
    const size_t ng_ = static_cast<size_t>(6) * ns_gas * N, nd_ = static_cast<size_t>(4) * ns_dust * N;
    int nthreads = 1;
    if (nb >= 8) {
      nthreads = static_cast<int>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u));
      if (artemis::opt(artemis::OPT_HOST_THREADS) > 0) nthreads = static_cast<int>(artemis::opt(artemis::OPT_HOST_THREADS));
      nthreads = std::min(nthreads, nb);
    }
    // batches of `nthreads` blocks: generate concurrently, then upload in block order
    std::vector<std::vector<Real>> hgs(nthreads, std::vector<Real>(ng_)), hds(nthreads, std::vector<Real>(nd_));
    // (a remesh during the run: only the blocks whose `ic` faces keep reading the generated state need it, and of
    //  those only the ones the old state cannot hand over -- a handful, so the batches run over that list)
    auto wanted = [&](int b) {
      if (!adopting) return true;
      if (!ic_gas.ok()) return false;
      if (reuse_from && reuse_from->ic_gas.ok() && reuse_block(b) >= 0) return false; // copied from the old state below
      for (int f = 0; f < 6; ++f)
        if (blocks[b].bc[f] == ARTEMIS_BC_IC) return true;
      return false;
    };
    std::vector<int> todo;
    for (int b = 0; b < nb; ++b)
      if (wanted(b)) todo.push_back(b);
    const int ntodo = static_cast<int>(todo.size());
    for (int q0 = 0; q0 < ntodo; q0 += nthreads) {
      const int nbatch = std::min(nthreads, ntodo - q0);
      std::vector<std::thread> pool;
      std::vector<std::exception_ptr> errs(nbatch);
      for (int t = 1; t < nbatch; ++t)
        pool.emplace_back([&, t] {
          try {
            generate_block(todo[q0 + t], hgs[t], hds[t], adopting);
          } catch (...) {
            errs[t] = std::current_exception();
          }
        });
      try {
        generate_block(todo[q0], hgs[0], hds[0], adopting);
      } catch (...) {
        errs[0] = std::current_exception();
      }
      for (auto &th : pool) th.join();
      for (int t = 0; t < nbatch; ++t)
        if (errs[t]) std::rethrow_exception(errs[t]);
      for (int t = 0; t < nbatch; ++t) {
        const int b = todo[q0 + t];
        if (do_gas && !adopting) upload_block(gprim[0], b, hgs[t]); // (a remesh: adopt_state_from brings the state)
        if (do_dust && !adopting) upload_block(dprim[0], b, hds[t]);
        if (ic_gas.ok()) upload_block(ic_gas, b, hgs[t]); // as generated, before PrimToCons applies the floors
        if (ic_dust.ok()) upload_block(ic_dust, b, hds[t]);
      }
    }
  }
  pg_lap("generate + upload blocks");
  if (adopting && reuse_from && ic_gas.ok() && reuse_from->ic_gas.ok()) {
    auto ob = [&](int b) { return reuse_block(b); };
    copy_rows(ic_gas, reuse_from->ic_gas, ob);
    if (do_dust) copy_rows(ic_dust, reuse_from->ic_dust, ob);
  }
  pg_lap("copy ic rows");
  if (multilevel && ic_gas.ok()) {
    // The `ic` condition on a coarse buffer takes the profile at the buffer's own zone centres (the reference's
    // DiskBoundaryIC re-evaluates it wherever it is applied): run the generator once more on the coarse geometry
    // of every block whose buffer carries a physical condition.  The generator reads the block geometry through
    // the members below, so they are switched to the coarse buffers' for the duration.
    const int cg[3] = {ng, ndim > 1 ? ng : 0, ndim > 2 ? ng : 0};
    const int cnx[3] = {mbnx[0] / 2, ndim > 1 ? mbnx[1] / 2 : 1, ndim > 2 ? mbnx[2] / 2 : 1};
    const int fi = ni, fj = nj, fk = nk, fm[3] = {mbnx[0], mbnx[1], mbnx[2]};
    const size_t fN = N;
    std::vector<Real> fgeom, fmetric;
    fgeom.swap(hgeom), fmetric.swap(hmetric);
    hgeom = ml.cgeom_h, hmetric = ml.cmetric_h;
    ni = cnx[0] + 2 * cg[0], nj = cnx[1] + 2 * cg[1], nk = cnx[2] + 2 * cg[2];
    N = static_cast<size_t>(ni) * nj * nk;
    for (int d = 0; d < 3; ++d) mbnx[d] = cnx[d];
    try {
      ic_gas_c.alloc(nb, 6 * ns_gas, N), ic_dust_c.alloc(nb, 4 * ns_dust, N);
      ic_coarse_valid.assign(nb, 0);
      std::vector<Real> hgc(static_cast<size_t>(6) * ns_gas * N), hdc(static_cast<size_t>(4) * ns_dust * N);
      for (int b : ml_host.restrict_blocks) {
        bool any = false;
        for (int f = 0; f < 2 * ndim; ++f) any = any || blocks[b].bc[f] == ARTEMIS_BC_IC;
        if (!any) continue;
        ic_coarse_valid[b] = 1;
        const int ob = adopting ? reuse_block(b) : -1;
        if (ob >= 0 && reuse_from->ic_gas_c.ok() && ob < static_cast<int>(reuse_from->ic_coarse_valid.size()) && reuse_from->ic_coarse_valid[ob]) {
          if (do_gas) CK(artemis_rt_memcpy_d2d(ic_gas_c.var(b, 0), reuse_from->ic_gas_c.var(ob, 0), sizeof(Real) * 6 * ns_gas * N, stream), "d2d");
          if (do_dust) CK(artemis_rt_memcpy_d2d(ic_dust_c.var(b, 0), reuse_from->ic_dust_c.var(ob, 0), sizeof(Real) * 4 * ns_dust * N, stream), "d2d");
          continue;
        }
        generate_block(b, hgc, hdc, true); // (coarse buffers: the `ic` condition's ghost zones are all that is ever read)
        if (do_gas) upload_block(ic_gas_c, b, hgc);
        if (do_dust) upload_block(ic_dust_c, b, hdc);
      }
    } catch (...) {
      ni = fi, nj = fj, nk = fk, N = fN, hgeom.swap(fgeom), hmetric.swap(fmetric);
      for (int d = 0; d < 3; ++d) mbnx[d] = fm[d];
      throw;
    }
    ni = fi, nj = fj, nk = fk, N = fN, hgeom.swap(fgeom), hmetric.swap(fmetric);
    for (int d = 0; d < 3; ++d) mbnx[d] = fm[d];
  }
  pg_lap("coarse-buffer ic states");
  base = 0;
  if (do_cooling && do_gas) {
    cool.cv = cv_gas;
    cool_tref.alloc(nb, 1, N), cool_beta.alloc(nb, 1, N);
    const artemis_pack_t pk = make_pack(0);
    std::vector<Real> ht(N), hb(N);
    for (int b = 0; b < nb; ++b) {
      CK(artemis_hip_cooling_table_fill(&pk, hgeom.data(), hmetric.empty() ? nullptr : hmetric.data(), &cool, b,
                                        ht.data(), hb.data()), "cooling tables");
      upload_block(cool_tref, b, ht), upload_block(cool_beta, b, hb);
    }
    cool.tref = cool_tref.tab(), cool.beta = cool_beta.tab();
  }
  if (do_viscosity && (diff.visc.type == ARTEMIS_VISCOSITY_ALPHA || diff.visc.r_exp != 0.0)) {
    // the std::pow of the cell position in DiffusionCoeff::Get (diffusion_coeff.hpp:222-224, :262-264)
    visc_radial.alloc(nb, 1, N);
    const artemis_pack_t pk = make_pack(0);
    // host libm per zone (one std::pow): a few threads, each its own run of blocks; uploads in block order afterwards
    int nthreads = 1;
    if (nb >= 8) {
      nthreads = static_cast<int>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u));
      if (artemis::opt(artemis::OPT_HOST_THREADS) > 0) nthreads = static_cast<int>(artemis::opt(artemis::OPT_HOST_THREADS));
      nthreads = std::min(nthreads, nb);
    }
    const int chunk = 1024; // blocks per batch: bounds the host staging memory
    const bool can_copy = adopting && reuse_from && reuse_from->visc_radial.ok();
    std::vector<int> todo; // (a remesh: the blocks the old state cannot hand over)
    for (int b = 0; b < nb; ++b)
      if (!(can_copy && reuse_block(b) >= 0)) todo.push_back(b);
    const int ntodo = static_cast<int>(todo.size());
    std::vector<Real> hr(static_cast<size_t>(std::min(std::max(ntodo, 1), chunk)) * N);
    for (int q0 = 0; q0 < ntodo; q0 += chunk) {
      const int nbatch = std::min(chunk, ntodo - q0);
      std::vector<std::thread> pool;
      std::vector<int> rcs(nthreads, 0);
      auto work = [&](int t) {
        for (int q = t; q < nbatch; q += nthreads)
          if (int rc = artemis_hip_diffusion_radial_fill(&pk, hgeom.data(), hmetric.empty() ? nullptr : hmetric.data(),
                                                         &diff.visc, todo[q0 + q], hr.data() + static_cast<size_t>(q) * N))
            rcs[t] = rc;
      };
      for (int t = 1; t < std::min(nthreads, nbatch); ++t) pool.emplace_back(work, t);
      work(0);
      for (auto &th : pool) th.join();
      for (int t = 0; t < nthreads; ++t) CK(rcs[t], "viscosity radial table");
      for (int q = 0; q < nbatch;) { // runs of consecutive blocks: one copy each, one synchronisation per chunk
        int len = 1;
        while (q + len < nbatch && todo[q0 + q + len] == todo[q0 + q] + len) ++len;
        CK(artemis_rt_memcpy_h2d(visc_radial.var(todo[q0 + q], 0), hr.data() + static_cast<size_t>(q) * N, sizeof(Real) * N * len, stream), "h2d");
        q += len;
      }
      CK(artemis_rt_stream_sync(stream), "sync");
    }
    if (adopting && reuse_from && reuse_from->visc_radial.ok()) copy_rows(visc_radial, reuse_from->visc_radial, [&](int b) { return reuse_block(b); });
    diff.visc.radial = visc_radial.tab();
  }
  pg_lap("viscosity radial table");
  if ((do_viscosity || do_conduction) && !artemis::opt(artemis::OPT_NO_DISTANCE_TABLE)) {
    // Coords::Distance between neighbouring cell centres is geometry: tabulated once per mesh (a remesh builds
    // a new state through this constructor path, so the table follows the blocks)
    const artemis_pack_t pk = make_pack(0);
    diff_dist.alloc(artemis_hip_viscous_distance_count(&pk));
    CK(artemis_rt_device_sync(), "sync"); // (the allocation is cleared on the null stream)
    CK(artemis_hip_viscous_distance_fill(&pk, diff_dist.p, stream), "viscous distance table");
    diff.dist = diff_dist.p;
  }
  // PostInitialization = PrimToCons on every block (main.cpp:43, fill_derived.cpp:284-287),
  // then parthenon Mesh::Initialize communicates boundaries (upstream, recalled):
  // PreCommFillDerived (ConsToPrim, artemis.cpp:122) -> exchange + physical BCs ->
  // FillDerived (PrimToCons, artemis.cpp:123).
  pg_lap("distance table");
  if (adopting) { // adopt_state_from brings the state and runs this sequence on it
    CK(artemis_rt_stream_sync(stream), "sync");
    return;
  }
  const artemis_pack_t p = make_pack(0);
  CK(artemis_hip_prim_to_cons(&p, stream), "PrimToCons");
  CK(artemis_hip_cons_to_prim(&p, stream), "ConsToPrim");
  fill_ghosts(0);
  CK(artemis_hip_prim_to_cons(&p, stream), "PrimToCons");
  cons_valid = true;
  CK(artemis_rt_stream_sync(stream), "sync");
}

// What the old state of a remesh still owes the new one is its conserved variables (and the geometry they live on);
// everything else -- primitive ping-pong buffers, the start-of-step copy, flux and diffusion-flux arrays, coarse
// buffers, operation lists, message buffers, tables -- is returned to the allocator BEFORE the new state is built, so
// that the two meshes coexist at ~1.2 x instead of 2 x the memory.  The old state cannot step any more afterwards.
void artemis_sim_impl::release_for_adoption() {
  materialise_cons();
  CK(artemis_rt_stream_sync(stream), "sync");
  if (comm_stream) CK(artemis_rt_stream_sync(comm_stream), "sync");
  for (int q = 0; q < 3; ++q) gprim[q].release(), dprim[q].release();
  gu1.release(), du1.release();
  for (int d = 0; d < 3; ++d) gflux[d].release(), gpflux[d].release(), gvface[d].release(), dflux[d].release(), gdflux[d].release();
  gdsum.release(), diff_dist.release(); // (visc_radial and the `ic` states stay: the new state copies rows from them)
  cool_tref.release(), cool_beta.release();
  plmtab.release();
  ml.gcoarse.release(), ml.dcoarse.release(), ml.cgeom.release(), ml.cmetric.release();
  ml.ops_a.release(), ml.ops_u.release(), ml.ops_b.release(), ml.ops_fx.release(), ml.ops_fxu.release();
  ml.fine_boxes.release(), ml.fix_cells.release(), ml.restrict_blocks.release(), ml.boxes.release(), ml.floor_blocks.release();
  ml.gsend.release(), ml.grecv.release(), ml.fsend.release(), ml.frecv.release();
}

// ---------------------------------------------------------------------------------------
// Adaptive refinement.  Tagging = the gas package's CheckRefinementBlock (gas.cpp:305-380: ScalarFirstDerivative or
// ScalarMagnitude of the density / pressure of species 0) on every local block; the tree update and the rebuild live
// at the C handle (below), the data hand-over here.
std::vector<int> artemis_sim_impl::amr_tags() {
  std::vector<int> tags(nb, 0);
  if (!refine_field || !refine_type) return tags;
  // one launch for all blocks, one copy back (artemis_hip_amr_block_maxima); thresholds as the per-block calls.  The
  // pressure is recomputed from rho and sie (field 2: the bits PrimToCons over the entire block would leave in the
  // pressure slot, ghost zones included, without that pass)
  if (amr_maxima.n < static_cast<size_t>(nb)) amr_maxima.alloc(nb);
  DevBuf &maxima = amr_maxima;
  const artemis_pack_t p = make_pack(base);
  CK(artemis_hip_amr_block_maxima(&p, refine_field == 1 ? 0 : 2, refine_type == 2 ? 1 : 0, maxima.p, stream), "refinement criterion");
  std::vector<double> h(nb);
  CK(artemis_rt_memcpy_d2h(h.data(), maxima.p, sizeof(double) * nb, stream), "d2h");
  CK(artemis_rt_stream_sync(stream), "sync");
  for (int b = 0; b < nb; ++b) {
    if (refine_type == 1) { // ScalarFirstDerivative (amr_criteria.hpp:122-131); 1-D blocks are never tagged
      if (ndim == 1) continue;
      tags[b] = (h[b] > refine_thr) ? 1 : ((h[b] < 0.25 * refine_thr) ? -1 : 0);
    } else { // ScalarMagnitude (:160-167)
      tags[b] = (h[b] > refine_thr) ? 1 : ((h[b] < deref_thr) ? -1 : 0);
    }
  }
  return tags;
}

// The new state takes over from the old one (same deck, another set of leaves): blocks that exist in both are copied,
// children of a refined block are prolongated from it with ProlongateSharedMinMod, a derefined block is the
// RestrictAverage of its children -- all on the CONSERVED variables (mass, momentum, energy: what refinement
// must conserve), ghost zones of the old state included for the prolongation stencil.  Then the sequence every
// (re)initialisation ends with: ConsToPrim, boundary exchange + conditions, PrimToCons (parthenon Mesh::Initialize).
void artemis_sim_impl::adopt_state_from(artemis_sim_impl &old) {
  if (old.gprim[0].ok()) old.materialise_cons(); // (after release_for_adoption the conserved state is all there is)
  CK(artemis_rt_stream_sync(old.stream), "sync");
  typedef std::tuple<int, int, int, int> Key;
  auto key_of = [](const artemis_host::Leaf &l) { return Key(l.level, l.lx[0], l.lx[1], l.lx[2]); };
  // global ids: leaves are numbered in Z-order and dealt to the ranks in contiguous runs (build_mesh_multilevel); the
  // runs are equal counts or cost-weighted (<artemis_amd/loadbalance>), so each state carries its own split
  auto owner = [&](const artemis_sim_impl &S, long gid, int &local) {
    if (gid < 0 || gid >= static_cast<long>(S.split_rank.size())) throw std::runtime_error("adopt_state_from: leaf id outside the state's split");
    local = S.split_local[gid];
    return S.split_rank[gid];
  };
  const std::vector<artemis_host::Leaf> &oldL = old.tree_leaves, &newL = tree_leaves;
  std::map<Key, long> where;
  for (size_t g = 0; g < oldL.size(); ++g) where[key_of(oldL[g])] = static_cast<long>(g);
  const int s3[3] = {is, js, ks}, n3[3] = {mbnx[0], mbnx[1], mbnx[2]};
  const long mstride = artemis::metric_block_stride(nj, nk);
  // a block as the refinement operators see it: edge table, metric table, tables of its gas / dust arrays
  struct View {
    const double *geom, *metric;
    double *const *gas, *const *dust;
  };
  auto local_view = [&](const artemis_sim_impl &S, int b) {
    View v;
    v.geom = S.geom.p + 6 * b, v.metric = S.metric.p ? S.metric.p + b * mstride : nullptr;
    v.gas = S.do_gas ? S.gu0.tab() + static_cast<size_t>(b) * 6 * ns_gas : nullptr;
    v.dust = S.do_dust ? S.du0.tab() + static_cast<size_t>(b) * 4 * ns_dust : nullptr;
    return v;
  };
  // old blocks that live on another rank arrive in temporaries with tables of their own
  struct Remote {
    DevBuf gas, dust, geom, metric;
    DevArr gtab, dtab;
    View view;
  };
  std::map<long, std::unique_ptr<Remote>> remote; // by old global id
  auto remote_view = [&](long og) -> View {
    auto it = remote.find(og);
    if (it != remote.end()) return it->second->view;
    std::unique_ptr<Remote> R(new Remote());
    const artemis_host::Leaf &lf = oldL[og];
    double g6[6];
    for (int d = 0; d < 3; ++d) { // the block's edges exactly as build_mesh_multilevel computes them
      const int n = (d < ndim) ? (nblk[d] << lf.level) : 1;
      const Real rl = static_cast<Real>(lf.lx[d]) / n, rr = static_cast<Real>(lf.lx[d] + 1) / n;
      const Real lo = (lf.lx[d] == 0) ? xmin[d] : xmin[d] * (1.0 - rl) + xmax[d] * rl;
      const Real hi = (lf.lx[d] + 1 == n) ? xmax[d] : xmin[d] * (1.0 - rr) + xmax[d] * rr;
      const Real dx = (hi - lo) / mbnx[d];
      g6[2 * d] = lo - ((d < ndim) ? ng : 0) * dx, g6[2 * d + 1] = dx;
    }
    R->geom.alloc(6);
    CK(artemis_rt_memcpy_h2d(R->geom.p, g6, sizeof g6, stream), "h2d");
    artemis_pack_t p1;
    std::memset(&p1, 0, sizeof p1);
    p1.nblocks = 1, p1.nghost = ng, p1.nx1 = mbnx[0], p1.nx2 = mbnx[1], p1.nx3 = mbnx[2], p1.coords = coords;
    const long nm = artemis_hip_metric_count(&p1);
    if (nm > 0) {
      std::vector<Real> hm(nm, 0.0);
      CK(artemis_hip_metric_fill(&p1, g6, hm.data()), "metric of a migrating block");
      R->metric.alloc(nm);
      CK(artemis_rt_memcpy_h2d(R->metric.p, hm.data(), nm * sizeof(Real), stream), "h2d");
    }
    CK(artemis_rt_stream_sync(stream), "sync"); // g6 / hm are locals
    std::vector<double *> hg, hd;
    if (do_gas) {
      R->gas.alloc(static_cast<size_t>(6) * ns_gas * N);
      for (int v = 0; v < 6 * ns_gas; ++v) hg.push_back(R->gas.p + static_cast<size_t>(v) * N);
      R->gtab.upload(hg);
    }
    if (do_dust) {
      R->dust.alloc(static_cast<size_t>(4) * ns_dust * N);
      for (int v = 0; v < 4 * ns_dust; ++v) hd.push_back(R->dust.p + static_cast<size_t>(v) * N);
      R->dtab.upload(hd);
    }
    R->view.geom = R->geom.p, R->view.metric = R->metric.p;
    R->view.gas = static_cast<double *const *>(R->gtab.p), R->view.dust = static_cast<double *const *>(R->dtab.p);
    const View out = R->view;
    remote[og] = std::move(R);
    return out;
  };
  auto refine_args = [&](const View &fine, const View &coarse, const int child[3], bool gas_vars) {
    artemis_refine_t r;
    std::memset(&r, 0, sizeof r);
    r.coords = coords, r.ndim = ndim, r.nvar = gas_vars ? 6 * ns_gas : 4 * ns_dust;
    r.fni = r.cni = ni, r.fnj = r.cnj = nj, r.fnk = r.cnk = nk;
    r.fgeom = fine.geom, r.cgeom = coarse.geom, r.fmetric = fine.metric, r.cmetric = coarse.metric;
    r.fine = gas_vars ? fine.gas : fine.dust, r.coarse = gas_vars ? coarse.gas : coarse.dust;
    int lo[3], hi[3];
    for (int d = 0; d < 3; ++d) {
      const bool act = d < ndim;
      lo[d] = act ? s3[d] + child[d] * (n3[d] / 2) : s3[d];
      hi[d] = act ? lo[d] + n3[d] / 2 - 1 : s3[d];
    }
    r.cis = lo[0], r.cie = hi[0], r.cjs = lo[1], r.cje = hi[1], r.cks = lo[2], r.cke = hi[2];
    r.cib = lo[0], r.cjb = lo[1], r.ckb = lo[2], r.fib = s3[0], r.fjb = s3[1], r.fkb = s3[2];
    return r;
  };
  // every (old block -> new block) dependency of the new mesh, in the same order on every rank
  struct Dep {
    long og, ng_;      // old / new global id
    int kind;          // 0 same, 1 new block is a child of og, 2 og is a child of the new block
    int child[3];
  };
  std::vector<Dep> deps;
  for (size_t g = 0; g < newL.size(); ++g) {
    const artemis_host::Leaf &B = newL[g];
    auto it = where.find(key_of(B));
    if (it != where.end()) {
      deps.push_back(Dep{it->second, static_cast<long>(g), 0, {0, 0, 0}});
      continue;
    }
    if (B.level > 0) {
      int pl[3] = {0, 0, 0}, ch[3] = {0, 0, 0};
      for (int d = 0; d < 3; ++d) pl[d] = (d < ndim) ? B.lx[d] >> 1 : 0, ch[d] = (d < ndim) ? B.lx[d] & 1 : 0;
      it = where.find(Key(B.level - 1, pl[0], pl[1], pl[2]));
      if (it != where.end()) {
        deps.push_back(Dep{it->second, static_cast<long>(g), 1, {ch[0], ch[1], ch[2]}});
        continue;
      }
    }
    for (int c3 = 0; c3 < (ndim > 2 ? 2 : 1); ++c3)
      for (int c2 = 0; c2 < (ndim > 1 ? 2 : 1); ++c2)
        for (int c1 = 0; c1 < 2; ++c1) {
          it = where.find(Key(B.level + 1, 2 * B.lx[0] + c1, (ndim > 1 ? 2 * B.lx[1] + c2 : 0), (ndim > 2 ? 2 * B.lx[2] + c3 : 0)));
          if (it == where.end()) throw std::runtime_error("remesh: a new block has no counterpart in the old mesh");
          deps.push_back(Dep{it->second, static_cast<long>(g), 2, {c1, c2, c3}});
        }
  }
  // blocks that change rank travel whole (conserved variables, ghost zones included), one message per fluid
  std::vector<artemis_msg_t> msgs;
  std::map<std::pair<long, int>, bool> sent; // (old gid, destination rank): send an old block to a rank once
  for (size_t q = 0; q < deps.size(); ++q) {
    int ol, nl;
    const int orank = owner(old, deps[q].og, ol), nrank = owner(*this, deps[q].ng_, nl);
    if (orank == nrank) continue;
    const int tag = 9000000 + 2 * static_cast<int>(deps[q].og);
    if (orank == rank) {
      if (sent[std::make_pair(deps[q].og, nrank)]) continue;
      sent[std::make_pair(deps[q].og, nrank)] = true;
      artemis_msg_t m;
      m.peer = nrank, m.recv = nullptr;
      if (do_gas) m.tag = tag, m.send = old.gu0.var(ol, 0), m.count = static_cast<long>(6) * ns_gas * N, msgs.push_back(m);
      if (do_dust) m.tag = tag + 1, m.send = old.du0.var(ol, 0), m.count = static_cast<long>(4) * ns_dust * N, msgs.push_back(m);
    } else if (nrank == rank) {
      if (remote.count(deps[q].og)) continue;
      remote_view(deps[q].og);
      Remote &R = *remote[deps[q].og];
      artemis_msg_t m;
      m.peer = orank, m.send = nullptr;
      if (do_gas) m.tag = tag, m.recv = R.gas.p, m.count = static_cast<long>(6) * ns_gas * N, msgs.push_back(m);
      if (do_dust) m.tag = tag + 1, m.recv = R.dust.p, m.count = static_cast<long>(4) * ns_dust * N, msgs.push_back(m);
    }
  }
  exchange_messages(msgs);
  std::vector<int> same_local(nb, -1); // new local block -> old local block it is a copy of
  for (const Dep &D : deps) {
    int ol, nl;
    const int orank = owner(old, D.og, ol), nrank = owner(*this, D.ng_, nl);
    if (nrank != rank) continue;
    const View src = (orank == rank) ? local_view(old, ol) : remote_view(D.og);
    const View dst = local_view(*this, nl);
    if (D.kind == 0) { // unchanged block
      if (orank == rank) {
        same_local[nl] = ol; // copied in runs below
        continue;
      }
      if (do_gas) CK(artemis_rt_memcpy_d2d(gu0.var(nl, 0), remote[D.og]->gas.p, sizeof(Real) * 6 * ns_gas * N, stream), "d2d");
      if (do_dust) CK(artemis_rt_memcpy_d2d(du0.var(nl, 0), remote[D.og]->dust.p, sizeof(Real) * 4 * ns_dust * N, stream), "d2d");
    } else if (D.kind == 1) { // refined: prolongate my octant of the parent
      for (int gas_vars = 1; gas_vars >= 0; --gas_vars) {
        if ((gas_vars && !do_gas) || (!gas_vars && !do_dust)) continue;
        const artemis_refine_t r = refine_args(dst, src, D.child, gas_vars != 0);
        CK(artemis_hip_prolongate_minmod(&r, stream), "ProlongateSharedMinMod (remesh)");
      }
    } else { // derefined: restrict this child into its octant of me
      for (int gas_vars = 1; gas_vars >= 0; --gas_vars) {
        if ((gas_vars && !do_gas) || (!gas_vars && !do_dust)) continue;
        const artemis_refine_t r = refine_args(src, dst, D.child, gas_vars != 0);
        CK(artemis_hip_restrict_average(&r, stream), "RestrictAverage (remesh)");
      }
    }
  }
  {
    auto ob = [&](int b) { return same_local[b]; };
    if (do_gas) copy_rows(gu0, old.gu0, ob);
    if (do_dust) copy_rows(du0, old.du0, ob);
  }
  CK(artemis_rt_stream_sync(stream), "sync"); // the temporaries of migrated blocks go out of scope below
  time = old.time, dt = old.dt, ncycle = old.ncycle, tlim = old.tlim, nlim = old.nlim;
  old.flush_nbody_force();
  particle_force = old.particle_force;
  overlap = old.overlap, time_kernels = old.time_kernels;
  base = 0;
  if (artemis::opt(artemis::OPT_AMR_DEBUG)) { // energy integral of the conserved state before and after the hand-over
    auto total = [](artemis_sim_impl &S, int var) {
      double sum = 0.0;
      for (int b = 0; b < S.nb; ++b) {
        const std::vector<Real> h = S.download(S.gu0, b);
        for (int k = S.ks; k <= S.ke; ++k)
          for (int j = S.js; j <= S.je; ++j)
            for (int i = S.is; i <= S.ie; ++i)
              sum += h[static_cast<size_t>(var) * S.N + (static_cast<size_t>(k) * S.nj + j) * S.ni + i] * S.cell_coords(b, k, j, i).volume();
      }
      return sum;
    };
    std::fprintf(stderr, "[amr] cycle %ld blocks %d -> %d  E %.17g -> %.17g  D %.17g -> %.17g\n", old.ncycle, old.nb, nb,
                 total(old, 4 * ns_gas), total(*this, 4 * ns_gas), total(old, 0), total(*this, 0));
  }
  const artemis_pack_t p = make_pack(0);
  // What Mesh::Initialize does for a modified mesh (upstream): PreCommFillDerived = ConsToPrim, the boundary
  // exchange, FillDerived = PrimToCons (artemis.cpp:122-123).  SetAuxillaryFields is a task of the stage list, not a
  // package callback, and is NOT called here (fill_derived.cpp:28 says so): ConsToPrim takes the specific internal
  // energy from the internal-energy variable as it was prolongated / restricted, and PrimToCons rebuilds the total
  // energy from it -- a remesh therefore conserves mass and momentum to round-off and the total energy to the
  // O(dx^2) difference between the two energy variables, exactly as the reference.
  CK(artemis_hip_cons_to_prim(&p, stream), "ConsToPrim");
  fill_ghosts(0);
  CK(artemis_hip_prim_to_cons(&p, stream), "PrimToCons");
  cons_valid = true;
  CK(artemis_rt_stream_sync(stream), "sync");
  // the step's own estimate came from the old mesh; new fine blocks may need less (never more than it allowed)
  {
    Real est = new_dt_unfused();
    if (has_comm && nranks > 1 && comm.allreduce_min(comm.ctx, &est)) throw std::runtime_error("allreduce failed");
    dt = std::min(dt, est);
  }
}

// ---------------------------------------------------------------------------------------
// PostStepTasks -> EstimateTimestep (artemis_driver.cpp:279-297): min over packages, local.
Real artemis_sim_impl::new_dt_unfused() {
  const artemis_pack_t p = make_pack(base);
  *dt_host = DBL_MAX;
  CK(artemis_rt_memcpy_h2d(dt_dev.p, dt_host, sizeof(double), stream), "h2d");
  // gas.cpp:411-467 (cfl * min(hydro, viscous, conductive)) and dust.cpp:256-272 in one pass over the state
  CK(artemis_hip_timestep_all(&p, cfl_gas, cfl_dust, (do_gas && (do_viscosity || do_conduction)) ? &diff : nullptr, dt_dev.p, stream),
     "EstimateTimestepMesh");
  CK(artemis_rt_memcpy_d2h(dt_host, dt_dev.p, sizeof(double), stream), "d2h");
  CK(artemis_rt_stream_sync(stream), "sync");
  return *dt_host;
}

// One step on the fused path: one kernel per stage, primitives ping-ponged between buffers.
// One step on the general fused path: one cell-centred kernel per fluid and stage (plus the drag /
// SetAuxillaryFields / ConsToPrim trio when drag couples the fluids), primitives of both fluids
// ping-ponged between buffers exactly like the tuned path.
void artemis_sim_impl::step_general(bool want_dt, bool device_dt) {
  tiny_valid = false; // (another producer of the primitives)
  for (int q = 1; q < 3; ++q) {
    if (!gprim[q].ok()) gprim[q].alloc(nb, 6 * ns_gas, N);
    if (!dprim[q].ok()) dprim[q].alloc(nb, 4 * ns_dust, N);
  }
  // Viscosity without heat conduction: ZeroDiffusionFlux + ViscousFlux + the viscous part of DiffusionUpdate as one
  // source (artemis_hip_viscous_source: five sums per zone, no diffusion-flux arrays) where the march covers the pack
  bool visc_source = false;
  if (do_gas && do_viscosity && !do_conduction && !artemis::opt(artemis::OPT_NO_VISC_SOURCE)) {
    const artemis_pack_t p0 = make_pack(base);
    visc_source = artemis_hip_viscous_source_covers(&p0) != 0;
  }
  if (visc_source && !gdsum.ok()) gdsum.alloc(nb, 5, N);
  if (do_gas && (do_viscosity || do_conduction) && !visc_source)
    for (int d = 0; d < ndim; ++d)
      if (!gdflux[d].ok()) gdflux[d].alloc(nb, 4 * ns_gas, N);
  const int A = base;
  int cur = A;
  if (want_dt && !device_dt) {
    *dt_host = DBL_MAX;
    CK(artemis_rt_memcpy_h2d(dt_dev.p, dt_host, sizeof(double), stream), "h2d");
  }
  for (int stage = 1; stage <= nstages; ++stage) {
    const bool last = (stage == nstages);
    int out;
    if (last && cur != A) out = A;
    else out = (cur + 1) % 3 == A ? (cur + 2) % 3 : (cur + 1) % 3;
    const artemis_pack_t p = make_pack(cur);
    artemis_stage_general_args_t a;
    std::memset(&a, 0, sizeof a);
    a.gam0 = gam0[stage - 1], a.gam1 = gam1[stage - 1];
    a.beta_dt = beta[stage - 1] * dt, a.bdt = beta[stage - 1] * dt;
    a.pcm = (stage == 1 && integrator == "vl2");
    a.time = time;
    a.gas_in = gprim[cur].tab(), a.gas_u1 = gprim[A].tab(), a.gas_out = gprim[out].tab();
    a.dust_in = dprim[cur].tab(), a.dust_u1 = dprim[A].tab(), a.dust_out = dprim[out].tab();
    place_binary();
    a.gravity = do_gravity ? &grav : nullptr;
    a.rf_omega = do_rframe ? rf_omega : 0.0, a.rf_qshear = rf_qshear;
    drag.damp_visc = damp_to_visc ? &diff.visc : nullptr;
    a.drag = do_drag ? &drag : nullptr;
    a.cfl_gas = cfl_gas, a.cfl_dust = cfl_dust;
    a.dt_dev = (last && want_dt) ? (device_dt ? tstate.p + 2 : dt_dev.p) : nullptr;
    if (device_dt) a.beta_dt_dev = tstate.p + 3 + (stage - 1); // beta*dt stays on the device
    if (do_gravity && grav_nbody) nbody_stage_args(p, a, a.bdt);
    const bool diffuse = do_gas && (do_viscosity || do_conduction);
    if (diffuse) { // artemis_driver.cpp:189-194 on the stage's input primitives
      if (visc_source) {
        CK(artemis_hip_viscous_source(&p, &diff, a.bdt, a.beta_dt_dev, gdsum.tab(), stream), "viscous source");
        a.diffusion_sums = gdsum.tab();
      } else {
        if (do_viscosity) CK(artemis_hip_zero_viscous_flux(&p, &diff, stream), "Gas::ZeroDiffusionFlux + ViscousFlux");
        else CK(artemis_hip_zero_diffusion_flux(&p, stream), "Gas::ZeroDiffusionFlux");
        if (do_conduction) CK(artemis_hip_thermal_flux(&p, &diff, stream), "Gas::ThermalFlux");
      }
      a.diffusion = &diff;
    }
    if (do_cooling && do_gas) a.cooling = &cool;
    void *e0 = nullptr, *e1 = nullptr;
    if (time_kernels && kev.size() < kMaxTimedLaunches) {
      e0 = artemis_rt_event_create(), e1 = artemis_rt_event_create();
      CK(artemis_rt_event_record(e0, stream), "event");
    }
    general_variant = artemis_hip_stage_general_variant(&p, &a);
    // One block whose four faces carry the `strat` problem's conditions (inputs/ssheet: x1 extrap, x2 inflow) on the 2-D
    // row march: the kernel forms the conditions' values on the rows it loads and reads no ghost zone -- the two boundary
    // launches per stage (7 % of a 1024^2 stage) are gone; evolve() completes the ghost zones before it returns.
    bool strat_in_kernel = false;
    if (general_variant == 1 && nb == 1 && links.empty() && ndim == 2 && !artemis::opt(artemis::OPT_NO_STRAT_IN_KERNEL) &&
        bc_flat[0] == ARTEMIS_BC_STRAT_EXTRAP && bc_flat[1] == ARTEMIS_BC_STRAT_EXTRAP && bc_flat[2] == ARTEMIS_BC_STRAT_INFLOW &&
        bc_flat[3] == ARTEMIS_BC_STRAT_INFLOW) {
      a.strat_faces = 15, a.strat_qshear = bcpar.qshear, a.strat_omega = bcpar.omega;
      strat_in_kernel = artemis_hip_stage_general_variant(&p, &a) == 1;
      if (!strat_in_kernel) a.strat_faces = 0;
    }
    CK(artemis_hip_stage_general(&p, &a, stream), "stage_general");
    if (e0) {
      CK(artemis_rt_event_record(e1, stream), "event");
      kev.emplace_back(e0, e1);
    }
    if (diffuse && a.dt_dev) { // gas.cpp:435-467: the diffusive limits of the new state
      const artemis_pack_t pn = make_pack(out);
      CK(artemis_hip_diffusion_dt(&pn, &diff, cfl_gas, a.dt_dev, stream), "dt diffusion");
    }
    if (strat_in_kernel) ghosts_stale = true; // (no neighbour, no other condition: nothing else to fill)
    else fill_ghosts(out);
    cur = out;
  }
  base = cur;
  cons_valid = false;
}

// Gravity::NBodyGravity inside the stage (gravity.cpp:150-155 decides whether the task runs): the particle array on the
// device (re-sent when the host copy changed), the accelerations applied by the stage kernels, the seven sums per
// particle by one light pass over the stage's input primitives.
void artemis_sim_impl::nbody_stage_args(const artemis_pack_t &p, artemis_stage_general_args_t &a, Real bdt) {
  a.gravity = nullptr;
  if (!(time >= grav.tstart && time < grav.tstop) || particles.empty()) return;
  const int np = static_cast<int>(particles.size());
  const size_t bytes = sizeof(artemis_nbody_particle_t) * particles.size();
  if (!nb_dev.p) {
    nb_dev.alloc((bytes + sizeof(double) - 1) / sizeof(double));
    nb_force_dev.alloc(7 * particles.size());
  }
  const int rows = artemis_hip_nbody_force_scratch(&p);
  if (rows < 0) throw HipFail(std::string("nbody force scratch: ") + artemis_hip_last_error());
  if (nb_scratch.n < static_cast<size_t>(7) * np * rows) nb_scratch.alloc(static_cast<size_t>(7) * np * rows);
  if (nb_uploaded.size() != particles.size() || std::memcmp(nb_uploaded.data(), particles.data(), bytes) != 0) {
    CK(artemis_rt_stream_sync(stream), "sync"); // (kernels in flight read the old array)
    CK(artemis_rt_memcpy_h2d(nb_dev.p, particles.data(), bytes, stream), "h2d particles");
    CK(artemis_rt_stream_sync(stream), "sync");
    nb_uploaded = particles;
  }
  a.nbody_dev = reinterpret_cast<const artemis_nbody_particle_t *>(nb_dev.p);
  a.nbody_n = np;
  a.nbody_omf = (do_rframe && nbody_frame_correction) ? rf_omega : 0.0;
  CK(artemis_hip_nbody_force_sums(&p, a.nbody_dev, np, a.nbody_omf, bdt, a.beta_dt_dev, nb_scratch.p, nb_force_dev.p, stream),
     "NBodyGravity (force sums)");
}
// the device accumulators into the host rows (before anybody reads them, and before a remesh hands the rows on)
void artemis_sim_impl::flush_nbody_force() {
  if (!nb_force_dev.p || particles.empty()) return;
  std::vector<double> h(7 * particles.size());
  CK(artemis_rt_stream_sync(stream), "sync");
  CK(artemis_rt_memcpy_d2h(h.data(), nb_force_dev.p, h.size() * sizeof(double), stream), "d2h");
  CK(artemis_rt_stream_sync(stream), "sync");
  for (size_t q = 0; q < h.size(); ++q) particle_force[q] += h[q];
  CK(artemis_rt_memset(nb_force_dev.p, 0, h.size() * sizeof(double), stream), "memset");
}

void artemis_sim_impl::step_fused(bool want_dt, bool device_dt) {
  if (!tuned) {
    step_general(want_dt, device_dt);
    return;
  }
  for (int q = 1; q < 3; ++q)
    if (!gprim[q].ok()) gprim[q].alloc(nb, 6 * ns_gas, N);
  ensure_redo_scratch();
  const int A = base;
  int cur = A;
  if (want_dt && !device_dt) {
    *dt_host = DBL_MAX;
    CK(artemis_rt_memcpy_h2d(dt_dev.p, dt_host, sizeof(double), stream), "h2d");
  }
  for (int stage = 1; stage <= nstages; ++stage) {
    const bool last = (stage == nstages);
    int out;
    if (last && cur != A) out = A; // aliases prim_u1: cell-wise access only
    else out = (cur + 1) % 3 == A ? (cur + 2) % 3 : (cur + 1) % 3;
    const artemis_pack_t p = make_pack(cur);
    artemis_stage_args_t a;
    std::memset(&a, 0, sizeof a);
    a.gam0 = gam0[stage - 1], a.gam1 = gam1[stage - 1];
    a.beta_dt = beta[stage - 1] * dt; // artemis_integrator.hpp:66
    a.bdt = beta[stage - 1] * dt;     // artemis_driver.cpp:168
    a.pcm = (stage == 1 && integrator == "vl2"); // artemis_driver.cpp:182
    a.prim_in = gprim[cur].tab(), a.prim_u1 = gprim[A].tab(), a.prim_out = gprim[out].tab();
    a.cons_out = ((dropin == 1 && last) || dropin == 2) ? gu0.tab() : nullptr;
    a.redo_scratch = redo_scratch.p; // (this state's own detect-and-redo lists: several states may be in flight on one thread)
    a.cfl = cfl_gas;
    a.dt_dev = (last && want_dt) ? (device_dt ? tstate.p + 2 : dt_dev.p) : nullptr;
    if (device_dt) a.beta_dt_dev = tstate.p + 3 + (stage - 1); // beta*dt stays on the device
    bool any_remote = false;
    int faces = 0; // faces through which some block of the pack feeds a neighbour
    for (auto &L : links) any_remote = any_remote || remote(*L), faces |= (1 << L->face);
    {
      bool copy_bcs = true; // conditions that only copy zones this driver wrote (a user condition computes new values)
      for (int f = 0; f < 6; ++f)
        copy_bcs = copy_bcs && (mesh_bc[f] == ARTEMIS_BC_PERIODIC || mesh_bc[f] == ARTEMIS_BC_OUTFLOW ||
                                mesh_bc[f] == ARTEMIS_BC_REFLECT || mesh_bc[f] == ARTEMIS_BC_NONE);
      if (!any_remote && !loopback && copy_bcs && nstages >= 2 && !artemis::opt(artemis::OPT_NO_TINY_HINT)) {
        // word q describes the state before stage q + 1 of a step (the same pointers every step: a captured
        // graph of one step stays valid); the state after the last stage is the next step's word 0
        unsigned *w = reinterpret_cast<unsigned *>(tiny_words.p);
        a.tiny_in = tiny_valid ? w + (stage - 1) : nullptr;
        a.tiny_out = w + (stage % nstages), a.tiny_clear = w + (stage - 1);
        tiny_valid = true;
      } else {
        tiny_valid = false;
      }
    }
    a.shell_faces = faces;
    {
      // Faces that are `outflow` on every block of this rank (x1 always is on the Sedov deck: the decomposition never
      // cuts x1), with nobody else reading the state inside the loop (no drop-in accounting): the kernel stages the edge
      // zone instead of the ghost zones behind them.  If that covers every physical face of the rank the per-stage
      // boundary fill is skipped; otherwise only the x1 ghost columns are left alone (both x1 faces covered) and the
      // kernel keeps reading the x2 / x3 ghosts the fill provides.  evolve() completes the ghost zones before it returns.
      int mask = 0;
      bool all_covered = true; // every physical face of every block is an outflow face of the mask
      if (dropin == 0) {
        for (int f = 0; f < 2 * ndim; ++f) {
          bool every = true;
          for (int b = 0; b < nb && every; ++b) every = bc_flat[6 * b + f] == ARTEMIS_BC_OUTFLOW;
          if (every) mask |= 1 << f;
        }
      }
      for (int b = 0; b < nb; ++b)
        for (int f = 0; f < 2 * ndim; ++f)
          all_covered = all_covered && (bc_flat[6 * b + f] == ARTEMIS_BC_NONE || ((mask >> f) & 1));
      // Several blocks per rank (stacked blocks: the face between two of them is a neighbour's, the outer ones are physical):
      // the same rule block by block -- every physical face of every block is `outflow` -- with one mask per block
      outflow_by_block.clear();
      if (dropin == 0 && !all_covered && nb > 1 && nb <= 10) {
        bool every_physical_is_outflow = true;
        int common = 63;
        for (int b = 0; b < nb; ++b) {
          int mb = 0;
          for (int f = 0; f < 2 * ndim; ++f) {
            if (bc_flat[6 * b + f] == ARTEMIS_BC_OUTFLOW) mb |= 1 << f;
            else every_physical_is_outflow = every_physical_is_outflow && bc_flat[6 * b + f] == ARTEMIS_BC_NONE;
          }
          outflow_by_block.push_back(static_cast<unsigned char>(mb));
          common &= mb;
        }
        if (every_physical_is_outflow) mask = common | 64, all_covered = true; // (bit 6: "by block"; stripped below)
        else outflow_by_block.clear();
      }
      const bool by_block = (mask & 64) != 0;
      mask &= 63;
      a.outflow_faces = mask;
      a.outflow_faces_by_block = by_block ? outflow_by_block.data() : nullptr;
      x1_done_hint = ((mask & 3) == 3) ? 3 : 0;
      skip_bc_hint = (mask != 0 || by_block) && all_covered;
      if (!skip_bc_hint && x1_done_hint == 0) a.outflow_faces = mask = 0; // (nothing to gain: keep the plain protocol)
      if (!skip_bc_hint) a.outflow_faces = mask & 3;                       // (the boundary fill still provides x2 / x3 ghosts)
      ghosts_stale = ghosts_stale || a.outflow_faces != 0 || a.outflow_faces_by_block != nullptr;
    }
    // (ARTEMIS_FORCE_OVERLAP=1: diagnostic, shell-first ordering even when every link is local)
    const bool force_ovl = artemis::opt(artemis::OPT_FORCE_OVERLAP) != 0;
    const bool ovl = overlap && (any_remote || (force_ovl && !links.empty()));
    void *e0 = nullptr, *e1 = nullptr;
    if (time_kernels && kev.size() < kMaxTimedLaunches) { // (bounded: long runs with timing on must not leak events)
      e0 = artemis_rt_event_create(), e1 = artemis_rt_event_create();
      CK(artemis_rt_event_record(e0, stream), "event");
    }
    if (!ovl) {
      a.region = 0;
      CK(artemis_hip_stage_fused(&p, &a, stream), "stage_fused");
      if (e0) {
        CK(artemis_rt_event_record(e1, stream), "event");
        kev.emplace_back(e0, e1);
      }
      fill_ghosts(out);
    } else if (overlap == 2) {
      // ONE launch, boundary-shell workgroups first; they count themselves into a device counter
      // that a one-wave kernel on the comm stream waits for, so the slabs are packed and sent
      // while the bulk of the same launch is still running.
      unsigned *counter = reinterpret_cast<unsigned *>(signal.p);
      unsigned target = 0;
      shell_wait_used = true;
      CK(artemis_rt_memset(counter, 0, sizeof(unsigned), stream), "memset"); // (the timeout flag stays sticky)
      CK(artemis_rt_event_record(ev0, stream), "event");
      a.region = 0, a.shell_done = counter, a.shell_target = &target;
      CK(artemis_hip_stage_fused(&p, &a, stream), "stage_fused");
      if (e0) {
        CK(artemis_rt_event_record(e1, stream), "event");
        kev.emplace_back(e0, e1);
      }
      CK(artemis_rt_stream_wait_event(comm_stream, ev0), "wait");
      // (ARTEMIS_TEST_SHELL_TARGET_BUMP: test hook, waits for more workgroups than exist -> the timeout path)
      target += static_cast<unsigned>(artemis::opt(artemis::OPT_TEST_SHELL_TARGET_BUMP));
      CK(artemis_hip_wait_counter(counter, target, counter + 1, comm_stream), "wait_counter");
      // shell zones the kernel deferred to the exact path (next to vanishing velocities) are final before they are packed
      CK(artemis_hip_stage_fused_redo_shell(&p, &a, comm_stream), "stage_fused (shell redo)");
      fill_ghosts_start(out, comm_stream);
      fill_ghosts_finish(out, comm_stream);
    } else {
      // boundary shell first; its slabs travel on the comm stream while the bulk is computed
      a.region = 1;
      unsigned *const clear_word = a.tiny_clear;
      a.tiny_clear = nullptr; // (include/artemis_hip.h: with region 1 / 2 launches only the LAST launch of the stage clears
                              //  the hint word -- the bulk launch still has to read it)
      CK(artemis_hip_stage_fused(&p, &a, stream), "stage_fused shell");
      CK(artemis_rt_event_record(ev0, stream), "event");
      CK(artemis_rt_stream_wait_event(comm_stream, ev0), "wait");
      fill_ghosts_start(out, comm_stream);
      a.region = 2;
      a.tiny_clear = clear_word;
      CK(artemis_hip_stage_fused(&p, &a, stream), "stage_fused bulk");
      if (e0) {
        CK(artemis_rt_event_record(e1, stream), "event");
        kev.emplace_back(e0, e1);
      }
      fill_ghosts_finish(out, comm_stream);
    }
    if (dropin == 1) { // FillDerived = PrimToCons over the entire block (artemis.cpp:123, artemis_driver.cpp:261)
      const artemis_pack_t po = make_pack(out);
      CK(artemis_hip_prim_to_cons(&po, stream), "PrimToCons");
    } else if (dropin == 2) { // the kernel stored cons of the zones it updated: only the ghost zones are left
      const artemis_pack_t po = make_pack(out);
      CK(artemis_hip_prim_to_cons_ghosts(&po, stream), "PrimToCons (ghost zones)");
    }
    cur = out;
  }
  base = cur;
  cons_valid = dropin != 0;
  x1_done_hint = 0, skip_bc_hint = false;
}
// the ghost zones the stage loop left alone (see skip_bc_hint / x1_done_hint): the plain boundary fill of the current state
void artemis_sim_impl::fill_stale_ghosts() {
  if (!ghosts_stale) return;
  const artemis_pack_t p = make_pack(base);
  artemis_bc_params_t bp = bcpar;
  bp.floor_ghosts = 0, bp.x1_interior_done = 0;
  CK(artemis_hip_apply_bc(&p, bc_flat.data(), &bp, stream), "apply_bc (ghost zones the stage loop left alone)");
  ghosts_stale = false;
}

// One step on a refined mesh with the one-kernel stages: every block through artemis_hip_stage_fused /
// artemis_hip_stage_general as on a uniform mesh (no flux arrays), then the flux correction of
// artemis_driver.cpp:196-202 as a fix-up -- the fine side's faces on the coarse-fine boundaries solved on their own,
// restricted (and sent between ranks) by the same artemis_hip_ml_flux_correction calls as the per-task chain, and the
// coarse zones that touch such a face redone with them.  Same bits as step_unfused (tests/test_multilevel.py runs both).
void artemis_sim_impl::step_ml_fused() {
  tiny_valid = false;
  // (the fine-side faces and their restrictions live in the flux arrays; no u1: 160 B per zone less.  Diffusion fluxes
  //  written for the whole pack -- heat conduction, or a pack the viscous-source march does not cover -- need every row)
  ensure_ml_flux_arrays();
  for (int q = 1; q < 3; ++q) {
    if (!gprim[q].ok()) gprim[q].alloc(nb, 6 * ns_gas, N);
    if (!dprim[q].ok()) dprim[q].alloc(nb, 4 * ns_dust, N);
  }
  const int A = base;
  int cur = A;
  for (int stage = 1; stage <= nstages; ++stage) {
    // the fix-up reads the stage's input AND the start-of-step state after the stage kernel has written its output:
    // the output never aliases either (three buffers; the last stage's output becomes the new base)
    const int out = (cur == A) ? (A + 1) % 3 : 3 - cur - A;
    const artemis_pack_t p = make_pack(cur);
    artemis_stage_general_args_t a;
    std::memset(&a, 0, sizeof a);
    a.gam0 = gam0[stage - 1], a.gam1 = gam1[stage - 1];
    a.beta_dt = beta[stage - 1] * dt, a.bdt = beta[stage - 1] * dt;
    a.pcm = (stage == 1 && integrator == "vl2");
    a.time = time;
    a.gas_in = gprim[cur].tab(), a.gas_u1 = gprim[A].tab(), a.gas_out = gprim[out].tab();
    a.dust_in = dprim[cur].tab(), a.dust_u1 = dprim[A].tab(), a.dust_out = dprim[out].tab();
    place_binary();
    a.gravity = do_gravity ? &grav : nullptr;
    a.rf_omega = do_rframe ? rf_omega : 0.0, a.rf_qshear = rf_qshear;
    a.cfl_gas = cfl_gas, a.cfl_dust = cfl_dust;
    a.dt_dev = nullptr; // the timestep is estimated after the fix-up (new_dt_unfused)
    if (do_gravity && grav_nbody) nbody_stage_args(p, a, a.bdt);
    if (do_drag) { // DragSource couples the fluids: it runs once, over every zone, after the fix-up
      drag.damp_visc = damp_to_visc ? &diff.visc : nullptr;
      // one gas species coupled by simple_dust drag: the stage finishes every zone itself (inside the dust march where
      // it runs) and only the fix-up's listed zones are finished again afterwards; other laws: every zone after the fix-up
      a.drag = &drag, a.defer_finish = (drag.type == ARTEMIS_DRAG_SIMPLE_DUST && ns_gas == 1 && ns_dust >= 1) ? 2 : 1;
    }
    const bool diffuse = do_gas && (do_viscosity || do_conduction);
    // viscosity alone: the five sums of artemis_hip_viscous_source for the whole pack, and the viscous fluxes themselves
    // only on the faces the flux correction touches (artemis_hip_ml_viscous_faces, below)
    const bool visc_source = diffuse && do_viscosity && !do_conduction && ns_gas == 1 && !artemis::opt(artemis::OPT_NO_VISC_SOURCE) &&
                             artemis_hip_viscous_source_covers(&p) != 0;
    if (visc_source && !gdsum.ok()) gdsum.alloc(nb, 5, N);
    if (diffuse) { // artemis_driver.cpp:189-194 on the stage's input primitives
      if (visc_source) {
        CK(artemis_hip_viscous_source(&p, &diff, a.bdt, nullptr, gdsum.tab(), stream), "viscous source");
        a.diffusion_sums = gdsum.tab();
      } else {
        if (do_viscosity) CK(artemis_hip_zero_viscous_flux(&p, &diff, stream), "Gas::ZeroDiffusionFlux + ViscousFlux");
        else CK(artemis_hip_zero_diffusion_flux(&p, stream), "Gas::ZeroDiffusionFlux");
        if (do_conduction) CK(artemis_hip_thermal_flux(&p, &diff, stream), "Gas::ThermalFlux");
      }
      a.diffusion = &diff;
    }
    if (do_cooling && do_gas) a.cooling = &cool;
    if (ml_tuned) {
      artemis_stage_args_t t;
      std::memset(&t, 0, sizeof t);
      t.gam0 = a.gam0, t.gam1 = a.gam1, t.beta_dt = a.beta_dt, t.bdt = a.bdt, t.pcm = a.pcm;
      t.prim_in = a.gas_in, t.prim_u1 = a.gas_u1, t.prim_out = a.gas_out;
      t.cfl = cfl_gas;
      ensure_redo_scratch();
      t.redo_scratch = redo_scratch.p;
      CK(artemis_hip_stage_fused(&p, &t, stream), "stage_fused");
    } else {
      general_variant = artemis_hip_stage_general_variant(&p, &a);
      CK(artemis_hip_stage_general(&p, &a, stream), "stage_general");
    }
    CK(artemis_hip_ml_face_fluxes(&p, &a, static_cast<const artemis_ml_face_box_t *>(ml.fine_boxes.p), ml.fine_boxes.n, stream),
       "fine-side faces of the coarse-fine boundaries");
    if (visc_source) {
      CK(artemis_hip_ml_viscous_faces(&p, &diff, static_cast<const artemis_ml_face_box_t *>(ml.fine_boxes.p), ml.fine_boxes.n,
                                      static_cast<const artemis_ml_fix_cell_t *>(ml.fix_cells.p), ml.fix_cells.n, stream),
         "viscous fluxes of the faces the flux correction touches");
      a.diffusion_sums = nullptr; // the fix-up forms the listed zones' sums from the corrected arrays
    }
    flux_correction_multilevel(p);
    CK(artemis_hip_ml_stage_fixup(&p, &a, static_cast<const artemis_ml_fix_cell_t *>(ml.fix_cells.p), ml.fix_cells.n, stream),
       "coarse zones on coarse-fine faces");
    if (a.defer_finish == 2) {
      const artemis_pack_t po = make_pack(out);
      CK(artemis_hip_stage_finish_cells(&po, &drag, time, a.bdt, static_cast<const artemis_ml_fix_cell_t *>(ml.fix_cells.p),
                                        ml.fix_cells.n, stream),
         "DragSource + SetAuxillaryFields + ConsToPrim of the fix-up zones");
    } else if (a.defer_finish) {
      const artemis_pack_t po = make_pack(out);
      CK(artemis_hip_stage_finish(&po, &drag, time, a.bdt, stream), "DragSource + SetAuxillaryFields + ConsToPrim");
    }
    fill_ghosts(out);
    cur = out;
  }
  base = cur;
  cons_valid = false;
}

// One step on the per-task path (artemis_driver.cpp:157-261 literally).
void artemis_sim_impl::step_unfused() {
  tiny_valid = false; // (another producer of the primitives)
  ensure_unfused();
  materialise_cons();
  const artemis_pack_t p = make_pack(base);
  CK(artemis_hip_deep_copy_conserved(&p, stream), "DeepCopyConservedData");
  for (int stage = 1; stage <= nstages; ++stage) {
    const Real bdt = beta[stage - 1] * dt;
    const int do_pcm = (stage == 1 && integrator == "vl2");
    if (do_gas) CK(artemis_hip_calculate_fluxes(&p, ARTEMIS_GAS, do_pcm, stream), "Gas::CalculateFluxes");
    if (do_dust) CK(artemis_hip_calculate_fluxes(&p, ARTEMIS_DUST, do_pcm, stream), "Dust::CalculateFluxes");
    if (do_viscosity || do_conduction) { // artemis_driver.cpp:189-194
      if (do_viscosity) CK(artemis_hip_zero_viscous_flux(&p, &diff, stream), "Gas::ZeroDiffusionFlux + ViscousFlux");
      else CK(artemis_hip_zero_diffusion_flux(&p, stream), "Gas::ZeroDiffusionFlux");
      if (do_conduction) CK(artemis_hip_thermal_flux(&p, &diff, stream), "Gas::ThermalFlux");
    }
    if (multilevel) flux_correction_multilevel(p); // artemis_driver.cpp:196-202
    place_binary();
    if (!do_drag && !grav_nbody && !artemis::opt(artemis::OPT_NO_EPILOGUE)) {
      // everything between the flux tasks and the boundary exchange is cell-local: one pass over the
      // stored fluxes (ApplyUpdate ... ConsToPrim, artemis_driver.cpp:205-255) instead of eight
      artemis_stage_general_args_t a;
      std::memset(&a, 0, sizeof a);
      a.gam0 = gam0[stage - 1], a.gam1 = gam1[stage - 1], a.beta_dt = beta[stage - 1] * dt, a.bdt = bdt;
      a.time = time;
      a.gravity = do_gravity ? &grav : nullptr;
      a.rf_omega = do_rframe ? rf_omega : 0.0, a.rf_qshear = rf_qshear;
      a.diffusion = (do_gas && (do_viscosity || do_conduction)) ? &diff : nullptr;
      a.cooling = (do_cooling && do_gas) ? &cool : nullptr;
      CK(artemis_hip_stage_epilogue(&p, &a, stream), "stage epilogue");
    } else if (!(do_cooling && do_gas) && !artemis::opt(artemis::OPT_NO_EPILOGUE)) {
      // drag and / or N-body gravity: the same pass up to the sources that may run before them, the state left
      // conserved; then NBodyGravity and RotatingFrameForce as tasks (artemis_driver.cpp:222-236 order); then
      // DragSource + SetAuxillaryFields + ConsToPrim in one pass -- three to five launches instead of ten
      artemis_stage_general_args_t a;
      std::memset(&a, 0, sizeof a);
      a.gam0 = gam0[stage - 1], a.gam1 = gam1[stage - 1], a.beta_dt = beta[stage - 1] * dt, a.bdt = bdt;
      a.time = time;
      a.diffusion = (do_gas && (do_viscosity || do_conduction)) ? &diff : nullptr;
      if (!grav_nbody) {
        a.gravity = do_gravity ? &grav : nullptr;
        a.rf_omega = do_rframe ? rf_omega : 0.0, a.rf_qshear = rf_qshear;
      }
      CK(artemis_hip_stage_epilogue_cons(&p, &a, stream), "stage epilogue (conserved)");
      if (grav_nbody) {
        if (do_gravity) { // gravity.cpp:150-155
          const Real omf = (do_rframe && nbody_frame_correction) ? rf_omega : 0.0;
          if (time >= grav.tstart && time < grav.tstop)
            CK(artemis_hip_nbody_gravity(&p, particles.data(), static_cast<int>(particles.size()), omf, time, bdt,
                                         particle_force.data(), stream), "NBodyGravity");
        }
        if (do_rframe) CK(artemis_hip_rotating_frame_force(&p, rf_omega, rf_qshear, time, bdt, stream), "RotatingFrameForce");
      }
      drag.damp_visc = damp_to_visc ? &diff.visc : nullptr;
      CK(artemis_hip_stage_finish(&p, do_drag ? &drag : nullptr, time, bdt, stream), "DragSource + SetAuxillaryFields + ConsToPrim");
    } else {
      CK(artemis_hip_apply_update(&p, gam0[stage - 1], gam1[stage - 1], beta[stage - 1] * dt, stream), "ApplyUpdate");
      if (do_gas) CK(artemis_hip_flux_source(&p, ARTEMIS_GAS, bdt, stream), "Gas::FluxSource");
      if (do_dust) CK(artemis_hip_flux_source(&p, ARTEMIS_DUST, bdt, stream), "Dust::FluxSource");
      if (do_viscosity || do_conduction) // artemis_driver.cpp:218-221
        CK(artemis_hip_diffusion_update(&p, &diff, bdt, stream), "Gas::DiffusionUpdate");
      // artemis_driver.cpp:222-241: gravity, rotating frame, drag, in this order, with the time at
      // the start of the step (:167)
      if (do_gravity && grav_nbody) { // gravity.cpp:150-155
        const Real omf = (do_rframe && nbody_frame_correction) ? rf_omega : 0.0;
        if (time >= grav.tstart && time < grav.tstop)
          CK(artemis_hip_nbody_gravity(&p, particles.data(), static_cast<int>(particles.size()), omf, time, bdt,
                                       particle_force.data(), stream), "NBodyGravity");
      } else if (do_gravity) {
        CK(artemis_hip_external_gravity(&p, &grav, time, bdt, stream), "ExternalGravity");
      }
      if (do_rframe) CK(artemis_hip_rotating_frame_force(&p, rf_omega, rf_qshear, time, bdt, stream), "RotatingFrameForce");
      if (do_drag) {
        drag.damp_visc = damp_to_visc ? &diff.visc : nullptr;
        CK(artemis_hip_drag_source(&p, &drag, time, bdt, stream), "DragSource");
      }
      if (do_cooling && do_gas) CK(artemis_hip_cooling_source(&p, &cool, time, bdt, stream), "CoolingSource"); // :243-248
      CK(artemis_hip_set_aux(&p, stream), "SetAuxillaryFields");
      CK(artemis_hip_cons_to_prim(&p, stream), "ConsToPrim");
    }
    fill_ghosts(base);
    CK(artemis_hip_prim_to_cons(&p, stream), "PrimToCons");
  }
  cons_valid = true;
}

// parthenon EvolutionDriver::Execute + SetGlobalTimeStep (upstream, recalled; pinned by
// tst/scripts/advection/advection.py:100-118).
long artemis_sim_impl::evolve(long max_cycles) {
  auto global_min = [&](Real v) {
    if (has_comm && nranks > 1 && comm.allreduce_min(comm.ctx, &v)) throw std::runtime_error("allreduce failed");
    return v;
  };
  if (ncycle == 0 && time == 0.0 && dt == DBL_MAX) {
    dt = global_min(new_dt_unfused());
    if (tlim > 0.0 && time < tlim && (tlim - time) < dt) dt = tlim - time;
  }
  for (auto &pr : kev) artemis_rt_event_destroy(pr.first), artemis_rt_event_destroy(pr.second);
  kev.clear();
  CK(artemis_rt_memset(signal.p, 0, 2 * sizeof(unsigned), stream), "memset"); // counter + sticky timeout flag
  shell_wait_used = false;
  CK(artemis_rt_device_sync(), "sync");
  const auto t0 = std::chrono::steady_clock::now();
  long n = 0;
  // The one-kernel stages keep {time, dt, dt_est} on the device (kernels read dt there, artemis_hip_advance_dt applies
  // SetGlobalTimeStep's rules): the host only decides when to stop, and synchronises as rarely as that allows.
  const bool multi = has_comm && (nranks > 1 || loopback);
  // (a gravity time window is evaluated against the host's clock, which the device loop does not keep)
  // (so is the orbit of a binary)
  const bool grav_window = do_gravity && (grav.tstart > -DBL_MAX || grav.tstop < DBL_MAX || grav.type == ARTEMIS_GRAVITY_BINARY);
  const bool async_ok = use_fused && !grav_window && (!multi || comm.allreduce_min_dev) && !artemis::opt(artemis::OPT_SYNC_LOOP);
  void *gexec[3] = {nullptr, nullptr, nullptr};
  int gnext[3] = {0, 0, 0};
  long plain_steps = 0; // (the first few steps of an evolve() run plainly: they allocate the ping-pong buffers and scratch)
  // One time step is a fixed launch sequence when dt lives on the device: capture it into a hipGraph after a few plain
  // steps and replay it, one graph per ping-pong phase.  Single rank, no overlap streams, no per-kernel timing.
  bool graph_ok = !multi && !time_kernels && overlap == 0 && !artemis::opt(artemis::OPT_NO_GRAPH);
  auto cycles_left = [&]() {
    long todo = -1;
    if (nlim >= 0) todo = nlim - ncycle;
    if (max_cycles >= 0) todo = (todo < 0) ? max_cycles - n : std::min(todo, max_cycles - n);
    return todo;
  };
  // `todo` cycles with {time, dt, dt_est} on the device and no synchronisation inside
  auto run_async = [&](long todo) {
    double h[6] = {time, dt, DBL_MAX, 0.0, 0.0, 0.0};
    for (int q = 0; q < nstages; ++q) h[3 + q] = beta[q] * dt;
    CK(artemis_rt_memcpy_h2d(tstate.p, h, sizeof h, stream), "h2d");
    CK(artemis_rt_stream_sync(stream), "sync");
    for (long m = 0; m < todo; ++m, ++n, ++plain_steps) {
      const int key = base;
      if (graph_ok && plain_steps >= 3 && gexec[key]) {
        base = gnext[key], cons_valid = false;
        CK(artemis_rt_graph_launch(gexec[key], stream), "graph launch");
        ncycle++;
        continue;
      }
      const bool capture = graph_ok && plain_steps >= 3 && artemis_rt_capture_begin(stream) == 0;
      if (graph_ok && plain_steps >= 3 && !capture) graph_ok = false; // (the CPU stand-in has no graphs)
      step_fused(true, true);
      if (multi && comm.allreduce_min_dev(comm.ctx, tstate.p + 2, stream))
        throw std::runtime_error("allreduce_min_dev failed");
      CK(artemis_hip_advance_dt(tstate.p, tlim, nstages, beta, stream), "advance_dt");
      if (capture) { // nothing has run yet: instantiate and launch what was recorded
        gexec[key] = artemis_rt_capture_end(stream);
        if (!gexec[key]) throw HipFail(std::string("graph capture failed: ") + artemis_hip_last_error());
        gnext[key] = base;
        CK(artemis_rt_graph_launch(gexec[key], stream), "graph launch");
      }
      ncycle++;
    }
    CK(artemis_rt_memcpy_d2h(h, tstate.p, sizeof h, stream), "d2h");
    CK(artemis_rt_stream_sync(stream), "sync");
    time = h[0], dt = h[1];
    if (!std::isfinite(dt) || !(dt > 0.0) || !std::isfinite(time))
      throw std::runtime_error("the device-side time loop produced a non-finite or non-positive dt");
  };
  if (async_ok && tlim < 0.0) {
    // Without a time limit nothing on the host depends on dt: never synchronise inside the loop, so launches queue
    // ahead of the GPU.
    const long todo = cycles_left();
    if (todo < 0) throw std::runtime_error("no tlim, no nlim and no cycle budget: nothing bounds the run");
    run_async(todo);
  }
  while (!(async_ok && tlim < 0.0) && (tlim < 0.0 || time < tlim) && (nlim < 0 || ncycle < nlim) &&
         (max_cycles < 0 || n < max_cycles)) {
    if (async_ok) {
      // With a time limit the host decides one thing only: when to stop.  SetGlobalTimeStep at most doubles dt, so k
      // cycles from here end no later than time + dt (2^k - 1): while that stays below tlim the loop condition holds
      // for every one of them whatever the estimates turn out to be, and they run as above (advance_dt applies the
      // same rules, the tlim clamp included).  Close to tlim the cycles run one by one below.
      long k = 0;
      const double room = (tlim - time) / dt;
      while (k < 40 && std::ldexp(1.0, static_cast<int>(k)) * 1.001 < room) ++k; // time + dt 2^(k-1) < tlim, with a margin
      const long todo = cycles_left();
      if (todo >= 0) k = std::min(k, todo);
      if (k >= 2) {
        run_async(k);
        continue;
      }
    }
    Real est;
    if (use_fused) {
      step_fused(true, false);
      const bool dev_reduce = has_comm && (nranks > 1 || loopback) && comm.allreduce_min_dev;
      if (dev_reduce && comm.allreduce_min_dev(comm.ctx, dt_dev.p, stream))
        throw std::runtime_error("allreduce_min_dev failed");
      CK(artemis_rt_memcpy_d2h(dt_host, dt_dev.p, sizeof(double), stream), "d2h");
      CK(artemis_rt_stream_sync(stream), "sync");
      est = dev_reduce ? *dt_host : global_min(*dt_host);
    } else {
      if (ml_fused) step_ml_fused();
      else step_unfused();
      est = global_min(new_dt_unfused());
    }
    time += dt;
    ncycle++, n++;
    Real ndt = dt;
    if (ndt < 0.1 * DBL_MAX) ndt *= 2.0;
    ndt = std::min(ndt, est);
    if (tlim > 0.0 && time < tlim && (tlim - time) < ndt) ndt = tlim - time;
    dt = ndt;
  }
  for (void *g : gexec) artemis_rt_graph_destroy(g);
  fill_stale_ghosts(); // (inside the timed region: part of the work)
  CK(artemis_rt_device_sync(), "sync");
  last_wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (shell_wait_used) {
    // The comm stream's wait kernel gives up after its spin limit instead of hanging the GPU (the two
    // streams did not run concurrently, or the counter never reached its target).  The slabs packed
    // behind it may then hold unfinished shell data: the run is invalid, say so and stop overlapping.
    unsigned sig[2] = {0u, 0u};
    CK(artemis_rt_memcpy_d2h(sig, signal.p, sizeof sig, stream), "d2h");
    CK(artemis_rt_stream_sync(stream), "sync");
    if (sig[1] != 0u) {
      overlap = 0;
      throw HipFail("overlap mode 2: the comm stream timed out waiting for the boundary-shell workgroups "
                    "(stale ghost zones may have been exchanged); falling back to overlap = 0 -- re-run from a valid state");
    }
  }
  kernel_ms_sum = 0.0, kernel_launches = 0;
  for (auto &pr : kev) {
    const double ms = artemis_rt_event_elapsed_ms(pr.first, pr.second);
    if (ms >= 0.0) kernel_ms_sum += ms, kernel_launches++;
  }
  return n;
}

// utils/history.hpp:29-100 (volume integrals of the conserved fields)
int artemis_sim_impl::history(double *out) {
  materialise_cons();
  const int nout = 6 + 4 * ns_dust;
  for (int q = 0; q < nout; ++q) out[q] = 0.0;
  for (int b = 0; b < nb; ++b) {
    std::vector<Real> g, d;
    if (do_gas) g = download(gu0, b);
    if (do_dust) d = download(du0, b);
    for (int k = ks; k <= ke; ++k)
      for (int j = js; j <= je; ++j)
        for (int i = is; i <= ie; ++i) {
          const Real vv = cell_coords(b, k, j, i).volume(); // history.hpp:49-50
          const size_t c = (static_cast<size_t>(k) * nj + j) * ni + i;
          if (do_gas) {
            out[0] += g[0 * N + c] * vv;
            for (int q = 0; q < 3; ++q) out[1 + q] += g[(ns_gas + q) * N + c] * vv;
            out[4] += g[(4 * ns_gas) * N + c] * vv;
            out[5] += g[(5 * ns_gas) * N + c] * vv;
          }
          for (int n = 0; n < ns_dust; ++n) {
            out[6 + 4 * n] += d[n * N + c] * vv;
            for (int q = 0; q < 3; ++q) out[6 + 4 * n + 1 + q] += d[(ns_dust + 3 * n + q) * N + c] * vv;
          }
        }
  }
  if (has_comm && nranks > 1 && comm.allreduce_sum(comm.ctx, out, nout)) throw std::runtime_error("allreduce failed");
  return nout;
}

// linear_wave.hpp:267-377 / advection.hpp:224-405 UserWorkAfterLoop
int artemis_sim_impl::errors(double *out) {
  if (pgen != PG_LINWAVE && pgen != PG_ADVECTION) return 0;
  materialise_cons();
  Real l1[13];
  for (int q = 0; q < 13; ++q) l1[q] = 0.0;
  for (int b = 0; b < nb; ++b) {
    std::vector<Real> g, d;
    if (do_gas) g = download(gu0, b);
    if (do_dust) d = download(du0, b);
    const Real dx[3] = {(blocks[b].xmax[0] - blocks[b].xmin[0]) / mbnx[0],
                        (blocks[b].xmax[1] - blocks[b].xmin[1]) / mbnx[1],
                        (blocks[b].xmax[2] - blocks[b].xmin[2]) / mbnx[2]};
    const Real f0[3] = {blocks[b].xmin[0] - is * dx[0], blocks[b].xmin[1] - js * dx[1],
                        blocks[b].xmin[2] - ks * dx[2]};
    for (int k = ks; k <= ke; ++k)
      for (int j = js; j <= je; ++j)
        for (int i = is; i <= ie; ++i) {
          const Real b1[2] = {f0[0] + i * dx[0], f0[0] + (i + 1) * dx[0]};
          const Real b2[2] = {f0[1] + j * dx[1], f0[1] + (j + 1) * dx[1]};
          const Real b3[2] = {f0[2] + k * dx[2], f0[2] + (k + 1) * dx[2]};
          const Real x1v = 0.5 * (b1[0] + b1[1]), x2v = 0.5 * (b2[0] + b2[1]), x3v = 0.5 * (b3[0] + b3[1]);
          const Real vol = (b1[1] - b1[0]) * (b2[1] - b2[0]) * (b3[1] - b3[0]);
          const Real x = lw.cos_a2 * (x1v * lw.cos_a3 + x2v * lw.sin_a3) + x3v * lw.sin_a2;
          const Real sn = std::sin(lw.k_par * x);
          const size_t c = (static_cast<size_t>(k) * nj + j) * ni + i;
          if (pgen == PG_LINWAVE) {
            const int wf = lw.wave_flag;
            const Real mx = lw.d0 * lw.vflow + lw.amp * sn * lw.rem[1][wf];
            const Real my = lw.amp * sn * lw.rem[2][wf];
            const Real mz = lw.amp * sn * lw.rem[3][wf];
            const Real ca = lw.d0 + lw.amp * sn * lw.rem[0][wf];
            const Real cm1 = mx * lw.cos_a2 * lw.cos_a3 - my * lw.sin_a3 - mz * lw.sin_a2 * lw.cos_a3;
            const Real cm2 = mx * lw.cos_a2 * lw.sin_a3 + my * lw.cos_a3 - mz * lw.sin_a2 * lw.sin_a3;
            const Real cm3 = mx * lw.sin_a2 + mz * lw.cos_a2;
            const Real ce = lw.p0 / lw.gm1 + 0.5 * lw.d0 * (lw.v1_0) * (lw.v1_0) + lw.amp * sn * lw.rem[4][wf];
            l1[0] += vol * std::abs(g[0 * N + c] - ca);
            l1[1] += vol * std::abs(g[(ns_gas + 0) * N + c] - cm1);
            l1[2] += vol * std::abs(g[(ns_gas + 1) * N + c] - cm2);
            l1[3] += vol * std::abs(g[(ns_gas + 2) * N + c] - cm3);
            l1[4] += vol * std::abs(g[(4 * ns_gas) * N + c] - ce);
          } else {
            const Real mx = lw.d0 * lw.vflow + lw.amp * sn * lw.v1_0;
            const Real cd = lw.d0 + lw.amp * sn;
            const Real cm1 = mx * lw.cos_a2 * lw.cos_a3, cm2 = mx * lw.cos_a2 * lw.sin_a3, cm3 = mx * lw.sin_a2;
            const Real ce = lw.p0 / lw.gm1 + 0.5 * lw.d0 * SQR(lw.v1_0) + 0.5 * lw.d0 * lw.amp * sn * SQR(lw.v1_0);
            if (do_gas) {
              l1[0] += vol * std::abs(g[0 * N + c] - cd);
              l1[1] += vol * std::abs(g[(ns_gas + 0) * N + c] - cm1);
              l1[2] += vol * std::abs(g[(ns_gas + 1) * N + c] - cm2);
              l1[3] += vol * std::abs(g[(ns_gas + 2) * N + c] - cm3);
              l1[4] += vol * std::abs(g[(4 * ns_gas) * N + c] - ce);
            }
            if (do_dust) {
              l1[5] += vol * std::abs(d[0 * N + c] - cd);
              l1[6] += vol * std::abs(d[(ns_dust + 0) * N + c] - cm1);
              l1[7] += vol * std::abs(d[(ns_dust + 1) * N + c] - cm2);
              l1[8] += vol * std::abs(d[(ns_dust + 2) * N + c] - cm3);
              l1[9] += vol * std::abs(d[1 * N + c] - cd);
              l1[10] += vol * std::abs(d[(ns_dust + 3) * N + c] + cm1);
              l1[11] += vol * std::abs(d[(ns_dust + 4) * N + c] + cm2);
              l1[12] += vol * std::abs(d[(ns_dust + 5) * N + c] + cm3);
            }
          }
        }
  }
  const int nv = (pgen == PG_LINWAVE) ? 5 : 13;
  if (has_comm && nranks > 1 && comm.allreduce_sum(comm.ctx, l1, nv)) throw std::runtime_error("allreduce failed");
  const Real vol = (xmax[0] - xmin[0]) * (xmax[1] - xmin[1]) * (xmax[2] - xmin[2]);
  for (int q = 0; q < nv; ++q) l1[q] = l1[q] / vol;
  if (pgen == PG_LINWAVE) {
    Real rms = 0.0;
    for (int q = 0; q < 5; ++q) rms += SQR(l1[q]), out[1 + q] = l1[q];
    out[0] = std::sqrt(rms);
    return 6;
  }
  Real rg = 0, r1 = 0, r2 = 0;
  for (int q = 0; q < 5; ++q) rg += SQR(l1[q]);
  for (int q = 5; q < 9; ++q) r1 += SQR(l1[q]);
  for (int q = 9; q < 13; ++q) r2 += SQR(l1[q]);
  out[0] = std::sqrt(rg), out[1] = std::sqrt(r1), out[2] = std::sqrt(r2);
  for (int q = 0; q < 13; ++q) out[3 + q] = l1[q];
  return 16;
}

// =======================================================================================
#define GUARD(expr, onerr)                                                                 \
  try {                                                                                    \
    expr;                                                                                  \
  } catch (const std::exception &e) {                                                      \
    g_sim_err = e.what();                                                                  \
    onerr;                                                                                 \
  }

// streams, events and pinned memory of a simulation state are raw handles: release them before the state goes
static void release_impl(std::unique_ptr<artemis_sim_impl> &p) {
  if (!p) return;
  for (auto &pr : p->kev) artemis_rt_event_destroy(pr.first), artemis_rt_event_destroy(pr.second);
  if (p->dt_host) artemis_rt_free_host(p->dt_host);
  if (p->ev0) artemis_rt_event_destroy(p->ev0);
  if (p->ev1) artemis_rt_event_destroy(p->ev1);
  if (p->stream) artemis_rt_stream_destroy(p->stream);
  if (p->comm_stream) artemis_rt_stream_destroy(p->comm_stream);
  p.reset();
}

// The C handle: owns the simulation state (rebuilt wholesale when an adaptive mesh changes, so the handle's address
// stays valid for the caller) and what is needed to rebuild it.
struct artemis_sim {
  std::unique_ptr<artemis_sim_impl> p;
  std::string deck;
  std::vector<std::string> overrides;
  bool has_comm = false;
  artemis_comm_t comm;
  // adaptive runs: consecutive cycles a leaf has asked to be derefined (parthenon's derefine_count rule, upstream)
  std::map<std::tuple<int, int, int, int>, int> deref_count;
  long remeshes = 0;
  // wall-clock seconds of the remeshes of the run (after the initial refinement): total, building the new state,
  // handing the data over; and how many there were
  double remesh_s = 0.0, remesh_build_s = 0.0, remesh_adopt_s = 0.0, remesh_tag_s = 0.0;
  long remesh_n = 0;
  // the last remesh of the run: leaves before / after, leaves created (not in the old mesh) / destroyed (not in the new
  // one), its wall-clock seconds (total, build, hand-over)
  long last_remesh[4] = {0, 0, 0, 0};
  double last_remesh_s[3] = {0.0, 0.0, 0.0};
  // A lean remesh releases the old state's work arrays BEFORE the new state allocates (peak = resident bytes); if the
  // new state then cannot be built or filled, the old one can no longer step: the handle is dead and says why
  std::string dead;
};

// A fresh state for the same deck on a given set of leaves (the problem generator runs on it: tables, `ic` states
// and -- during the initial refinement -- the initial condition itself at the new resolution).
static std::unique_ptr<artemis_sim_impl> build_state(const artemis_sim &h, const std::vector<artemis_host::Leaf> *leaves,
                                                     bool adopting = false) {
  std::unique_ptr<artemis_sim_impl> np(new artemis_sim_impl());
  if (leaves) np->forced_leaves = *leaves, np->have_forced = true;
  np->adopting = adopting;
  if (adopting && h.p) {
    np->reuse_from = h.p.get();
    for (int b = 0; b < h.p->nb; ++b) {
      const Block &B = h.p->blocks[b];
      np->reuse_lookup[std::make_tuple(B.level, B.lx[0], B.lx[1], B.lx[2])] = b;
    }
  }
  std::vector<const char *> ov;
  for (const std::string &o : h.overrides) ov.push_back(o.c_str());
  try {
    np->setup(h.deck.c_str(), static_cast<int>(ov.size()), ov.data(), h.has_comm ? &h.comm : nullptr);
  } catch (...) {
    release_impl(np);
    throw;
  }
  return np;
}

// The tree an adaptive mesh moves to: leaves tagged +1 split (up to numlevel - 1 levels above the root grid),
// sibling groups whose members have all asked for it `derefine_count` cycles in a row merge, and 2:1 balance over
// faces, edges and corners is restored by the tree itself (BlockTree::refine) -- which also vetoes a merge that a
// finer neighbour forbids.  Static regions stay refined.  Returns true if the set of leaves changed.
static bool next_leaves(artemis_sim &h, const std::vector<int> &tags, bool allow_derefine,
                        std::vector<artemis_host::Leaf> &out) {
  artemis_sim_impl &S = *h.p;
  typedef std::tuple<int, int, int, int> Key;
  const std::vector<artemis_host::Leaf> &old = S.tree_leaves;
  auto key = [](const artemis_host::Leaf &l) { return Key(l.level, l.lx[0], l.lx[1], l.lx[2]); };
  std::map<Key, int> tag_of;
  for (size_t q = 0; q < old.size(); ++q) tag_of[key(old[q])] = tags[q];
  // derefinement counters
  std::map<Key, int> cnt;
  for (const artemis_host::Leaf &l : old) {
    const Key k = key(l);
    cnt[k] = (tag_of[k] < 0) ? h.deref_count[k] + 1 : 0;
  }
  h.deref_count.swap(cnt);
  artemis_host::BlockTree t;
  t.ndim = S.ndim;
  for (int d = 0; d < 3; ++d) t.nrb[d] = S.nblk[d], t.periodic[d] = (d < S.ndim) && S.mesh_bc[2 * d] == ARTEMIS_BC_PERIODIC;
  const int nch = 1 << S.ndim;
  // the tree the step ran on: a leaf with a FINER neighbour (face, edge or corner) is not flagged for derefinement
  // this cycle, however long it has asked (upstream MeshRefinement::SetRefinement: the `nblevel > level` count) --
  // so a block loses at most one level per remesh and a coarse-fine-finer staircase unwinds from the top
  artemis_host::BlockTree told;
  told.ndim = S.ndim;
  for (int d = 0; d < 3; ++d) told.nrb[d] = t.nrb[d], told.periodic[d] = t.periodic[d];
  for (const artemis_host::Leaf &l : old) told.ensure(l.level, l.lx);
  auto finer_neighbour = [&](int level, const artemis_host::Loc &lx) {
    for (int o3 = (S.ndim > 2 ? -1 : 0); o3 <= (S.ndim > 2 ? 1 : 0); ++o3)
      for (int o2 = (S.ndim > 1 ? -1 : 0); o2 <= (S.ndim > 1 ? 1 : 0); ++o2)
        for (int o1 = -1; o1 <= 1; ++o1) {
          if (!o1 && !o2 && !o3) continue;
          artemis_host::Loc n = {lx[0] + o1, lx[1] + o2, lx[2] + o3};
          if (told.wrap(level, n) && told.is_internal(level, n)) return true;
        }
    return false;
  };
  auto merge_ok = [&](const artemis_host::Leaf &l) { // all siblings are leaves flagged for derefinement this cycle
    if (!allow_derefine || l.level == 0) return false;
    artemis_host::Loc par = t.parent(l.lx);
    for (int c = 0; c < nch; ++c) {
      artemis_host::Loc sib = {2 * par[0] + (c & 1), S.ndim > 1 ? 2 * par[1] + ((c >> 1) & 1) : 0, S.ndim > 2 ? 2 * par[2] + ((c >> 2) & 1) : 0};
      auto it = h.deref_count.find(Key(l.level, sib[0], sib[1], sib[2]));
      if (it == h.deref_count.end() || it->second < S.derefine_count) return false;
      if (finer_neighbour(l.level, sib)) return false;
    }
    return true;
  };
  for (const artemis_host::Leaf &l : old) {
    if (merge_ok(l)) t.ensure(l.level - 1, t.parent(l.lx)); // the parent becomes a leaf unless balance re-splits it
    else t.ensure(l.level, l.lx);
  }
  for (const artemis_host::Leaf &l : old)
    if (tag_of[key(l)] > 0 && l.level < S.amr_max_level) t.refine(l.level, l.lx);
  for (const artemis_sim_impl::Region &r : S.regions) t.add_region(r.level, r.lo, r.hi, S.xmin, S.xmax);
  out = t.leaves();
  if (out.size() != old.size()) return true;
  for (size_t q = 0; q < out.size(); ++q)
    if (key(out[q]) != key(old[q])) return true;
  return false;
}

// One remesh check (parthenon LoadBalancingAndAdaptiveMeshRefinement, upstream, after every cycle).  initial: the
// loop of Mesh::Initialize -- refine only, and the problem generator fills the new mesh instead of a prolongation.
static bool remesh(artemis_sim &h, bool initial, long force_refine_gid = -1, const std::vector<long> *inject = nullptr) {
  if (!h.p->adaptive || !h.p->refine_field) return false;
  const auto t_start = std::chrono::steady_clock::now();
  auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  // tags of this rank's blocks, gathered into the global (Z-ordered) leaf list: every rank then takes the same
  // decision (an all-reduce(sum) of a vector that is zero outside the rank's own slots)
  const std::vector<int> local = h.p->amr_tags();
  std::vector<int> tags(h.p->tree_leaves.size(), 0);
  {
    std::vector<double> g(tags.size(), 0.0);
    for (int b = 0; b < h.p->nb; ++b) g[h.p->blocks[b].gid] = local[b];
    if (h.p->has_comm && h.p->nranks > 1 && h.p->comm.allreduce_sum(h.p->comm.ctx, g.data(), static_cast<int>(g.size())))
      throw std::runtime_error("allreduce of the refinement tags failed");
    for (size_t q = 0; q < g.size(); ++q) tags[q] = static_cast<int>(g[q]);
  }
  if (force_refine_gid >= 0) { // artemis_sim_force_refine: this leaf splits whatever the criterion says, nothing merges
    for (int &t : tags) t = std::max(t, 0);
    if (force_refine_gid < static_cast<long>(tags.size())) tags[force_refine_gid] = 1;
  }
  if (inject) // artemis_sim_inject_refine_tags: these leaves are tagged +1 NEXT TO what the criterion says (merges go on)
    for (long g : *inject)
      if (g >= 0 && g < static_cast<long>(tags.size())) tags[g] = 1;
  std::vector<artemis_host::Leaf> leaves;
  const bool changed = next_leaves(h, tags, !initial, leaves);
  if (!initial) h.remesh_tag_s += since(t_start);
  if (!changed) return false;
  { // what this remesh changes, in leaves
    std::set<std::tuple<int, int, int, int>> was, is;
    for (const artemis_host::Leaf &l : h.p->tree_leaves) was.insert(std::make_tuple(l.level, l.lx[0], l.lx[1], l.lx[2]));
    for (const artemis_host::Leaf &l : leaves) is.insert(std::make_tuple(l.level, l.lx[0], l.lx[1], l.lx[2]));
    long created = 0, destroyed = 0;
    for (const auto &k : is) created += was.count(k) ? 0 : 1;
    for (const auto &k : was) destroyed += is.count(k) ? 0 : 1;
    h.last_remesh[0] = static_cast<long>(was.size()), h.last_remesh[1] = static_cast<long>(is.size());
    h.last_remesh[2] = created, h.last_remesh[3] = destroyed;
  }
  const double build_before = h.remesh_build_s, adopt_before = h.remesh_adopt_s;
  const auto t_build = std::chrono::steady_clock::now();
  const bool lean = !initial && !artemis::opt(artemis::OPT_FULL_REMESH);
  std::unique_ptr<artemis_sim_impl> np;
  try {
    if (lean) h.p->release_for_adoption();
    np = build_state(h, &leaves, lean);
    np->ml_fused = np->ml_fused_possible && h.p->ml_fused; // (artemis_sim_set_path outlives a remesh)
    if (!initial) {
      h.remesh_build_s += since(t_build);
      const auto t_adopt = std::chrono::steady_clock::now();
      np->adopt_state_from(*h.p);
      h.remesh_adopt_s += since(t_adopt);
    }
  } catch (const std::exception &e) {
    if (np) release_impl(np);
    if (lean) // the old state has given up its primitives, fluxes, coarse buffers and tables: it cannot step again
      h.dead = std::string("remesh failed after the old state released its work arrays (") + e.what() +
               "); the simulation cannot continue -- ARTEMIS_FULL_REMESH=1 keeps the old state whole at twice the peak memory";
    throw;
  }
  artemis_rt_device_sync();
  np->reuse_from = nullptr, np->reuse_lookup.clear(); // (the old state goes away)
  release_impl(h.p);
  h.p = std::move(np);
  // What is left in artemis_rt's buffer cache now belonged to the old mesh.  It stays there (round 6): the cache's own
  // limit returns the oldest buffers a few at a time as newer ones arrive.  Returning everything after every remesh
  // (round 5, ARTEMIS_TRIM_POOL=1) kept the footprint 30 % lower but stalled the device for 1.1 - 1.8 s whenever the
  // slabs of a whole size class went back at once -- three times in fifteen remeshes of the growing configs[4] mesh,
  // sixty cycles of work each (scripts/remesh_cycles.py).
  if (!initial && artemis::opt(artemis::OPT_TRIM_POOL)) artemis_rt_pool_trim(0);
  // counters of leaves that no longer exist are dropped; new leaves start at zero
  std::map<std::tuple<int, int, int, int>, int> keep;
  for (const artemis_host::Leaf &l : h.p->tree_leaves) {
    const std::tuple<int, int, int, int> k(l.level, l.lx[0], l.lx[1], l.lx[2]);
    auto it = h.deref_count.find(k);
    keep[k] = (it == h.deref_count.end()) ? 0 : it->second;
  }
  h.deref_count.swap(keep);
  h.remeshes++;
  if (!initial) {
    h.remesh_s += since(t_start), h.remesh_n++;
    h.last_remesh_s[0] = since(t_start), h.last_remesh_s[1] = h.remesh_build_s - build_before;
    h.last_remesh_s[2] = h.remesh_adopt_s - adopt_before;
  }
  return true;
}

extern "C" {

const char *artemis_sim_last_error(void) { return g_sim_err.c_str(); }

artemis_sim_t *artemis_sim_create(const char *deck_text, int noverrides,
                                  const char *const *overrides, const artemis_comm_t *comm) {
  if (!deck_text) {
    g_sim_err = "null deck";
    return nullptr;
  }
  artemis_sim *s = new artemis_sim();
  auto build = [&]() {
    s->deck = deck_text;
    for (int q = 0; q < noverrides; ++q) s->overrides.push_back(overrides[q]);
    if (comm) {
      s->comm = *comm;
      s->has_comm = true;
    }
    s->p = build_state(*s, nullptr);
    if (s->p->adaptive && s->p->refine_field && !artemis::opt(artemis::OPT_NO_POOL)) {
      // adaptive meshes re-allocate tens of GB per remesh: artemis_rt's buffer cache (off by default for library hosts)
      const long gb = artemis::opt(artemis::OPT_POOL_GB); // (ARTEMIS_POOL_GB / artemis_hip_set_option("pool_gb", n); default 64)
      artemis_rt_pool_limit(static_cast<size_t>(gb > 0 ? gb : 64) << 30);
    }
    // Mesh::Initialize's refinement loop (upstream): tag the initial condition, refine, regenerate -- until the mesh
    // stops changing (blocks that 2:1 balance created are tagged one pass later than the ones tags created, so this
    // can take more than numlevel passes; the cap only guards against a criterion that never settles)
    for (int pass = 0; s->p->adaptive && pass < 8 * (s->p->amr_max_level + 2); ++pass)
      if (!remesh(*s, true)) break;
    if (s->p->adaptive) artemis_rt_pool_trim(0); // (what the smaller meshes of the loop left in artemis_rt's buffer cache)
  };
  GUARD(build(), {
    artemis_sim_destroy(s);
    return nullptr;
  })
  return s;
}
void artemis_sim_destroy(artemis_sim_t *sim) {
  if (!sim) return;
  artemis_rt_device_sync();
  release_impl(sim->p);
  delete sim;
}
long artemis_sim_evolve(artemis_sim_t *sim, long max_cycles) {
  long n = -1;
  if (!sim->dead.empty()) {
    g_sim_err = sim->dead;
    return -1;
  }
  if (!(sim->p->adaptive && sim->p->refine_field)) {
    GUARD(n = sim->p->evolve(max_cycles), return -1)
    return n;
  }
  // adaptive mesh: one cycle at a time, a remesh check after each (the state object may be replaced)
  auto run = [&]() {
    n = 0;
    while (max_cycles < 0 || n < max_cycles) {
      const long k = sim->p->evolve(1);
      if (k <= 0) break;
      n += k;
      remesh(*sim, false);
    }
  };
  GUARD(run(), return -1)
  return n;
}
double artemis_sim_time(const artemis_sim_t *s) { return s->p->time; }
double artemis_sim_dt(const artemis_sim_t *s) { return s->p->dt; }
double artemis_sim_tlim(const artemis_sim_t *s) { return s->p->tlim; }
long artemis_sim_ncycle(const artemis_sim_t *s) { return s->p->ncycle; }
long artemis_sim_local_zones(const artemis_sim_t *s) {
  return static_cast<long>(s->p->nb) * s->p->mbnx[0] * s->p->mbnx[1] * s->p->mbnx[2];
}
long artemis_sim_total_zones(const artemis_sim_t *s) {
  return s->p->nblocks_global * s->p->mbnx[0] * s->p->mbnx[1] * s->p->mbnx[2];
}
int artemis_sim_block_level(const artemis_sim_t *s, int block) {
  return (block >= 0 && block < s->p->nb) ? s->p->blocks[block].level : -1;
}
long artemis_sim_nblocks_global(const artemis_sim_t *s) { return s->p->nblocks_global; }
int artemis_sim_uses_fused_path(const artemis_sim_t *s) { return (s->p->use_fused || (s->p->multilevel && s->p->ml_fused)) ? 1 : 0; }
int artemis_sim_uses_tuned_kernel(const artemis_sim_t *s) { return (s->p->use_fused && s->p->tuned) ? 1 : 0; }
const char *artemis_sim_stage_kernel(const artemis_sim_t *s) {
  if (s->p->multilevel && s->p->ml_fused) {
    if (s->p->ml_tuned) return "stage_fused_kernel + coarse-fine fix-up";
    if (s->p->general_variant == 3) return "stage_curv_kernel + coarse-fine fix-up";
    return s->p->general_variant == 2 ? "stage_fused_kernel<curvilinear> + coarse-fine fix-up" : "stage_cell_kernel + coarse-fine fix-up";
  }
  if (!s->p->use_fused) return "per-task chain";
  if (s->p->tuned) return "stage_fused_kernel";
  if (s->p->general_variant < 0) return "general stage (not run yet)";
  if (s->p->general_variant == 3) return "stage_curv_kernel";
  return s->p->general_variant == 1 ? "stage2d_kernel" : (s->p->general_variant == 2 ? "stage_fused_kernel<curvilinear>" : "stage_cell_kernel");
}
long artemis_sim_remeshes(const artemis_sim_t *s) { return s->remeshes; }
double artemis_sim_load_balance(const artemis_sim_t *s) { return s->p->lb_max_over_mean; }
int artemis_sim_force_refine(artemis_sim_t *s, long gid) {
  int changed = 0;
  if (!s->dead.empty()) {
    g_sim_err = s->dead;
    return -1;
  }
  GUARD(changed = remesh(*s, false, gid) ? 1 : 0, return -1)
  return changed;
}
void artemis_sim_last_remesh(const artemis_sim_t *s, long *leaves4, double *seconds3) {
  for (int q = 0; q < 4 && leaves4; ++q) leaves4[q] = s->last_remesh[q];
  for (int q = 0; q < 3 && seconds3; ++q) seconds3[q] = s->last_remesh_s[q];
}
int artemis_sim_inject_refine_tags(artemis_sim_t *s, const long *gids, int n) {
  int changed = 0;
  if (!s->dead.empty()) {
    g_sim_err = s->dead;
    return -1;
  }
  const std::vector<long> v(gids, gids + (n > 0 ? n : 0));
  GUARD(changed = remesh(*s, false, -1, &v) ? 1 : 0, return -1)
  return changed;
}
long artemis_sim_remesh_seconds(const artemis_sim_t *s, double *out) {
  if (out) out[0] = s->remesh_s, out[1] = s->remesh_build_s, out[2] = s->remesh_adopt_s, out[3] = s->remesh_tag_s;
  return s->remesh_n;
}
int artemis_sim_set_path(artemis_sim_t *s, const char *which) {
  const std::string w = which ? which : "";
  if (s->p->multilevel && (w == "fused" || w == "unfused")) {
    if (w == "fused" && !s->p->ml_fused_possible) {
      g_sim_err = "refined meshes: the one-kernel stages do not cover drag, n-body gravity or the curvilinear decks of the cell-centred stage";
      return 1;
    }
    s->p->ml_fused = (w == "fused");
    return 0;
  }
  if (w == "fused") {
    if (!s->p->fused_possible) {
      g_sim_err = "the fused paths do not cover gas diffusion";
      return 1;
    }
    s->p->use_fused = true;
    return 0;
  }
  if (w == "unfused") {
    GUARD(s->p->ensure_unfused(), return 1)
    s->p->use_fused = false;
    return 0;
  }
  g_sim_err = "path must be fused|unfused";
  return 1;
}
int artemis_sim_set_overlap(artemis_sim_t *s, int overlap) {
  if (overlap < 0 || overlap > 2) {
    g_sim_err = "overlap must be 0 (off), 1 (two launches) or 2 (in-kernel signalling)";
    return 1;
  }
  s->p->overlap = overlap;
  return 0;
}
void artemis_sim_set_kernel_timing(artemis_sim_t *s, int on) { s->p->time_kernels = on != 0; }
int artemis_sim_set_dropin(artemis_sim_t *s, int on) {
  if (on && !(s->p->use_fused && s->p->tuned)) {
    g_sim_err = "drop-in accounting applies to the tuned fused kernel only";
    return 1;
  }
  s->p->dropin = (on == 2) ? 2 : (on ? 1 : 0);
  return 0;
}
int artemis_sim_overlap(const artemis_sim_t *s) { return s->p->overlap; }
int artemis_sim_nbody_force(artemis_sim_t *s, double *out, int reset) {
  const int n = static_cast<int>(s->p->particles.size());
  if (!out) return n; // size query
  try {
    s->p->flush_nbody_force();
  } catch (const std::exception &e) {
    g_sim_err = e.what();
    return -1;
  }
  std::vector<double> f = s->p->particle_force;
  if (n && s->p->has_comm && s->p->nranks > 1 && s->p->comm.allreduce_sum(s->p->comm.ctx, f.data(), 7 * n)) return -1; // nbody_advance.cpp:123-131
  for (int q = 0; q < 7 * n; ++q) out[q] = f[q];
  if (reset) std::fill(s->p->particle_force.begin(), s->p->particle_force.end(), 0.0);
  return n;
}
void artemis_sim_species(const artemis_sim_t *s, int *ns_gas, int *ns_dust) {
  if (ns_gas) *ns_gas = s->p->ns_gas;
  if (ns_dust) *ns_dust = s->p->ns_dust;
}
void artemis_sim_dims(const artemis_sim_t *s, int *d) {
  d[0] = s->p->nb, d[1] = s->p->ni, d[2] = s->p->nj, d[3] = s->p->nk, d[4] = s->p->is, d[5] = s->p->ie;
  d[6] = s->p->js, d[7] = s->p->je, d[8] = s->p->ks, d[9] = s->p->ke, d[10] = s->p->ng;
}
void artemis_sim_block_bounds(const artemis_sim_t *s, int b, double *o) {
  for (int d = 0; d < 3; ++d) o[2 * d] = s->p->blocks[b].xmin[d], o[2 * d + 1] = s->p->blocks[b].xmax[d];
}
int artemis_sim_get_field(artemis_sim_t *s, const char *field, int block, double *host_out) {
  const std::string f = field ? field : "";
  if (block < 0 || block >= s->p->nb) {
    g_sim_err = "bad block";
    return -1;
  }
  int nv = -1;
  GUARD(
      {
        s->p->materialise_cons();
        const Field *F = nullptr;
        if (f == "gas.prim") F = &s->p->gprim[s->p->base];
        else if (f == "gas.cons") F = &s->p->gu0;
        else if (f == "dust.prim") F = &s->p->dprim[s->p->base];
        else if (f == "dust.cons") F = &s->p->du0;
        else throw std::runtime_error("unknown field " + f);
        const std::vector<Real> h = s->p->download(*F, block);
        std::memcpy(host_out, h.data(), h.size() * sizeof(Real));
        nv = F->nvar;
      },
      return -1)
  return nv;
}
int artemis_sim_history(artemis_sim_t *s, double *out) {
  int n = -1;
  GUARD(n = s->p->history(out), return -1)
  return n;
}
int artemis_sim_errors(artemis_sim_t *s, double *out) {
  int n = -1;
  GUARD(n = s->p->errors(out), return -1)
  return n;
}
double artemis_sim_last_wall_seconds(const artemis_sim_t *s) { return s->p->last_wall; }
double artemis_sim_kernel_ms(const artemis_sim_t *s, long *nlaunch) {
  if (nlaunch) *nlaunch = s->p->kernel_launches;
  return s->p->kernel_launches ? s->p->kernel_ms_sum / s->p->kernel_launches : 0.0;
}

} // extern "C"
