// TEST HARNESS (tests/test_integration_adapter.py): runs the reference's task list for a time step
// (ArtemisDriver<GEOM>::StepTasks, artemis_driver.cpp:145-273, in its order) through
// integration/artemis_hip_adapter.hpp on host arrays behind the stand-in SparsePack (tests/mock_parthenon/), linked
// against the CPU test double of the library (tests/_build/libartemis_cpudouble.so).  Two MeshData partitions with one
// block each (independent problems with outflow conditions, which this harness applies in Parthenon's order), a u0
// and a u1 register per partition, RK2.  The Python side generates the initial primitives, runs the oracle on the
// same data and compares bit for bit.
//
//   run_stage <gas|full|fused> <in.bin> <out.bin> <dt> <nsteps> <realloc>
//     gas   : gas, HLLC + PLM, per-task forwarders
//     full  : gas + one dust species, uniform gravity, shearing box, simple_dust drag, constant viscosity
//     fused : gas through the opt-in StageFused / StageFusedFillDerived forwarders
//     stage_gas / stage_full : the `gas` / `full` problems through the DEFAULT stage task of INTEGRATION.md section 3
//     covered_multilevel     : Mesh::multilevel set -> StageCovered must be false and Stage must refuse (exit 0)
//             (ArtemisHip::StageCovered -> Stage -> boundary conditions -> StageFillDerived: the tuned kernel for gas
//             alone, artemis_hip_stage_general with the diffusion-flux tasks inside for gas + dust + gravity +
//             shearing box + drag + viscosity)
//     realloc = 1: after the first step every variable moves to a new allocation (what a remesh or a restart does
//                  to the addresses) -- the adapter has to notice by itself
//     realloc = 2: ONE partition holding both blocks; after the first step the SECOND block is replaced by a new
//                  MeshBlock / MeshBlockData object with another gid and logical location and fresh allocations (what
//                  a remesh does to a refined, derefined or migrated block) while the first block stays where it
//                  was -- the adapter's tables of the second block must follow
#include <cstdio>
#include <cstdlib>
#include <string>

#include "artemis_hip_adapter.hpp"

int parthenon::Globals::nghost = 2;
using namespace parthenon;

namespace {
const int NX[3] = {32, 8, 8}, NG = 2;
const int NI = NX[0] + 2 * NG, NJ = NX[1] + 2 * NG, NK = NX[2] + 2 * NG;
const size_t N = static_cast<size_t>(NI) * NJ * NK;

std::shared_ptr<Variable> make_var(int ncomp, bool fluxes, bool faces) {
  auto v = std::make_shared<Variable>();
  v->ncomp = ncomp, v->N = N, v->sj = NI, v->sk = static_cast<size_t>(NI) * NJ;
  v->data.assign(ncomp * N, 0.0);
  for (int d = 0; d < 3; ++d) {
    if (fluxes) v->flux[d].assign(ncomp * N, 0.0);
    if (faces) v->face[d].assign(ncomp * N, 0.0);
  }
  return v;
}
struct Partition {
  MeshBlock blk;
  std::shared_ptr<MeshBlockData<Real>> b0 = std::make_shared<MeshBlockData<Real>>(), b1 = std::make_shared<MeshBlockData<Real>>();
  MeshData<Real> u0, u1;
};
const char *CONS[] = {"gas.cons.density", "gas.cons.momentum", "gas.cons.total_energy", "gas.cons.internal_energy",
                      "dust.cons.density", "dust.cons.momentum"};
void build(Partition &P, Mesh *pm, int id, bool dust, bool diffusion) {
  for (int d = 0; d < 3; ++d) P.blk.coords.dx[d] = 1.0 / 16.0;
  P.blk.coords.xf0[0] = -1.0 - NG / 16.0, P.blk.coords.xf0[1] = -0.25 - NG / 16.0, P.blk.coords.xf0[2] = -0.25 - NG / 16.0;
  auto &v = P.b0->vars;
  v["gas.prim.density"] = make_var(1, true, false), v["gas.prim.velocity"] = make_var(3, true, false);
  v["gas.prim.pressure"] = make_var(1, true, false), v["gas.prim.sie"] = make_var(1, true, false);
  v["gas.cons.density"] = make_var(1, true, false), v["gas.cons.momentum"] = make_var(3, true, false);
  v["gas.cons.total_energy"] = make_var(1, true, false), v["gas.cons.internal_energy"] = make_var(1, true, false);
  v["gas.face.velocity"] = make_var(1, false, true);
  if (diffusion) v["gas.diff.momentum"] = make_var(3, false, true), v["gas.diff.energy"] = make_var(1, false, true);
  if (dust) {
    v["dust.prim.density"] = make_var(1, false, false), v["dust.prim.velocity"] = make_var(3, false, false);
    v["dust.cons.density"] = make_var(1, true, false), v["dust.cons.momentum"] = make_var(3, true, false);
  }
  // u1 = Add("u1", u0) (artemis_driver.cpp:137-139): its own copies of the independent (conserved) fields, the OneCopy
  // primitives shared with u0
  P.b1->vars = P.b0->vars;
  for (const char *c : CONS)
    if (v.count(c)) P.b1->vars[c] = make_var(v[c]->ncomp, false, false);
  P.b0->pmb = P.b1->pmb = &P.blk;
  for (MeshData<Real> *md : {&P.u0, &P.u1}) {
    md->pm = pm, md->partition = id;
    md->ib = {NG, NG + NX[0] - 1}, md->jb = {NG, NG + NX[1] - 1}, md->kb = {NG, NG + NX[2] - 1};
  }
  P.u0.blocks = {P.b0}, P.u1.blocks = {P.b1};
}
// every variable moves to a fresh allocation with the same contents
void reallocate(Partition &P) {
  std::map<Variable *, std::shared_ptr<Variable>> moved;
  for (auto *blk : {P.b0.get(), P.b1.get()})
    for (auto &kv : blk->vars) {
      auto it = moved.find(kv.second.get());
      if (it == moved.end()) it = moved.emplace(kv.second.get(), std::make_shared<Variable>(*kv.second)).first;
      kv.second = it->second;
    }
}
// the block is replaced: new MeshBlock (other gid / logical location), new MeshBlockData objects, new allocations
// with the same contents -- the old objects stay alive (a stale table would silently keep writing into them)
std::vector<std::shared_ptr<MeshBlockData<Real>>> graveyard;
std::vector<std::shared_ptr<MeshBlock>> new_blocks;
void replace_block(Partition &P) {
  auto nb = std::make_shared<MeshBlock>(P.blk);
  nb->gid = P.blk.gid + 100, nb->loc.lev = P.blk.loc.lev + 1, nb->loc.l1 = 2 * P.blk.loc.l1 + 1;
  new_blocks.push_back(nb);
  graveyard.push_back(P.b0), graveyard.push_back(P.b1);
  auto n0 = std::make_shared<MeshBlockData<Real>>(*P.b0), n1 = std::make_shared<MeshBlockData<Real>>(*P.b1);
  n0->pmb = n1->pmb = nb.get();
  P.b0 = n0, P.b1 = n1;
  reallocate(P);
}
// parthenon's outflow condition on the FillGhost primitives: x1, then x2, then x3, each over the entire extent of the others
void outflow(Partition &P, bool dust) {
  std::vector<std::pair<Variable *, int>> fields;
  auto &v = P.b0->vars;
  fields.push_back({v["gas.prim.density"].get(), 0});
  for (int c = 0; c < 3; ++c) fields.push_back({v["gas.prim.velocity"].get(), c});
  fields.push_back({v["gas.prim.sie"].get(), 0});
  if (dust) {
    fields.push_back({v["dust.prim.density"].get(), 0});
    for (int c = 0; c < 3; ++c) fields.push_back({v["dust.prim.velocity"].get(), c});
  }
  for (auto &f : fields) {
    Real *q = f.first->data.data() + f.second * N;
    auto at = [&](int k, int j, int i) -> Real & { return q[(static_cast<size_t>(k) * NJ + j) * NI + i]; };
    for (int k = 0; k < NK; ++k)
      for (int j = 0; j < NJ; ++j)
        for (int g = 0; g < NG; ++g) at(k, j, g) = at(k, j, NG), at(k, j, NG + NX[0] + g) = at(k, j, NG + NX[0] - 1);
    for (int k = 0; k < NK; ++k)
      for (int g = 0; g < NG; ++g)
        for (int i = 0; i < NI; ++i) at(k, g, i) = at(k, NG, i), at(k, NG + NX[1] + g, i) = at(k, NG + NX[1] - 1, i);
    for (int g = 0; g < NG; ++g)
      for (int j = 0; j < NJ; ++j)
        for (int i = 0; i < NI; ++i) at(g, j, i) = at(NG, j, i), at(NG + NX[2] + g, j, i) = at(NG + NX[2] - 1, j, i);
  }
}
} // namespace

int main(int argc, char **argv) {
  if (argc < 7) return 2;
  const std::string mode = argv[1];
  const Real dt = std::atof(argv[4]);
  const int nsteps = std::atoi(argv[5]);
  const int realloc_mode = std::atoi(argv[6]);
  const bool realloc_after_first = realloc_mode == 1, one_partition = realloc_mode == 2;
  const bool stage_task = mode == "stage_gas" || mode == "stage_full";
  const bool full = mode == "full" || mode == "stage_full", fused = mode == "fused";
  const bool dust = full, diffusion = full;

  Mesh mesh;
  auto pkg = [&](const char *n) { return mesh.packages.m[n] = std::make_shared<StateDescriptor>(); };
  auto art = pkg("artemis");
  art->AddParam("do_gas", true), art->AddParam("do_dust", dust), art->AddParam("coords", Coordinates::cartesian);
  art->AddParam("do_rotating_frame", full), art->AddParam("do_viscosity", diffusion), art->AddParam("do_conduction", false);
  art->AddParam("do_gravity", full), art->AddParam("do_drag", full), art->AddParam("do_nbody", false);
  art->AddParam("do_cooling", false), art->AddParam("do_radiation", false), art->AddParam("do_diffusion", diffusion);
  auto gas = pkg("gas");
  gas->AddParam("nspecies", 1), gas->AddParam("adiabatic_index", Real(1.4));
  gas->AddParam("recon", ReconstructionMethod::plm), gas->AddParam("rsolver", RSolver::hllc);
  gas->AddParam("dfloor", Real(1e-10)), gas->AddParam("siefloor", Real(1e-10)), gas->AddParam("de_switch", Real(0.0));
  gas->AddParam("cfl", Real(0.3)), gas->AddParam("do_viscosity", diffusion), gas->AddParam("do_conduction", false);
  gas->AddParam("do_diffusion", diffusion), gas->AddParam("eos_h", ArtemisUtils::EOS{0.4, 2.5});
  if (diffusion) {
    Diffusion::DiffCoeffParams dp;
    dp.type = Diffusion::DiffType::viscosity_plaw, dp.avg = Diffusion::DiffAvg::arithmetic, dp.nu_s = 0.01, dp.eta = 0.0, dp.r_exp = 0.0, dp.R0 = 1.0;
    gas->AddParam("visc_params", dp);
  }
  if (dust) {
    auto d = pkg("dust");
    d->AddParam("nspecies", 1), d->AddParam("recon", ReconstructionMethod::plm), d->AddParam("rsolver", RSolver::hlle);
    d->AddParam("dfloor", Real(1e-10)), d->AddParam("cfl", Real(0.3)), d->AddParam("grain_density", Real(1.0));
  }
  if (full) {
    auto g = pkg("gravity");
    g->AddParam("type", Gravity::GravityType::uniform);
    g->AddParam("tstart", std::numeric_limits<Real>::lowest()), g->AddParam("tstop", Real(1.7976931348623157e308));
    g->AddParam("gx1", Real(0.1)), g->AddParam("gx2", Real(-0.2)), g->AddParam("gx3", Real(0.05));
    auto rf = pkg("rotating_frame");
    rf->AddParam("omega", Real(1.0)), rf->AddParam("qshear", Real(1.5));
    auto dr = pkg("drag");
    dr->AddParam("type", Drag::Coupling::simple_dust);
    dr->AddParam("gas_self_drag", Drag::SelfDragParams()), dr->AddParam("dust_self_drag", Drag::SelfDragParams());
    dr->AddParam("x1min", Real(-1.0)), dr->AddParam("x1max", Real(1.0)), dr->AddParam("x2min", Real(-0.25));
    dr->AddParam("x2max", Real(0.25)), dr->AddParam("x3min", Real(-0.25)), dr->AddParam("x3max", Real(0.25));
    Drag::StoppingTimeParams sp;
    sp.model = Drag::DragModel::constant, sp.scale = 1.0, sp.tau = ParArray1D<Real>("tau", 1), sp.tau(0) = 0.1;
    dr->AddParam("stopping_time_params", sp);
  }

  Partition part[2];
  for (int q = 0; q < 2; ++q) build(part[q], &mesh, q, dust, diffusion);
  for (int q = 0; q < 2; ++q) part[q].blk.gid = q, part[q].blk.loc.l1 = q;
  const int npart = one_partition ? 1 : 2;
  auto gather = [&]() { // realloc = 2: partition 0 holds both blocks
    if (!one_partition) return;
    part[0].u0.blocks = {part[0].b0, part[1].b0}, part[0].u1.blocks = {part[0].b1, part[1].b1};
  };
  gather();

  // initial primitives (entire blocks) from the Python side: per partition gas [6][N] then dust [4][N]
  FILE *f = std::fopen(argv[2], "rb");
  if (!f) return 3;
  auto rd = [&](Variable *v, int comp) { return std::fread(v->data.data() + comp * N, sizeof(Real), N, f) == N; };
  for (int q = 0; q < 2; ++q) {
    auto &v = part[q].b0->vars;
    bool ok = rd(v["gas.prim.density"].get(), 0);
    for (int c = 0; c < 3; ++c) ok = ok && rd(v["gas.prim.velocity"].get(), c);
    ok = ok && rd(v["gas.prim.pressure"].get(), 0) && rd(v["gas.prim.sie"].get(), 0);
    if (dust) {
      ok = ok && rd(v["dust.prim.density"].get(), 0);
      for (int c = 0; c < 3; ++c) ok = ok && rd(v["dust.prim.velocity"].get(), c);
    }
    if (!ok) return 4;
  }
  std::fclose(f);

  LowStorageIntegrator integ; // rk2
  integ.nstages = 2, integ.dt = dt, integ.gam0 = {0.0, 0.5}, integ.gam1 = {1.0, 0.5}, integ.beta = {1.0, 0.5};
  Real dt_est = 0.0;
  if (mode == "covered_multilevel") {
    // a refined mesh: the default wiring must fall back to the per-task list (the one-kernel stages store no flux
    // arrays for artemis_driver.cpp:196-202's flux correction) and Stage itself must refuse, loudly
    if (!ArtemisHip::StageCovered(&part[0].u0)) return 10; // (sanity: covered on the uniform mesh)
    mesh.multilevel = true;
    if (ArtemisHip::StageCovered(&part[0].u0)) return 8;
    try {
      ArtemisHip::Stage(&part[0].u0, 1, &integ, false, 0.0);
    } catch (const std::exception &e) {
      return std::string(e.what()).find("per-task list") != std::string::npos ? 0 : 11;
    }
    return 9;
  }
  try {
    for (int q = 0; q < npart; ++q) ArtemisHip::PrimToCons(&part[q].u0); // PostInitialization (main.cpp:43)
    Real time = 0.0;
    for (int step = 0; step < nsteps; ++step) {
      for (int q = 0; q < npart; ++q) ArtemisHip::DeepCopyConservedData(&part[q].u1, &part[q].u0); // artemis_driver.cpp:157-163
      for (int stage = 1; stage <= integ.nstages; ++stage) {
        const Real bdt = integ.beta[stage - 1] * integ.dt;
        for (int q = 0; q < npart; ++q) { // one task list per partition (artemis_driver.cpp:170-262)
          MeshData<Real> *u0 = &part[q].u0, *u1 = &part[q].u1;
          auto bcs = [&]() {
            outflow(part[q], dust);
            if (one_partition) outflow(part[1], dust);
          };
          if (fused) {
            ArtemisHip::StageFused(u0, stage, &integ, false);
            bcs();
            ArtemisHip::StageFusedFillDerived(u0);
            continue;
          }
          if (stage_task) { // the default wiring: one task per stage where the library covers the package set
            if (!ArtemisHip::StageCovered(u0)) return 6;
            if (ArtemisHip::StageTakesTunedKernel(u0) != (mode == "stage_gas")) return 7;
            ArtemisHip::Stage(u0, stage, &integ, false, time);
            bcs();
            ArtemisHip::StageFillDerived(u0);
            continue;
          }
          ArtemisHip::GasCalculateFluxes(u0, false);
          if (dust) ArtemisHip::DustCalculateFluxes(u0, false);
          if (diffusion) ArtemisHip::GasZeroDiffusionFlux(u0), ArtemisHip::GasViscousFlux(u0);
          ArtemisHip::ApplyUpdate(u0, u1, stage, &integ);
          ArtemisHip::GasFluxSource(u0, bdt);
          if (dust) ArtemisHip::DustFluxSource(u0, bdt);
          if (diffusion) ArtemisHip::GasDiffusionUpdate(u0, bdt);
          if (full) ArtemisHip::ExternalGravity(u0, time, bdt), ArtemisHip::RotatingFrameForce(u0, time, bdt), ArtemisHip::DragSource(u0, time, bdt);
          ArtemisHip::SetAuxillaryFields(u0);
          ArtemisHip::ConsToPrim(u0);
          bcs();
          ArtemisHip::PrimToCons(u0);
        }
      }
      time += integ.dt;
      if (step == 0 && realloc_after_first)
        for (int q = 0; q < 2; ++q) reallocate(part[q]);
      if (step == 0 && one_partition) replace_block(part[1]), gather();
    }
    // PostStepTasks: EstimateTimestep on the "base" MeshData (artemis_driver.cpp:286-288) -- here u0
    dt_est = ArtemisHip::GasEstimateTimestepMesh(&part[0].u0);
    if (dust) dt_est = std::min(dt_est, ArtemisHip::DustEstimateTimestepMesh(&part[0].u0));
  } catch (const std::exception &e) {
    std::fprintf(stderr, "run_stage: %s\n", e.what());
    return 5;
  }

  f = std::fopen(argv[3], "wb");
  auto wr = [&](Variable *v, int comp) { std::fwrite(v->data.data() + comp * N, sizeof(Real), N, f); };
  for (int q = 0; q < 2; ++q) {
    auto &v = part[q].b0->vars;
    wr(v["gas.prim.density"].get(), 0);
    for (int c = 0; c < 3; ++c) wr(v["gas.prim.velocity"].get(), c);
    wr(v["gas.prim.pressure"].get(), 0), wr(v["gas.prim.sie"].get(), 0);
    wr(v["gas.cons.density"].get(), 0);
    for (int c = 0; c < 3; ++c) wr(v["gas.cons.momentum"].get(), c);
    wr(v["gas.cons.total_energy"].get(), 0), wr(v["gas.cons.internal_energy"].get(), 0);
    if (dust) {
      wr(v["dust.prim.density"].get(), 0);
      for (int c = 0; c < 3; ++c) wr(v["dust.prim.velocity"].get(), c);
    }
  }
  std::fwrite(&dt_est, sizeof dt_est, 1, f);
  std::fclose(f);
  return 0;
}
