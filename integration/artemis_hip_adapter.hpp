// artemis_hip_adapter.hpp -- the file a maintainer adds to lanl/artemis (src/utils/) to route the hydro tasks
// through libartemis_hip.so.  It fills an `artemis_pack_t` (include/artemis_hip.h) from the SparsePacks the
// reference's own task functions build, once per MeshData partition (again when the mesh or the MeshData changes),
// and forwards each task.  Nothing else in Artemis changes: StateDescriptor registration, the TaskList of
// ArtemisDriver<GEOM>::StepTasks (artemis_driver.cpp:145-273), problem generators and decks stay as they are.
//
// Checked in this repository two ways (tests/test_integration_adapter.py): compiled with -Wall -Werror against a
// stand-in for the few Parthenon / Artemis names it touches (tests/mock_parthenon/), and RUN -- the reference's task
// list for an RK2 step, two MeshData partitions, through these forwarders on the CPU test double of the library --
// against the oracle, bit for bit (tests/adapter_live/run_stage.cpp).  In Artemis it includes the real headers.
#ifndef ARTEMIS_HIP_ADAPTER_HPP_
#define ARTEMIS_HIP_ADAPTER_HPP_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <vector>

#include "artemis.hpp"     // Real, Coordinates, RSolver, ReconstructionMethod, the gas:: / dust:: field types
#include "artemis_hip.h"   // this repository's include/
#include "drag/drag.hpp"                         // Drag::Coupling, SelfDragParams, StoppingTimeParams
#include "gravity/gravity.hpp"                   // Gravity::GravityType, Orbit
#include "nbody/particle_base.hpp"               // NBody::Particle
#include "utils/diffusion/diffusion_coeff.hpp"   // Diffusion::DiffCoeffParams

namespace ArtemisHip {
using parthenon::MeshData;
using parthenon::ParArray1D;
using parthenon::TaskStatus;
using TE = parthenon::TopologicalElement;

// Device tables of one MeshData partition.  What identifies the data they point at is remembered: `sig0` / `sig1`, a
// hash over EVERY block of the u0 / u1 MeshData of its global id, its logical location (level, lx1, lx2, lx3) and the
// address of its MeshBlockData object, plus `probe0` / `probe1`, the address of the first conserved variable of the
// first block, and the block count.  A remesh keeps unchanged blocks as the same objects at the same addresses and
// replaces refined, derefined or migrated ones by new objects with other locations / ids, so any change of any block
// of the partition -- not only of its first one -- changes the signature and the tables are rebuilt on the next call;
// a restart or a reallocation moves the probe.  No hook into Parthenon's remesher is needed (Invalidate() is there
// for hosts that have one).
struct PackCache {
  artemis_pack_t p{};
  ParArray1D<Real *> gprim, gcons0, gcons1, gflux[3], gpflux[3], gvface[3], gdflux[3];
  ParArray1D<Real *> dprim, dcons0, dcons1, dflux[3];
  ParArray1D<Real> geom, metric, plm_table;
  std::vector<Real> geom_host, metric_host;
  const Real *probe0 = nullptr, *probe1 = nullptr;
  std::uint64_t sig0 = 0, sig1 = 0;
  int nb = -1;
  // package tables that depend on the mesh (built on first use after a rebuild)
  ParArray1D<Real> visc_radial_data;
  ParArray1D<const Real *> visc_radial;
  bool visc_radial_built = false;
};

// entry [b * nvar + v] = address of variable v of block b at (k, j, i) = (0, 0, 0)
template <typename Pack>
void FillTable(ParArray1D<Real *> &tab, const Pack &v, const int nb, const int nvar, const int first) {
  tab = ParArray1D<Real *>("artemis_hip table", nb * nvar);
  auto t = tab;
  parthenon::par_for(
      DEFAULT_LOOP_PATTERN, "ArtemisHip::FillTable", parthenon::DevExecSpace(), 0, nb - 1, 0, nvar - 1,
      KOKKOS_LAMBDA(const int b, const int n) { t(b * nvar + n) = &v(b, first + n, 0, 0, 0); });
}
// flux slot of direction dir (1..3) of the same variables (fluid_fluxes.hpp:119-121 writes them)
template <typename Pack>
void FillFluxTable(ParArray1D<Real *> &tab, const Pack &v, const int nb, const int nvar, const int first, const int dir) {
  tab = ParArray1D<Real *>("artemis_hip flux table", nb * nvar);
  auto t = tab;
  parthenon::par_for(
      DEFAULT_LOOP_PATTERN, "ArtemisHip::FillFluxTable", parthenon::DevExecSpace(), 0, nb - 1, 0, nvar - 1,
      KOKKOS_LAMBDA(const int b, const int n) { t(b * nvar + n) = &v.flux(b, dir, first + n, 0, 0, 0); });
}
// face field (gas.face.velocity, gas.diff.*) on the faces of direction dir
template <typename Pack>
void FillFaceTable(ParArray1D<Real *> &tab, const Pack &v, const int nb, const int nvar, const int dir) {
  tab = ParArray1D<Real *>("artemis_hip face table", nb * nvar);
  auto t = tab;
  const TE te = (dir == 1) ? TE::F1 : ((dir == 2) ? TE::F2 : TE::F3);
  parthenon::par_for(
      DEFAULT_LOOP_PATTERN, "ArtemisHip::FillFaceTable", parthenon::DevExecSpace(), 0, nb - 1, 0, nvar - 1,
      KOKKOS_LAMBDA(const int b, const int n) { t(b * nvar + n) = &v(b, te, n, 0, 0, 0); });
}

// the descriptors Gas / Dust::CalculateFluxes build (gas.cpp:479-488, dust.cpp:287-290): pack order rho[n],
// v[ns+3n+d], P[4ns+n], sie[5ns+n]; D[n], M[ns+3n+d], E[4ns+n], e_int[5ns+n]
inline auto &GasPrimDesc(parthenon::ResolvedPackages *res) {
  static auto d = parthenon::MakePackDescriptor<gas::prim::density, gas::prim::velocity, gas::prim::pressure, gas::prim::sie>(
      res, {}, {parthenon::PDOpt::WithFluxes});
  return d;
}
inline auto &GasConsDesc(parthenon::ResolvedPackages *res) {
  static auto d = parthenon::MakePackDescriptor<gas::cons::density, gas::cons::momentum, gas::cons::total_energy,
                                                gas::cons::internal_energy>(res, {}, {parthenon::PDOpt::WithFluxes});
  return d;
}
inline auto &DustPrimDesc(parthenon::ResolvedPackages *res) {
  static auto d = parthenon::MakePackDescriptor<dust::prim::density, dust::prim::velocity>(res);
  return d;
}
inline auto &DustConsDesc(parthenon::ResolvedPackages *res) {
  static auto d = parthenon::MakePackDescriptor<dust::cons::density, dust::cons::momentum>(res, {}, {parthenon::PDOpt::WithFluxes});
  return d;
}
// address of the first conserved variable of the first block: what a MeshData's data "is" for the cache
inline const Real *Probe(MeshData<Real> *md) {
  auto pm = md->GetParentPointer();
  auto *res = pm->resolved_packages.get();
  if (pm->packages.Get("artemis")->template Param<bool>("do_gas")) return &GasConsDesc(res).GetPack(md)(0, 0, 0, 0, 0);
  return &DustConsDesc(res).GetPack(md)(0, 0, 0, 0, 0);
}

// identity of the blocks of a MeshData (FNV-1a over gid, logical location and MeshBlockData address of every block):
// host-side, a handful of integer operations per block and call
inline std::uint64_t Signature(MeshData<Real> *md) {
  std::uint64_t h = 1469598103934665603ull;
  auto mix = [&h](std::uint64_t v) {
    for (int q = 0; q < 8; ++q) h = (h ^ ((v >> (8 * q)) & 0xffu)) * 1099511628211ull;
  };
  const int nb = md->NumBlocks();
  mix(static_cast<std::uint64_t>(nb));
  for (int b = 0; b < nb; ++b) {
    auto &mbd = md->GetBlockData(b);
    const auto *pmb = mbd->GetBlockPointer();
    mix(static_cast<std::uint64_t>(pmb->gid));
    mix(static_cast<std::uint64_t>(pmb->loc.level()));
    mix(static_cast<std::uint64_t>(pmb->loc.lx1())), mix(static_cast<std::uint64_t>(pmb->loc.lx2()));
    mix(static_cast<std::uint64_t>(pmb->loc.lx3()));
    mix(static_cast<std::uint64_t>(reinterpret_cast<std::uintptr_t>(mbd.get())));
  }
  return h;
}

// everything that comes from u0: sizes, parameters, prim / cons0 / flux tables, edge and metric tables
inline void BuildFromU0(PackCache &c, MeshData<Real> *u0) {
  auto pm = u0->GetParentPointer();
  auto &artemis_pkg = pm->packages.Get("artemis");
  const bool do_gas = artemis_pkg->template Param<bool>("do_gas");
  const bool do_dust = artemis_pkg->template Param<bool>("do_dust");
  const int nb = u0->NumBlocks();
  const auto ib = u0->GetBoundsI(parthenon::IndexDomain::interior);
  const auto jb = u0->GetBoundsJ(parthenon::IndexDomain::interior);
  const auto kb = u0->GetBoundsK(parthenon::IndexDomain::interior);
  artemis_pack_t &p = c.p;
  p = artemis_pack_t{};
  p.nblocks = nb;
  p.nghost = parthenon::Globals::nghost;
  p.nx1 = ib.e - ib.s + 1, p.nx2 = jb.e - jb.s + 1, p.nx3 = kb.e - kb.s + 1;
  p.coords = static_cast<int>(artemis_pkg->template Param<Coordinates>("coords")); // same enum order (artemis.hpp:78-86)
  const int ndim = (p.nx3 > 1) ? 3 : ((p.nx2 > 1) ? 2 : 1);
  auto *res = pm->resolved_packages.get();

  if (do_gas) {
    auto &pkg = pm->packages.Get("gas");
    const int ns = pkg->template Param<int>("nspecies");
    p.gm1 = pkg->template Param<Real>("adiabatic_index") - 1.0;
    p.gas.nspecies = ns;
    p.gas.recon = static_cast<int>(pkg->template Param<ReconstructionMethod>("recon"));
    p.gas.riemann = static_cast<int>(pkg->template Param<RSolver>("rsolver"));
    p.gas.dfloor = pkg->template Param<Real>("dfloor"), p.gas.siefloor = pkg->template Param<Real>("siefloor");
    p.gas.de_switch = pkg->template Param<Real>("de_switch");
    static auto dface = parthenon::MakePackDescriptor<gas::face::velocity>(res);
    auto vprim = GasPrimDesc(res).GetPack(u0);
    auto vcons0 = GasConsDesc(res).GetPack(u0);
    auto vface = dface.GetPack(u0);
    FillTable(c.gprim, vprim, nb, 6 * ns, 0), p.gas.prim = c.gprim.data();
    FillTable(c.gcons0, vcons0, nb, 6 * ns, 0), p.gas.cons0 = c.gcons0.data();
    for (int d = 0; d < ndim; ++d) {
      FillFluxTable(c.gflux[d], vcons0, nb, 6 * ns, 0, d + 1), p.gas.flux[d] = c.gflux[d].data();
      // interface pressure = the flux slot of gas.prim.pressure (hllc.hpp:166): pack index 4 ns + n
      FillFluxTable(c.gpflux[d], vprim, nb, ns, 4 * ns, d + 1), p.gas.pflux[d] = c.gpflux[d].data();
      FillFaceTable(c.gvface[d], vface, nb, ns, d + 1), p.gas.vface[d] = c.gvface[d].data(); // hllc.hpp:179
    }
    if (pkg->template Param<bool>("do_diffusion")) { // gas::diff::momentum (3 n + component) then gas::diff::energy (3 ns + n), gas.cpp:276-284
      static auto ddiff = parthenon::MakePackDescriptor<gas::diff::momentum, gas::diff::energy>(res);
      auto vdiff = ddiff.GetPack(u0);
      for (int d = 0; d < ndim; ++d) FillFaceTable(c.gdflux[d], vdiff, nb, 4 * ns, d + 1), p.gas.diff_flux[d] = c.gdflux[d].data();
    }
  }
  if (do_dust) {
    auto &pkg = pm->packages.Get("dust");
    const int ns = pkg->template Param<int>("nspecies");
    p.dust.nspecies = ns;
    p.dust.recon = static_cast<int>(pkg->template Param<ReconstructionMethod>("recon"));
    p.dust.riemann = static_cast<int>(pkg->template Param<RSolver>("rsolver"));
    p.dust.dfloor = pkg->template Param<Real>("dfloor");
    auto vprim = DustPrimDesc(res).GetPack(u0);
    auto vcons0 = DustConsDesc(res).GetPack(u0);
    FillTable(c.dprim, vprim, nb, 4 * ns, 0), p.dust.prim = c.dprim.data();
    FillTable(c.dcons0, vcons0, nb, 4 * ns, 0), p.dust.cons0 = c.dcons0.data();
    for (int d = 0; d < ndim; ++d) FillFluxTable(c.dflux[d], vcons0, nb, 4 * ns, 0, d + 1), p.dust.flux[d] = c.dflux[d].data();
  }
  if (artemis_pkg->template Param<bool>("do_rotating_frame")) // fluid_fluxes.hpp:433-437
    p.omega_frame = pm->packages.Get("rotating_frame")->template Param<Real>("omega");

  // edge table: Coordinates_t::Xf<d>(idx) = xf0 + idx * dx with idx counted from the first ghost zone
  // (geometry.hpp:65-72); host copy kept for the metric / radial tables
  c.geom_host.assign(6 * nb, 0.0);
  for (int b = 0; b < nb; ++b) {
    const auto &pco = u0->GetBlockData(b)->GetBlockPointer()->coords;
    c.geom_host[6 * b + 0] = pco.template Xf<parthenon::X1DIR>(0), c.geom_host[6 * b + 1] = pco.template Dxf<parthenon::X1DIR>();
    c.geom_host[6 * b + 2] = pco.template Xf<parthenon::X2DIR>(0), c.geom_host[6 * b + 3] = pco.template Dxf<parthenon::X2DIR>();
    c.geom_host[6 * b + 4] = pco.template Xf<parthenon::X3DIR>(0), c.geom_host[6 * b + 5] = pco.template Dxf<parthenon::X3DIR>();
  }
  c.geom = ParArray1D<Real>("artemis_hip geom", 6 * nb);
  parthenon::deep_copy_from_host(c.geom, c.geom_host.data(), c.geom_host.size());
  p.geom = c.geom.data();
  // the trigonometry Coords<GEOM> evaluates per cell (spherical.hpp:53-146, ConvertCoordsToCart of every system):
  // tabulated once per (re)mesh with the host libm (count is 0 for Cartesian and spherical1D)
  const long nm = artemis_hip_metric_count(&p);
  c.metric_host.assign(nm > 0 ? nm : 0, 0.0);
  if (nm > 0) {
    PARTHENON_REQUIRE(artemis_hip_metric_fill(&p, c.geom_host.data(), c.metric_host.data()) == 0, artemis_hip_last_error());
    c.metric = ParArray1D<Real>("artemis_hip metric", nm);
    parthenon::deep_copy_from_host(c.metric, c.metric_host.data(), c.metric_host.size());
    p.metric = c.metric.data();
  }
  // PLM_G's geometric weights (plm.hpp:54-73), once per (re)mesh on the device (curvilinear meshes; optional)
  p.plm_table = nullptr;
  if (p.coords != ARTEMIS_CARTESIAN) {
    c.plm_table = ParArray1D<Real>("artemis_hip plm table", artemis_hip_plm_table_count(&p));
    PARTHENON_REQUIRE(artemis_hip_plm_table_fill(&p, c.plm_table.data(), nullptr) == 0, artemis_hip_last_error());
    p.plm_table = c.plm_table.data();
  }
  c.nb = nb;
  c.probe1 = nullptr; // the cons1 tables belong to the previous mesh
  c.visc_radial_built = false;
}
// the u1 register (the start-of-step copy, artemis_driver.cpp:137-139, 157-163): only the cons tables differ
inline void BuildFromU1(PackCache &c, MeshData<Real> *u1) {
  auto pm = u1->GetParentPointer();
  auto *res = pm->resolved_packages.get();
  const int nb = c.nb;
  if (c.p.gas.nspecies > 0) {
    auto v = GasConsDesc(res).GetPack(u1);
    FillTable(c.gcons1, v, nb, 6 * c.p.gas.nspecies, 0), c.p.gas.cons1 = c.gcons1.data();
  }
  if (c.p.dust.nspecies > 0) {
    auto v = DustConsDesc(res).GetPack(u1);
    FillTable(c.dcons1, v, nb, 4 * c.p.dust.nspecies, 0), c.p.dust.cons1 = c.dcons1.data();
  }
}

inline std::map<int, PackCache> &Caches() {
  static std::map<int, PackCache> caches; // one per MeshData partition
  return caches;
}
// For hosts with a remesh hook; not required (GetCache notices changed blocks and moved data by itself).
inline void Invalidate() { Caches().clear(); }

// u0 = the MeshData the task receives; u1 = the start-of-step copy for the tasks that read it (ApplyUpdate,
// DeepCopyConservedData), nullptr otherwise -- the cons1 tables are then left as they are.
inline PackCache &GetCache(MeshData<Real> *u0, MeshData<Real> *u1 = nullptr) {
  PackCache &c = Caches()[u0->GetPartitionId()];
  const Real *p0 = Probe(u0);
  const std::uint64_t s0 = Signature(u0);
  if (c.nb != u0->NumBlocks() || c.probe0 != p0 || c.sig0 != s0) {
    BuildFromU0(c, u0);
    c.probe0 = p0, c.sig0 = s0;
  }
  if (u1) {
    const Real *p1 = Probe(u1);
    const std::uint64_t s1 = Signature(u1);
    if (c.probe1 != p1 || c.sig1 != s1) {
      BuildFromU1(c, u1);
      c.probe1 = p1, c.sig1 = s1;
    }
  }
  return c;
}
inline artemis_pack_t &GetPack(MeshData<Real> *u0, MeshData<Real> *u1 = nullptr) { return GetCache(u0, u1).p; }

inline void *Stream() { return nullptr; } // or Kokkos::HIP().hip_stream(): the library shares the process's HIP runtime

#define ARTEMIS_HIP_TASK(call)                                        \
  do {                                                                \
    const int rc_ = (call);                                           \
    PARTHENON_REQUIRE(rc_ == 0, artemis_hip_last_error());            \
    return TaskStatus::complete;                                      \
  } while (0)

// ---- the task bodies (the reference's signatures; artemis_driver.cpp:157-255) -------------------------------------
inline TaskStatus DeepCopyConservedData(MeshData<Real> *to, MeshData<Real> *from) { // artemis_integrator.hpp:30-51
  ARTEMIS_HIP_TASK(artemis_hip_deep_copy_conserved(&GetPack(from, to), Stream()));
}
inline TaskStatus GasCalculateFluxes(MeshData<Real> *md, const bool pcm) { // Gas::CalculateFluxes, gas.cpp:473-494
  ARTEMIS_HIP_TASK(artemis_hip_calculate_fluxes(&GetPack(md), ARTEMIS_GAS, pcm, Stream()));
}
inline TaskStatus DustCalculateFluxes(MeshData<Real> *md, const bool pcm) { // Dust::CalculateFluxes, dust.cpp:281-298
  ARTEMIS_HIP_TASK(artemis_hip_calculate_fluxes(&GetPack(md), ARTEMIS_DUST, pcm, Stream()));
}
inline TaskStatus ApplyUpdate(MeshData<Real> *u0, MeshData<Real> *u1, const int stage,
                              const parthenon::LowStorageIntegrator *integ) { // artemis_integrator.hpp:57-110
  const Real g0 = integ->gam0[stage - 1], g1 = integ->gam1[stage - 1], bdt = integ->beta[stage - 1] * integ->dt;
  ARTEMIS_HIP_TASK(artemis_hip_apply_update(&GetPack(u0, u1), g0, g1, bdt, Stream()));
}
inline TaskStatus GasFluxSource(MeshData<Real> *md, const Real dt) { // Gas::FluxSource, gas.cpp:499-519
  ARTEMIS_HIP_TASK(artemis_hip_flux_source(&GetPack(md), ARTEMIS_GAS, dt, Stream()));
}
inline TaskStatus DustFluxSource(MeshData<Real> *md, const Real dt) { // Dust::FluxSource, dust.cpp:303-326
  ARTEMIS_HIP_TASK(artemis_hip_flux_source(&GetPack(md), ARTEMIS_DUST, dt, Stream()));
}
inline TaskStatus SetAuxillaryFields(MeshData<Real> *md) { // fill_derived.cpp:30-75
  ARTEMIS_HIP_TASK(artemis_hip_set_aux(&GetPack(md), Stream()));
}
inline void ConsToPrim(MeshData<Real> *md) { // PreCommFillDerivedMesh, artemis.cpp:122
  PARTHENON_REQUIRE(artemis_hip_cons_to_prim(&GetPack(md), Stream()) == 0, artemis_hip_last_error());
}
inline void PrimToCons(MeshData<Real> *md) { // PreFillDerivedMesh, artemis.cpp:123
  PARTHENON_REQUIRE(artemis_hip_prim_to_cons(&GetPack(md), Stream()) == 0, artemis_hip_last_error());
}

// ---- gas diffusion (artemis_driver.cpp:189-193, :218-221) ----------------------------------------------------------
inline artemis_diffcoeff_t Coeff(const Diffusion::DiffCoeffParams &dp) { // diffusion_coeff.hpp:58-136
  artemis_diffcoeff_t c;
  std::memset(&c, 0, sizeof c);
  c.avg = (dp.avg == Diffusion::DiffAvg::harmonic) ? 1 : 0;
  switch (dp.type) {
  case Diffusion::DiffType::viscosity_plaw:
    c.type = ARTEMIS_VISCOSITY_PLAW, c.coeff = dp.nu_s, c.eta = dp.eta, c.r_exp = dp.r_exp, c.r0 = dp.R0;
    break;
  case Diffusion::DiffType::viscosity_alpha:
    c.type = ARTEMIS_VISCOSITY_ALPHA, c.coeff = dp.alpha, c.eta = dp.eta, c.r0 = dp.R0, c.omega0 = dp.Omega0;
    break;
  case Diffusion::DiffType::conductivity_plaw:
    c.type = ARTEMIS_CONDUCTIVITY_PLAW, c.coeff = dp.hcond_0;
    c.temp_exp = dp.temp_exp, c.rho_exp = dp.rho_exp, c.rho_ref = dp.d0, c.T_ref = dp.T0;
    break;
  case Diffusion::DiffType::thermaldiff_plaw:
    c.type = ARTEMIS_THERMALDIFF_PLAW, c.coeff = dp.kappa_0;
    c.temp_exp = dp.temp_exp, c.rho_exp = dp.rho_exp, c.rho_ref = dp.d0, c.T_ref = dp.T0;
    break;
  default: c.type = ARTEMIS_DIFF_OFF;
  }
  return c;
}
// the gas package's diffusion parameters as the library takes them; the radial factor of a viscosity law
// (std::pow of the cell centre: diffusion_coeff.hpp:222-224, :262-264) is tabulated per block with the host libm
inline artemis_diffusion_t Diffusion_(PackCache &c, MeshData<Real> *md) {
  auto pm = md->GetParentPointer();
  auto &pkg = pm->packages.Get("gas");
  artemis_diffusion_t d;
  std::memset(&d, 0, sizeof d);
  if (pkg->template Param<bool>("do_viscosity")) d.visc = Coeff(pkg->template Param<Diffusion::DiffCoeffParams>("visc_params"));
  if (pkg->template Param<bool>("do_conduction")) d.cond = Coeff(pkg->template Param<Diffusion::DiffCoeffParams>("cond_params"));
  d.cv = pkg->template Param<ArtemisUtils::EOS>("eos_h").SpecificHeatFromDensityTemperature(1.0, 1.0);
  const bool radial = d.visc.type == ARTEMIS_VISCOSITY_ALPHA || (d.visc.type == ARTEMIS_VISCOSITY_PLAW && d.visc.r_exp != 0.0);
  if (radial) {
    const artemis_pack_t &p = c.p;
    const int ndim = (p.nx3 > 1) ? 3 : ((p.nx2 > 1) ? 2 : 1);
    const long N = static_cast<long>(p.nx1 + 2 * p.nghost) * (ndim > 1 ? p.nx2 + 2 * p.nghost : 1) * (ndim > 2 ? p.nx3 + 2 * p.nghost : 1);
    if (!c.visc_radial_built) {
      std::vector<Real> h(static_cast<size_t>(N) * p.nblocks);
      for (int b = 0; b < p.nblocks; ++b)
        PARTHENON_REQUIRE(artemis_hip_diffusion_radial_fill(&p, c.geom_host.data(), c.metric_host.empty() ? nullptr : c.metric_host.data(),
                                                            &d.visc, b, h.data() + static_cast<size_t>(b) * N) == 0,
                          artemis_hip_last_error());
      c.visc_radial_data = ParArray1D<Real>("artemis_hip viscosity radial factor", static_cast<int>(h.size()));
      parthenon::deep_copy_from_host(c.visc_radial_data, h.data(), h.size());
      c.visc_radial = ParArray1D<const Real *>("artemis_hip viscosity radial table", p.nblocks);
      auto t = c.visc_radial;
      const Real *base = c.visc_radial_data.data();
      parthenon::par_for(
          DEFAULT_LOOP_PATTERN, "ArtemisHip::FillRadialTable", parthenon::DevExecSpace(), 0, p.nblocks - 1, 0, 0,
          KOKKOS_LAMBDA(const int b, const int) { t(b) = base + static_cast<size_t>(b) * N; });
      c.visc_radial_built = true;
    }
    d.visc.radial = c.visc_radial.data();
  }
  return d;
}
inline TaskStatus GasZeroDiffusionFlux(MeshData<Real> *md) { // Gas::ZeroDiffusionFlux, gas.cpp:522-540
  ARTEMIS_HIP_TASK(artemis_hip_zero_diffusion_flux(&GetPack(md), Stream()));
}
inline TaskStatus GasViscousFlux(MeshData<Real> *md) { // Gas::ViscousFlux<GEOM>, gas.cpp:545-575
  PackCache &c = GetCache(md);
  const artemis_diffusion_t d = Diffusion_(c, md);
  ARTEMIS_HIP_TASK(artemis_hip_viscous_flux(&c.p, &d, Stream()));
}
inline TaskStatus GasThermalFlux(MeshData<Real> *md) { // Gas::ThermalFlux<GEOM>, gas.cpp:580-610
  PackCache &c = GetCache(md);
  const artemis_diffusion_t d = Diffusion_(c, md);
  ARTEMIS_HIP_TASK(artemis_hip_thermal_flux(&c.p, &d, Stream()));
}
inline TaskStatus GasDiffusionUpdate(MeshData<Real> *md, const Real dt) { // Gas::DiffusionUpdate<GEOM>, gas.cpp:615-641
  PackCache &c = GetCache(md);
  const artemis_diffusion_t d = Diffusion_(c, md);
  ARTEMIS_HIP_TASK(artemis_hip_diffusion_update(&c.p, &d, dt, Stream()));
}

// ---- source tasks between FluxSource and SetAuxillaryFields (artemis_driver.cpp:222-248) ---------------------------
// the gravity package's parameters as the library takes them (uniform, point, binary: gravity.cpp:36-118); false for
// the N-body type, whose particle list travels separately
inline bool GravityParams(MeshData<Real> *md, const Real time, artemis_gravity_t &g) {
  auto pm = md->GetParentPointer();
  auto &pkg = pm->packages.Get("gravity");
  const auto gtype = pkg->template Param<Gravity::GravityType>("type");
  std::memset(&g, 0, sizeof g);
  g.tstart = pkg->template Param<Real>("tstart"), g.tstop = pkg->template Param<Real>("tstop");
  if (gtype == Gravity::GravityType::uniform) {
    g.type = ARTEMIS_GRAVITY_UNIFORM;
    g.g[0] = pkg->template Param<Real>("gx1"), g.g[1] = pkg->template Param<Real>("gx2"), g.g[2] = pkg->template Param<Real>("gx3");
  } else if (gtype == Gravity::GravityType::point) {
    g.type = ARTEMIS_GRAVITY_POINT;
    g.gm = pkg->template Param<Real>("gm"), g.soft = pkg->template Param<Real>("soft");
    g.sink = pkg->template Param<Real>("sink"), g.sink_rate = pkg->template Param<Real>("sink_rate");
    g.pos[0] = pkg->template Param<Real>("x"), g.pos[1] = pkg->template Param<Real>("y"), g.pos[2] = pkg->template Param<Real>("z");
  } else if (gtype == Gravity::GravityType::binary) { // binary_mass.cpp:40-70: the orbit is solved on the host per call
    g.type = ARTEMIS_GRAVITY_BINARY;
    g.gm = pkg->template Param<Real>("gm"), g.q = pkg->template Param<Real>("q");
    g.soft = pkg->template Param<Real>("soft1"), g.soft2 = pkg->template Param<Real>("soft2");
    g.sink = pkg->template Param<Real>("sink1"), g.sink2 = pkg->template Param<Real>("sink2");
    g.sink_rate = pkg->template Param<Real>("sink_rate1"), g.sink_rate2 = pkg->template Param<Real>("sink_rate2");
    const Real com[3] = {pkg->template Param<Real>("x"), pkg->template Param<Real>("y"), pkg->template Param<Real>("z")};
    auto orb = pkg->template Param<Gravity::Orbit>("orb");
    Real omf = 0.0;
    if (pm->packages.Get("artemis")->template Param<bool>("do_rotating_frame"))
      omf = pm->packages.Get("rotating_frame")->template Param<Real>("omega");
    Real rb[3], vb[3];
    orb.solve(time, omf, rb, vb);
    const Real mu1 = 1.0 / (1.0 + g.q), mu2 = g.q / (1.0 + g.q);
    for (int d = 0; d < 3; ++d) g.pos[d] = com[d] - mu2 * rb[d], g.pos2[d] = com[d] + mu1 * rb[d];
  } else if (gtype == Gravity::GravityType::nbody) {
    return false;
  } else {
    PARTHENON_FAIL("Unknown gravity node!");
  }
  return true;
}
inline TaskStatus ExternalGravity(MeshData<Real> *md, const Real time, const Real dt) { // gravity/gravity.cpp:126-155
  auto pm = md->GetParentPointer();
  auto &pkg = pm->packages.Get("gravity");
  artemis_gravity_t g;
  if (!GravityParams(md, time, g)) { // gravity.cpp:150-155 -> nbody_gravity.hpp:160-221
    auto &nb = pm->packages.Get("nbody");
    if (nb->template Param<int>("npart") <= 0) return TaskStatus::complete;
    if (!((time >= g.tstart) && (time < g.tstop))) return TaskStatus::complete; // gravity.cpp:134
    Real omf = 0.0;
    if (pm->packages.Get("artemis")->template Param<bool>("do_rotating_frame") && nb->template Param<bool>("frame_correction"))
      omf = pm->packages.Get("rotating_frame")->template Param<Real>("omega");
    auto particles = nb->template Param<ParArray1D<NBody::Particle>>("particles").GetHostMirrorAndCopy();
    const int npart = static_cast<int>(particles.size());
    std::vector<artemis_nbody_particle_t> pl(npart);
    for (int n = 0; n < npart; ++n) {
      const NBody::Particle &q = particles(n);
      artemis_nbody_particle_t &a = pl[n];
      std::memset(&a, 0, sizeof a);
      a.gm = q.GM, a.rs = q.rs, a.racc = q.racc, a.gamma = q.gamma, a.beta = q.beta, a.spline = q.spline, a.couple = q.couple;
      for (int d = 0; d < 3; ++d) a.pos[d] = q.pos[d], a.vel[d] = q.vel[d], a.xf[d] = q.xf[d], a.vf[d] = q.vf[d];
    }
    auto pforce = nb->template Param<parthenon::ParArray2D<Real>>("particle_force");
    auto pforce_h = pforce.GetHostMirrorAndCopy();
    std::vector<double> f(static_cast<size_t>(7) * npart, 0.0);
    PARTHENON_REQUIRE(artemis_hip_nbody_gravity(&GetPack(md), pl.data(), npart, omf, time, dt, f.data(), Stream()) == 0,
                      artemis_hip_last_error());
    for (int n = 0; n < npart; ++n)
      for (int i = 0; i < 7; ++i) pforce_h(n, i) += f[static_cast<size_t>(7) * n + i]; // nbody_gravity.hpp:213-215
    pforce.DeepCopy(pforce_h);
    return TaskStatus::complete;
  }
  (void)pkg;
  ARTEMIS_HIP_TASK(artemis_hip_external_gravity(&GetPack(md), &g, time, dt, Stream()));
}
inline TaskStatus RotatingFrameForce(MeshData<Real> *md, const Real time, const Real dt) { // rotating_frame.cpp:56-86
  auto &pkg = md->GetParentPointer()->packages.Get("rotating_frame");
  ARTEMIS_HIP_TASK(artemis_hip_rotating_frame_force(&GetPack(md), pkg->template Param<Real>("omega"),
                                                    pkg->template Param<Real>("qshear"), time, dt, Stream()));
}
// the drag package's parameters as the library takes them (drag.cpp:35-86); `diff` receives the gas viscosity when
// <gas/damping> damp_to_visc points the damping at it (d.damp_visc then points into diff: keep both alive)
inline void DragParams(PackCache &c, MeshData<Real> *md, artemis_drag_t &d, artemis_diffusion_t &diff) {
  auto pm = md->GetParentPointer();
  auto &pkg = pm->packages.Get("drag");
  std::memset(&d, 0, sizeof d);
  const auto ctype = pkg->template Param<Drag::Coupling>("type");
  d.type = (ctype == Drag::Coupling::simple_dust) ? ARTEMIS_DRAG_SIMPLE_DUST : ARTEMIS_DRAG_SELF;
  auto damping = [](const Drag::SelfDragParams &s) {
    artemis_damping_t o;
    for (int i = 0; i < 3; ++i) o.ix[i] = s.ix[i], o.ox[i] = s.ox[i], o.irate[i] = s.irate[i], o.orate[i] = s.orate[i];
    return o;
  };
  const auto gpar = pkg->template Param<Drag::SelfDragParams>("gas_self_drag");
  d.gas = damping(gpar), d.dust = damping(pkg->template Param<Drag::SelfDragParams>("dust_self_drag"));
  d.xmin[0] = pkg->template Param<Real>("x1min"), d.xmin[1] = pkg->template Param<Real>("x2min"), d.xmin[2] = pkg->template Param<Real>("x3min");
  d.xmax[0] = pkg->template Param<Real>("x1max"), d.xmax[1] = pkg->template Param<Real>("x2max"), d.xmax[2] = pkg->template Param<Real>("x3max");
  d.scale = 1.0, d.grain_density = 1.0;
  if (ctype == Drag::Coupling::simple_dust) {
    const auto sp = pkg->template Param<Drag::StoppingTimeParams>("stopping_time_params");
    auto &dust_pkg = pm->packages.Get("dust");
    const int nd = dust_pkg->template Param<int>("nspecies");
    PARTHENON_REQUIRE(nd <= ARTEMIS_MAX_DUST_SPECIES, "too many dust species for artemis_drag_t");
    d.model = (sp.model == Drag::DragModel::stokes) ? ARTEMIS_DRAG_STOKES : ARTEMIS_DRAG_CONSTANT;
    d.scale = sp.scale;
    auto tau_h = sp.tau.GetHostMirrorAndCopy(); // already scaled for the constant model (drag.hpp:129-137)
    for (int n = 0; n < nd; ++n) d.tau[n] = tau_h(n);
    if (sp.model == Drag::DragModel::stokes) { // drag.hpp:407-409
      d.grain_density = dust_pkg->template Param<Real>("grain_density");
      auto sizes = dust_pkg->template Param<ParArray1D<Real>>("h_sizes");
      for (int n = 0; n < nd; ++n) d.sizes[n] = sizes(n);
    }
  }
  if (gpar.damp_to_visc) { // the gas package's viscosity (drag.cpp:109-121)
    diff = Diffusion_(c, md);
    PARTHENON_REQUIRE(diff.visc.type == ARTEMIS_VISCOSITY_PLAW || diff.visc.type == ARTEMIS_VISCOSITY_ALPHA,
                      "The chosen viscosity model does not work with damping");
    d.damp_visc = &diff.visc;
  }
}
inline TaskStatus DragSource(MeshData<Real> *md, const Real time, const Real dt) { // Drag::DragSource<GEOM>, drag.cpp:89-175
  PackCache &c = GetCache(md);
  artemis_drag_t d;
  artemis_diffusion_t diff;
  DragParams(c, md, d, diff);
  ARTEMIS_HIP_TASK(artemis_hip_drag_source(&c.p, &d, time, dt, Stream()));
}

// ---- time step (StateDescriptor::EstimateTimestepMesh of the gas and dust packages) --------------------------------
inline Real GasEstimateTimestepMesh(MeshData<Real> *md) { // gas.cpp:392-468 (incl. the diffusive limits, :435-467)
  auto &pkg = md->GetParentPointer()->packages.Get("gas");
  const Real cfl = pkg->template Param<Real>("cfl");
  PackCache &c = GetCache(md);
  if (pkg->template Param<bool>("do_diffusion")) { // min(hydro, viscous, conductive) on one device scalar
    const artemis_diffusion_t d = Diffusion_(c, md);
    ParArray1D<Real> dt_dev("artemis_hip dt", 1);
    const Real big = std::numeric_limits<Real>::max();
    parthenon::deep_copy_from_host(dt_dev, &big, 1);
    PARTHENON_REQUIRE(artemis_hip_estimate_dt_async(&c.p, ARTEMIS_GAS, cfl, dt_dev.data(), Stream()) == 0, artemis_hip_last_error());
    PARTHENON_REQUIRE(artemis_hip_diffusion_dt(&c.p, &d, cfl, dt_dev.data(), Stream()) == 0, artemis_hip_last_error());
    return dt_dev.GetHostMirrorAndCopy()(0); // (deep_copy fences the stream)
  }
  double dt = 0.0;
  PARTHENON_REQUIRE(artemis_hip_estimate_dt(&c.p, ARTEMIS_GAS, cfl, &dt, Stream()) == 0, artemis_hip_last_error());
  return dt;
}
inline Real DustEstimateTimestepMesh(MeshData<Real> *md) { // dust.cpp:239-276
  const Real cfl = md->GetParentPointer()->packages.Get("dust")->template Param<Real>("cfl");
  double dt = 0.0;
  PARTHENON_REQUIRE(artemis_hip_estimate_dt(&GetPack(md), ARTEMIS_DUST, cfl, &dt, Stream()) == 0, artemis_hip_last_error());
  return dt;
}

// ---- opt-in fast path ------------------------------------------------------------------------------------------------
// One launch for CalculateFluxes .. ConsToPrim of a stage (artemis_driver.cpp:184-255 with every optional package
// off: gas, one species, Cartesian, PCM / PLM).  A host that takes it replaces the tasks from Gas::CalculateFluxes
// to PreCommFillDerived by this one and FillDerived (artemis_driver.cpp:261) by StageFusedFillDerived: `cons` of u0 is
// current after every stage exactly as with the task chain (the kernel stores the conserved state of the zones it
// updates; only the ghost zones are converted after the boundary exchange).  The kernel reads the primitives at the
// start of the stage AND at the start of the step (it rebuilds u1 from them; the primitives are OneCopy in Artemis,
// gas.cpp:244-270, so the u1 MeshData does not hold them) and must not write where it reads: the adapter keeps two
// primitive buffers of its own per partition -- the start-of-step snapshot and the stage's output -- and moves the
// active zones between them and u0's arrays in ONE device pass per stage (CopyInterior).  (A host that can swap the variables' data pointers instead saves the copies; the repository's own
// driver ping-pongs three buffers.)
// One thread per (table entry, zone): the copies are bandwidth-bound device passes, not serial host-style loops.
// CopyInterior: dst(e)[z] = src(e)[z] over the ACTIVE zones z of every entry (the stage kernel writes no ghost zone,
// so none is copied back); with `keep` the value dst held goes to keep(e)[z] first -- the start-of-step snapshot and
// the copy-back of stage 1 in one pass.  Extents in long: a table of 6 x 260^3 zones has more than 2^31 elements.
struct ZoneBox {
  int nx1, nx2, nx3, g1, g2, g3;
  long ni, nj;
};
inline ZoneBox ActiveZones(const artemis_pack_t &p) {
  ZoneBox z;
  z.nx1 = p.nx1, z.nx2 = p.nx2, z.nx3 = p.nx3;
  z.g1 = p.nghost, z.g2 = (p.nx2 > 1) ? p.nghost : 0, z.g3 = (p.nx3 > 1) ? p.nghost : 0;
  z.ni = p.nx1 + 2 * z.g1, z.nj = p.nx2 + 2 * z.g2;
  return z;
}
template <typename TA, typename TB, typename TC>
inline void CopyInterior(const TA &dst, const TB &src, const TC &keep, const bool with_keep, const int nentries,
                         const ZoneBox zb) {
  const int nact = zb.nx1 * zb.nx2 * zb.nx3; // (< 2^31: one block)
  parthenon::par_for(
      DEFAULT_LOOP_PATTERN, "ArtemisHip::CopyInterior", parthenon::DevExecSpace(), 0, nentries - 1, 0, nact - 1,
      KOKKOS_LAMBDA(const int e, const int q) {
        const int i = q % zb.nx1, j = (q / zb.nx1) % zb.nx2, k = q / (zb.nx1 * zb.nx2);
        const long z = (static_cast<long>(k + zb.g3) * zb.nj + (j + zb.g2)) * zb.ni + (i + zb.g1);
        Real *d = dst(e);
        if (with_keep) keep(e)[z] = d[z];
        d[z] = src(e)[z];
      });
}
struct StageBuffers {
  ParArray1D<Real> step_data, new_data;
  ParArray1D<Real *> step, out;
  const Real *probe = nullptr;
  std::uint64_t sig = 0;
};
inline std::map<int, StageBuffers> &StageBufferCache() {
  static std::map<int, StageBuffers> m;
  return m;
}
inline TaskStatus StageFused(MeshData<Real> *u0, const int stage, const parthenon::LowStorageIntegrator *integ, const bool pcm) {
  PackCache &c = GetCache(u0);
  const artemis_pack_t &p = c.p;
  const int ndim = (p.nx3 > 1) ? 3 : ((p.nx2 > 1) ? 2 : 1);
  const long N = static_cast<long>(p.nx1 + 2 * p.nghost) * (ndim > 1 ? p.nx2 + 2 * p.nghost : 1) * (ndim > 2 ? p.nx3 + 2 * p.nghost : 1);
  const int nent = p.nblocks * 6 * p.gas.nspecies;
  StageBuffers &sb = StageBufferCache()[u0->GetPartitionId()];
  if (sb.probe != c.probe0 || sb.sig != c.sig0 || sb.step.size() != nent) {
    sb.step_data = ParArray1D<Real>("artemis_hip start-of-step primitives", static_cast<long>(nent) * N);
    sb.new_data = ParArray1D<Real>("artemis_hip stage output primitives", static_cast<long>(nent) * N);
    sb.step = ParArray1D<Real *>("artemis_hip table", nent), sb.out = ParArray1D<Real *>("artemis_hip table", nent);
    auto ts = sb.step, to = sb.out;
    Real *bs = sb.step_data.data(), *bo = sb.new_data.data();
    parthenon::par_for(
        DEFAULT_LOOP_PATTERN, "ArtemisHip::FillStageTables", parthenon::DevExecSpace(), 0, nent - 1, 0, 0,
        KOKKOS_LAMBDA(const int e, const int) { ts(e) = bs + e * N, to(e) = bo + e * N; });
    sb.probe = c.probe0, sb.sig = c.sig0;
  }
  artemis_stage_args_t a;
  std::memset(&a, 0, sizeof a);
  a.gam0 = integ->gam0[stage - 1], a.gam1 = integ->gam1[stage - 1];
  a.beta_dt = a.bdt = integ->beta[stage - 1] * integ->dt;
  a.pcm = pcm;
  a.prim_in = p.gas.prim, a.prim_u1 = (stage == 1) ? p.gas.prim : sb.step.data(), a.prim_out = sb.out.data();
  a.cons_out = p.gas.cons0;
  const int rc = artemis_hip_stage_fused(&p, &a, Stream());
  PARTHENON_REQUIRE(rc == 0, artemis_hip_last_error());
  // new primitives into u0's arrays (active zones; the boundary exchange that follows fills the ghost zones); in
  // stage 1 the values they replace are the start-of-step primitives the later stages rebuild u1 from
  CopyInterior(c.gprim, sb.out, sb.step, stage == 1, nent, ActiveZones(p));
  return TaskStatus::complete;
}
inline void StageFusedFillDerived(MeshData<Real> *md) {
  PARTHENON_REQUIRE(artemis_hip_prim_to_cons_ghosts(&GetPack(md), Stream()) == 0, artemis_hip_last_error());
}

// ---- the DEFAULT stage task: CalculateFluxes .. ConsToPrim (artemis_driver.cpp:182-255) as ONE task -----------------
// What a maintainer wires into StepTasks: per stage
//     if (ArtemisHip::StageCovered(u0)) { Stage(u0, stage, integrator, pcm, time); <boundary exchange>; StageFillDerived(u0); }
//     else                              { the per-task forwarders above, in the reference's order }
// Stage takes the tuned Cartesian kernel (StageFused above: it also stores the conserved state, so only the ghost zones
// are converted afterwards) where that applies, and artemis_hip_stage_general for everything else the one-kernel
// stages cover: gas and / or dust, any coordinate system, PCM / PLM / PPM, viscosity and conduction (their flux tasks
// run inside Stage), uniform / point / binary gravity, the rotating frame, drag.  Not covered (StageCovered is false,
// the per-task list stays): the N-body gravity type (its particle list lives with the NBody package and its
// back-reaction sums go to the host: see artemis_stage_general_args_t.nbody_dev for a host that keeps the particles
// on the device), cooling (this header forwards no CoolingSource: artemis_stage_general_args_t.cooling exists for a
// host that fills artemis_cooling_t), radiation -- and REFINED MESHES (pmesh->multilevel): the one-kernel stages keep
// no flux arrays, so the block SendBoundBufs<flxcor_send> .. SetFluxCorrections (artemis_driver.cpp:196-202) that this
// task would replace has nothing to send and conservation at coarse-fine faces would be lost silently.  A refined mesh
// keeps the per-task list, whose CalculateFluxes forwarders fill Parthenon's flux arrays (or the host wires the
// face-flux / fix-up sequence of INTEGRATION.md section 3a itself).
inline bool StageCovered(MeshData<Real> *md) {
  auto pm = md->GetParentPointer();
  if (pm->multilevel) return false;
  auto &art = pm->packages.Get("artemis");
  auto flag = [&](const char *n) { return art->template Param<bool>(n); }; // (artemis.cpp:73-83 registers every do_* flag)
  if (flag("do_radiation") || flag("do_cooling")) return false;
  if (flag("do_gravity") && pm->packages.Get("gravity")->template Param<Gravity::GravityType>("type") == Gravity::GravityType::nbody)
    return false;
  return true;
}
inline bool StageTakesTunedKernel(MeshData<Real> *md) {
  auto pm = md->GetParentPointer();
  auto &art = pm->packages.Get("artemis");
  auto flag = [&](const char *n) { return art->template Param<bool>(n); }; // (artemis.cpp:73-83 registers every do_* flag)
  const artemis_pack_t &p = GetCache(md).p;
  return p.coords == ARTEMIS_CARTESIAN && p.gas.nspecies == 1 && p.dust.nspecies == 0 && p.gas.recon != ARTEMIS_PPM &&
         !flag("do_gravity") && !flag("do_rotating_frame") && !flag("do_drag") && !flag("do_cooling") && !flag("do_viscosity") &&
         !flag("do_conduction");
}
struct GeneralBuffers {
  StageBuffers gas, dust;
};
inline std::map<int, GeneralBuffers> &GeneralBufferCache() {
  static std::map<int, GeneralBuffers> m;
  return m;
}
inline void EnsureStageBuffers(StageBuffers &sb, const PackCache &c, const int nent, const long N) {
  if (sb.probe == c.probe0 && sb.sig == c.sig0 && sb.step.size() == nent) return;
  sb.step_data = ParArray1D<Real>("artemis_hip start-of-step primitives", static_cast<long>(nent) * N);
  sb.new_data = ParArray1D<Real>("artemis_hip stage output primitives", static_cast<long>(nent) * N);
  sb.step = ParArray1D<Real *>("artemis_hip table", nent), sb.out = ParArray1D<Real *>("artemis_hip table", nent);
  auto ts = sb.step, to = sb.out;
  Real *bs = sb.step_data.data(), *bo = sb.new_data.data();
  parthenon::par_for(
      DEFAULT_LOOP_PATTERN, "ArtemisHip::FillStageTables", parthenon::DevExecSpace(), 0, nent - 1, 0, 0,
      KOKKOS_LAMBDA(const int e, const int) { ts(e) = bs + e * N, to(e) = bo + e * N; });
  sb.probe = c.probe0, sb.sig = c.sig0;
}
inline TaskStatus Stage(MeshData<Real> *u0, const int stage, const parthenon::LowStorageIntegrator *integ, const bool pcm,
                        const Real time) {
  PARTHENON_REQUIRE(!u0->GetParentPointer()->multilevel,
                    "ArtemisHip::Stage stores no flux arrays: a refined mesh keeps the per-task list (StageCovered) so that "
                    "the flux correction of artemis_driver.cpp:196-202 has fluxes to send");
  if (StageTakesTunedKernel(u0)) return StageFused(u0, stage, integ, pcm);
  auto pm = u0->GetParentPointer();
  auto &art = pm->packages.Get("artemis");
  auto flag = [&](const char *n) { return art->template Param<bool>(n); }; // (artemis.cpp:73-83 registers every do_* flag)
  PackCache &c = GetCache(u0);
  const artemis_pack_t &p = c.p;
  const int ndim = (p.nx3 > 1) ? 3 : ((p.nx2 > 1) ? 2 : 1);
  const long N = static_cast<long>(p.nx1 + 2 * p.nghost) * (ndim > 1 ? p.nx2 + 2 * p.nghost : 1) * (ndim > 2 ? p.nx3 + 2 * p.nghost : 1);
  const int ng = p.nblocks * 6 * p.gas.nspecies, nd = p.nblocks * 4 * p.dust.nspecies;
  GeneralBuffers &gb = GeneralBufferCache()[u0->GetPartitionId()];
  if (ng) EnsureStageBuffers(gb.gas, c, ng, N);
  if (nd) EnsureStageBuffers(gb.dust, c, nd, N);
  artemis_stage_general_args_t a;
  std::memset(&a, 0, sizeof a);
  a.gam0 = integ->gam0[stage - 1], a.gam1 = integ->gam1[stage - 1];
  a.beta_dt = a.bdt = integ->beta[stage - 1] * integ->dt;
  a.pcm = pcm, a.time = time;
  if (ng) a.gas_in = p.gas.prim, a.gas_u1 = (stage == 1) ? p.gas.prim : gb.gas.step.data(), a.gas_out = gb.gas.out.data();
  if (nd) a.dust_in = p.dust.prim, a.dust_u1 = (stage == 1) ? p.dust.prim : gb.dust.step.data(), a.dust_out = gb.dust.out.data();
  artemis_gravity_t grav;
  if (flag("do_gravity")) {
    PARTHENON_REQUIRE(GravityParams(u0, time, grav), "ArtemisHip::Stage: N-body gravity keeps the per-task list (StageCovered)");
    a.gravity = &grav;
  }
  if (flag("do_rotating_frame")) {
    auto &rf = pm->packages.Get("rotating_frame");
    a.rf_omega = rf->template Param<Real>("omega"), a.rf_qshear = rf->template Param<Real>("qshear");
  }
  artemis_drag_t drag;
  artemis_diffusion_t drag_visc, diff;
  if (flag("do_drag")) DragParams(c, u0, drag, drag_visc), a.drag = &drag;
  if (ng && (flag("do_viscosity") || flag("do_conduction"))) { // the diffusion-flux tasks of this stage's input (artemis_driver.cpp:189-193)
    diff = Diffusion_(c, u0);
    PARTHENON_REQUIRE(artemis_hip_zero_viscous_flux(&p, &diff, Stream()) == 0, artemis_hip_last_error());
    if (diff.cond.type != ARTEMIS_DIFF_OFF) PARTHENON_REQUIRE(artemis_hip_thermal_flux(&p, &diff, Stream()) == 0, artemis_hip_last_error());
    a.diffusion = &diff;
  }
  PARTHENON_REQUIRE(artemis_hip_stage_general(&p, &a, Stream()) == 0, artemis_hip_last_error());
  if (ng) CopyInterior(c.gprim, gb.gas.out, gb.gas.step, stage == 1, ng, ActiveZones(p));
  if (nd) CopyInterior(c.dprim, gb.dust.out, gb.dust.step, stage == 1, nd, ActiveZones(p));
  return TaskStatus::complete;
}
// FillDerived after the boundary exchange (artemis_driver.cpp:261): the tuned kernel stored the conserved state of the
// zones it updated, so only the ghost zones are converted; the general stage stores primitives only
inline void StageFillDerived(MeshData<Real> *md) {
  if (StageTakesTunedKernel(md)) StageFusedFillDerived(md);
  else PrimToCons(md);
}
#undef ARTEMIS_HIP_TASK

} // namespace ArtemisHip
#endif // ARTEMIS_HIP_ADAPTER_HPP_
