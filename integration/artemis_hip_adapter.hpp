// artemis_hip_adapter.hpp -- the file a maintainer adds to lanl/artemis (src/utils/) to route the hydro tasks
// through libartemis_hip.so.  It fills an `artemis_pack_t` (include/artemis_hip.h) from the SparsePacks the
// reference's own task functions build, once per MeshData partition (again after a remesh), and forwards each
// task.  Nothing else in Artemis changes: StateDescriptor registration, the TaskList of
// ArtemisDriver<GEOM>::StepTasks (artemis_driver.cpp:145-273), problem generators and decks stay as they are.
//
// Compile-checked in this repository against a declarations-only stand-in for the few Parthenon / Artemis names
// it touches (tests/mock_parthenon/, tests/test_integration_adapter.py); in Artemis it includes the real headers.
#ifndef ARTEMIS_HIP_ADAPTER_HPP_
#define ARTEMIS_HIP_ADAPTER_HPP_

#include <vector>

#include "artemis.hpp"     // Real, Coordinates, RSolver, ReconstructionMethod, the gas:: / dust:: field types
#include "artemis_hip.h"   // this repository's include/

namespace ArtemisHip {
using parthenon::MeshData;
using parthenon::ParArray1D;
using parthenon::TaskStatus;
using TE = parthenon::TopologicalElement;

// Device tables of one fluid + what is shared by the pack.  One instance per MeshData partition.
struct PackCache {
  artemis_pack_t p{};
  ParArray1D<Real *> gprim, gcons0, gcons1, gflux[3], gpflux[3], gvface[3], gdflux[3];
  ParArray1D<Real *> dprim, dcons0, dcons1, dflux[3];
  ParArray1D<Real> geom, metric;
  bool built = false;
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
// face field (gas.face.velocity) on the faces of direction dir
template <typename Pack>
void FillFaceTable(ParArray1D<Real *> &tab, const Pack &v, const int nb, const int nvar, const int dir) {
  tab = ParArray1D<Real *>("artemis_hip face table", nb * nvar);
  auto t = tab;
  const TE te = (dir == 1) ? TE::F1 : ((dir == 2) ? TE::F2 : TE::F3);
  parthenon::par_for(
      DEFAULT_LOOP_PATTERN, "ArtemisHip::FillFaceTable", parthenon::DevExecSpace(), 0, nb - 1, 0, nvar - 1,
      KOKKOS_LAMBDA(const int b, const int n) { t(b * nvar + n) = &v(b, te, n, 0, 0, 0); });
}

// u0 = the MeshData the task receives, u1 = the start-of-step copy (artemis_driver.cpp:137-139); may be the same
// object for tasks that do not read cons1.
inline artemis_pack_t &GetPack(PackCache &c, MeshData<Real> *u0, MeshData<Real> *u1) {
  if (c.built) return c.p;
  auto pm = u0->GetParentPointer();
  auto &artemis_pkg = pm->packages.Get("artemis");
  const bool do_gas = artemis_pkg->template Param<bool>("do_gas");
  const bool do_dust = artemis_pkg->template Param<bool>("do_dust");
  const bool do_diffusion = do_gas && (artemis_pkg->template Param<bool>("do_viscosity") ||
                                       artemis_pkg->template Param<bool>("do_conduction"));
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
  using parthenon::MakePackDescriptor;
  using parthenon::PDOpt;

  if (do_gas) {
    auto &pkg = pm->packages.Get("gas");
    const int ns = pkg->template Param<int>("nspecies");
    p.gm1 = pkg->template Param<Real>("adiabatic_index") - 1.0;
    p.gas.nspecies = ns;
    p.gas.recon = static_cast<int>(pkg->template Param<ReconstructionMethod>("recon"));
    p.gas.riemann = static_cast<int>(pkg->template Param<RSolver>("rsolver"));
    p.gas.dfloor = pkg->template Param<Real>("dfloor"), p.gas.siefloor = pkg->template Param<Real>("siefloor");
    p.gas.de_switch = pkg->template Param<Real>("de_switch");
    // the descriptors Gas::CalculateFluxes builds (gas.cpp:479-488): pack order rho[n], v[ns+3n+d], P[4ns+n], sie[5ns+n]
    static auto dprim = MakePackDescriptor<gas::prim::density, gas::prim::velocity, gas::prim::pressure, gas::prim::sie>(
        res, {}, {PDOpt::WithFluxes});
    static auto dcons = MakePackDescriptor<gas::cons::density, gas::cons::momentum, gas::cons::total_energy,
                                           gas::cons::internal_energy>(res, {}, {PDOpt::WithFluxes});
    static auto dface = MakePackDescriptor<gas::face::velocity>(res);
    auto vprim = dprim.GetPack(u0);
    auto vcons0 = dcons.GetPack(u0), vcons1 = dcons.GetPack(u1);
    auto vface = dface.GetPack(u0);
    FillTable(c.gprim, vprim, nb, 6 * ns, 0), p.gas.prim = c.gprim.data();
    FillTable(c.gcons0, vcons0, nb, 6 * ns, 0), p.gas.cons0 = c.gcons0.data();
    FillTable(c.gcons1, vcons1, nb, 6 * ns, 0), p.gas.cons1 = c.gcons1.data();
    for (int d = 0; d < ndim; ++d) {
      FillFluxTable(c.gflux[d], vcons0, nb, 6 * ns, 0, d + 1), p.gas.flux[d] = c.gflux[d].data();
      // interface pressure = the flux slot of gas.prim.pressure (hllc.hpp:166): pack index 4 ns + n
      FillFluxTable(c.gpflux[d], vprim, nb, ns, 4 * ns, d + 1), p.gas.pflux[d] = c.gpflux[d].data();
      FillFaceTable(c.gvface[d], vface, nb, ns, d + 1), p.gas.vface[d] = c.gvface[d].data(); // hllc.hpp:179
    }
    if (do_diffusion) { // gas::diff::momentum (3 n + component) then gas::diff::energy (3 ns + n), gas.cpp:276-284
      static auto ddiff = MakePackDescriptor<gas::diff::momentum, gas::diff::energy>(res);
      auto vdiff = ddiff.GetPack(u0);
      for (int d = 0; d < ndim; ++d) {
        c.gdflux[d] = ParArray1D<Real *>("artemis_hip diffusion flux table", nb * 4 * ns);
        auto t = c.gdflux[d];
        const TE te = (d == 0) ? TE::F1 : ((d == 1) ? TE::F2 : TE::F3);
        parthenon::par_for(
            DEFAULT_LOOP_PATTERN, "ArtemisHip::FillDiffTable", parthenon::DevExecSpace(), 0, nb - 1, 0, 4 * ns - 1,
            KOKKOS_LAMBDA(const int b, const int n) { t(b * 4 * ns + n) = &vdiff(b, te, n, 0, 0, 0); });
        p.gas.diff_flux[d] = c.gdflux[d].data();
      }
    }
  }
  if (do_dust) {
    auto &pkg = pm->packages.Get("dust");
    const int ns = pkg->template Param<int>("nspecies");
    p.dust.nspecies = ns;
    p.dust.recon = static_cast<int>(pkg->template Param<ReconstructionMethod>("recon"));
    p.dust.riemann = static_cast<int>(pkg->template Param<RSolver>("rsolver"));
    p.dust.dfloor = pkg->template Param<Real>("dfloor");
    static auto dprim = MakePackDescriptor<dust::prim::density, dust::prim::velocity>(res);            // dust.cpp:287-290
    static auto dcons = MakePackDescriptor<dust::cons::density, dust::cons::momentum>(res, {}, {PDOpt::WithFluxes});
    auto vprim = dprim.GetPack(u0);
    auto vcons0 = dcons.GetPack(u0), vcons1 = dcons.GetPack(u1);
    FillTable(c.dprim, vprim, nb, 4 * ns, 0), p.dust.prim = c.dprim.data();
    FillTable(c.dcons0, vcons0, nb, 4 * ns, 0), p.dust.cons0 = c.dcons0.data();
    FillTable(c.dcons1, vcons1, nb, 4 * ns, 0), p.dust.cons1 = c.dcons1.data();
    for (int d = 0; d < ndim; ++d) FillFluxTable(c.dflux[d], vcons0, nb, 4 * ns, 0, d + 1), p.dust.flux[d] = c.dflux[d].data();
  }
  if (artemis_pkg->template Param<bool>("do_rotating_frame")) // fluid_fluxes.hpp:433-437
    p.omega_frame = pm->packages.Get("rotating_frame")->template Param<Real>("omega");

  // edge table: Coordinates_t::Xf<d>(idx) = xf0 + idx * dx with idx counted from the first ghost zone
  // (geometry.hpp:65-72); host copy kept for the metric tables
  std::vector<Real> geom_host(6 * nb);
  for (int b = 0; b < nb; ++b) {
    const auto &pco = u0->GetBlockData(b)->GetBlockPointer()->coords;
    geom_host[6 * b + 0] = pco.template Xf<parthenon::X1DIR>(0), geom_host[6 * b + 1] = pco.template Dxf<parthenon::X1DIR>();
    geom_host[6 * b + 2] = pco.template Xf<parthenon::X2DIR>(0), geom_host[6 * b + 3] = pco.template Dxf<parthenon::X2DIR>();
    geom_host[6 * b + 4] = pco.template Xf<parthenon::X3DIR>(0), geom_host[6 * b + 5] = pco.template Dxf<parthenon::X3DIR>();
  }
  c.geom = ParArray1D<Real>("artemis_hip geom", 6 * nb);
  parthenon::deep_copy_from_host(c.geom, geom_host.data(), geom_host.size());
  p.geom = c.geom.data();
  // the trigonometry Coords<GEOM> evaluates per cell (spherical.hpp:53-146, ConvertCoordsToCart of every system):
  // tabulated once per remesh with the host libm (count is 0 for Cartesian and spherical1D)
  const long nm = artemis_hip_metric_count(&p);
  if (nm > 0) {
    std::vector<Real> m(nm);
    PARTHENON_REQUIRE(artemis_hip_metric_fill(&p, geom_host.data(), m.data()) == 0, artemis_hip_last_error());
    c.metric = ParArray1D<Real>("artemis_hip metric", nm);
    parthenon::deep_copy_from_host(c.metric, m.data(), m.size());
    p.metric = c.metric.data();
  }
  c.built = true;
  return p;
}

// ---- the task bodies (the reference's signatures; artemis_driver.cpp:184-255) -------------------------------------
inline PackCache &Cache(MeshData<Real> *md) {
  static std::vector<PackCache> caches(64); // one per partition; clear `built` on remesh
  return caches[md->GetPartitionId() % 64];
}
inline void *Stream() { return nullptr; } // or Kokkos::HIP().hip_stream(): the library shares the process's HIP runtime

#define ARTEMIS_HIP_TASK(call)                                        \
  do {                                                                \
    const int rc_ = (call);                                           \
    PARTHENON_REQUIRE(rc_ == 0, artemis_hip_last_error());            \
    return TaskStatus::complete;                                      \
  } while (0)

inline TaskStatus GasCalculateFluxes(MeshData<Real> *md, const bool pcm) { // Gas::CalculateFluxes, gas.cpp:473-494
  ARTEMIS_HIP_TASK(artemis_hip_calculate_fluxes(&GetPack(Cache(md), md, md), ARTEMIS_GAS, pcm, Stream()));
}
inline TaskStatus DustCalculateFluxes(MeshData<Real> *md, const bool pcm) { // Dust::CalculateFluxes, dust.cpp:281-298
  ARTEMIS_HIP_TASK(artemis_hip_calculate_fluxes(&GetPack(Cache(md), md, md), ARTEMIS_DUST, pcm, Stream()));
}
inline TaskStatus ApplyUpdate(MeshData<Real> *u0, MeshData<Real> *u1, const int stage,
                              const parthenon::LowStorageIntegrator *integ) { // artemis_integrator.hpp:57-110
  const Real g0 = integ->gam0[stage - 1], g1 = integ->gam1[stage - 1], bdt = integ->beta[stage - 1] * integ->dt;
  ARTEMIS_HIP_TASK(artemis_hip_apply_update(&GetPack(Cache(u0), u0, u1), g0, g1, bdt, Stream()));
}
inline TaskStatus GasFluxSource(MeshData<Real> *md, const Real dt) { // Gas::FluxSource, gas.cpp:499-519
  ARTEMIS_HIP_TASK(artemis_hip_flux_source(&GetPack(Cache(md), md, md), ARTEMIS_GAS, dt, Stream()));
}
inline TaskStatus SetAuxillaryFields(MeshData<Real> *md) { // fill_derived.cpp:30-75
  ARTEMIS_HIP_TASK(artemis_hip_set_aux(&GetPack(Cache(md), md, md), Stream()));
}
inline void ConsToPrim(MeshData<Real> *md) { // PreCommFillDerivedMesh, artemis.cpp:122
  PARTHENON_REQUIRE(artemis_hip_cons_to_prim(&GetPack(Cache(md), md, md), Stream()) == 0, artemis_hip_last_error());
}
inline void PrimToCons(MeshData<Real> *md) { // PreFillDerivedMesh, artemis.cpp:123
  PARTHENON_REQUIRE(artemis_hip_prim_to_cons(&GetPack(Cache(md), md, md), Stream()) == 0, artemis_hip_last_error());
}
inline Real GasEstimateTimestepMesh(MeshData<Real> *md) { // gas.cpp:392-468
  double dt = 0.0;
  const Real cfl = md->GetParentPointer()->packages.Get("gas")->template Param<Real>("cfl");
  PARTHENON_REQUIRE(artemis_hip_estimate_dt(&GetPack(Cache(md), md, md), ARTEMIS_GAS, cfl, &dt, Stream()) == 0,
                    artemis_hip_last_error());
  return dt;
}
#undef ARTEMIS_HIP_TASK

} // namespace ArtemisHip
#endif // ARTEMIS_HIP_ADAPTER_HPP_
