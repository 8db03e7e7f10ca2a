// TEST STAND-IN, tests only.  The handful of Parthenon / Artemis names that integration/artemis_hip_adapter.hpp
// touches, with just enough BEHAVIOUR -- host arrays behind the SparsePack, a Params map, par_for as plain loops --
// that (i) `g++ -fsyntax-only` checks the adapter text against the C ABI of include/artemis_hip.h and (ii)
// tests/adapter_live/run_stage.cpp can run the reference's task list (artemis_driver.cpp:145-273) through the
// adapter on the CPU test double and be compared with the oracle.  It is NOT Parthenon, is not used to build or run
// anything of the reference, and is never shipped; in Artemis the adapter includes the real artemis.hpp
// (src/artemis.hpp:18-105).  Written from the call sites in the reference (gas.cpp:473-494, fluid_fluxes.hpp:78-213,
// artemis_driver.cpp:126-139); no upstream source was available to copy.
#pragma once
#include <any>
#include <cstddef>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

typedef double Real;
#define KOKKOS_LAMBDA [=]
#define DEFAULT_LOOP_PATTERN 0
#define PARTHENON_REQUIRE(cond, msg)                          \
  do {                                                        \
    if (!(cond)) throw std::runtime_error(std::string(msg)); \
  } while (0)
#define PARTHENON_FAIL(msg) throw std::runtime_error(std::string(msg))

enum class Coordinates { cartesian, cylindrical, spherical1D, spherical2D, spherical3D, axisymmetric }; // artemis.hpp:78-86
enum class RSolver { hllc, hlle, llf };                                                                 // :88
enum class ReconstructionMethod { pcm, plm, ppm };                                                      // :89-90

namespace parthenon {
enum CoordinateDirection { NODIR = 0, X1DIR = 1, X2DIR = 2, X3DIR = 3 };
enum class TopologicalElement { CC, F1, F2, F3 };
enum class IndexDomain { interior, entire };
enum class TaskStatus { complete, incomplete };
enum class PDOpt { WithFluxes };
struct IndexRange {
  int s, e;
};
namespace Globals {
extern int nghost;
}
inline int DevExecSpace() { return 0; }

template <class T>
struct ParArray1D { // a reference-counted 1-D array (host memory here)
  std::shared_ptr<std::vector<T>> v;
  ParArray1D() = default;
  ParArray1D(const std::string &, long n) : v(std::make_shared<std::vector<T>>(static_cast<size_t>(n))) {}
  T &operator()(int i) const { return (*v)[static_cast<size_t>(i)]; }
  T *data() const { return v ? v->data() : nullptr; }
  int size() const { return v ? static_cast<int>(v->size()) : 0; }
  ParArray1D GetHostMirrorAndCopy() const { return *this; }
};
template <class T>
void deep_copy_from_host(ParArray1D<T> &a, const T *src, size_t n) {
  std::memcpy(a.data(), src, n * sizeof(T));
}
template <class F>
inline void par_for(int, const char *, int, int b0, int b1, int n0, int n1, F f) {
  for (int b = b0; b <= b1; ++b)
    for (int n = n0; n <= n1; ++n) f(b, n);
}

struct StateDescriptor { // a package: only its Params are used
  std::map<std::string, std::any> params;
  template <class T>
  const T &Param(const std::string &k) const {
    auto it = params.find(k);
    if (it == params.end()) throw std::runtime_error("no such param: " + k);
    return *std::any_cast<T>(&it->second);
  }
  template <class T>
  void AddParam(const std::string &k, T v) {
    params[k] = std::move(v);
  }
};
struct Packages {
  std::map<std::string, std::shared_ptr<StateDescriptor>> m;
  std::shared_ptr<StateDescriptor> &Get(const std::string &k) {
    auto it = m.find(k);
    if (it == m.end()) throw std::runtime_error("no such package: " + k);
    return it->second;
  }
};
struct ResolvedPackages {};
struct Coordinates_t { // uniform logically-Cartesian block: Xf(idx) = xf0 + idx * dx, idx from the first ghost zone
  Real xf0[3] = {0, 0, 0}, dx[3] = {1, 1, 1};
  template <CoordinateDirection D>
  Real Xf(int i) const {
    return xf0[D - 1] + i * dx[D - 1];
  }
  template <CoordinateDirection D>
  Real Dxf() const {
    return dx[D - 1];
  }
};
struct LogicalLocation { // position of a block in the refinement tree
  int lev = 0;
  long l1 = 0, l2 = 0, l3 = 0;
  int level() const { return lev; }
  long lx1() const { return l1; }
  long lx2() const { return l2; }
  long lx3() const { return l3; }
};
struct MeshBlock {
  Coordinates_t coords;
  int gid = 0;
  LogicalLocation loc;
};
struct Mesh {
  Packages packages;
  std::shared_ptr<ResolvedPackages> resolved_packages = std::make_shared<ResolvedPackages>();
  int remesh_count = 0;
  bool multilevel = false; // parthenon::Mesh::multilevel: static or adaptive refinement is on
};
// one variable of one block: ncomp cell arrays [nk][nj][ni] (+ flux slots per direction / face copies)
struct Variable {
  int ncomp = 0;
  size_t N = 0, sj = 0, sk = 0;
  std::vector<Real> data, flux[3], face[3];
};
template <class T>
struct MeshBlockData {
  MeshBlock *pmb = nullptr;
  std::map<std::string, std::shared_ptr<Variable>> vars;
  MeshBlock *GetBlockPointer() const { return pmb; }
};
template <class T>
struct MeshData {
  Mesh *pm = nullptr;
  int partition = 0;
  IndexRange ib{0, 0}, jb{0, 0}, kb{0, 0};
  mutable std::vector<std::shared_ptr<MeshBlockData<T>>> blocks;
  Mesh *GetParentPointer() const { return pm; }
  int NumBlocks() const { return static_cast<int>(blocks.size()); }
  int GetPartitionId() const { return partition; }
  IndexRange GetBoundsI(IndexDomain) const { return ib; }
  IndexRange GetBoundsJ(IndexDomain) const { return jb; }
  IndexRange GetBoundsK(IndexDomain) const { return kb; }
  std::shared_ptr<MeshBlockData<T>> &GetBlockData(int b) const { return blocks[static_cast<size_t>(b)]; }
};
// pack over the listed field types in the order given: all components of the first type, then the second, ...
struct SparsePackStandIn {
  struct Ref {
    Variable *v;
    int comp;
  };
  std::vector<std::vector<Ref>> refs; // [block][pack index]
  Real &operator()(int b, int n, int k, int j, int i) const {
    const Ref &r = refs[b][n];
    return r.v->data[r.comp * r.v->N + k * r.v->sk + j * r.v->sj + i];
  }
  Real &operator()(int b, TopologicalElement te, int n, int k, int j, int i) const {
    const Ref &r = refs[b][n];
    const int d = (te == TopologicalElement::F1) ? 0 : ((te == TopologicalElement::F2) ? 1 : 2);
    return r.v->face[d][r.comp * r.v->N + k * r.v->sk + j * r.v->sj + i];
  }
  Real &flux(int b, int dir, int n, int k, int j, int i) const {
    const Ref &r = refs[b][n];
    return r.v->flux[dir - 1][r.comp * r.v->N + k * r.v->sk + j * r.v->sj + i];
  }
};
struct PackDescriptorStandIn {
  std::vector<std::string> names;
  template <class MD>
  SparsePackStandIn GetPack(MD *md) const {
    SparsePackStandIn p;
    for (auto &blk : md->blocks) {
      std::vector<SparsePackStandIn::Ref> r;
      for (const std::string &nm : names) {
        auto it = blk->vars.find(nm);
        if (it == blk->vars.end()) continue; // (sparse: absent fields contribute nothing)
        for (int c = 0; c < it->second->ncomp; ++c) r.push_back({it->second.get(), c});
      }
      p.refs.push_back(std::move(r));
    }
    return p;
  }
};
template <class... Ts>
PackDescriptorStandIn MakePackDescriptor(ResolvedPackages *, const std::vector<int> & = {}, const std::vector<PDOpt> & = {}) {
  PackDescriptorStandIn d;
  (d.names.push_back(Ts::name()), ...);
  return d;
}
struct LowStorageIntegrator {
  Real dt = 0.0;
  int nstages = 0;
  std::vector<Real> gam0, gam1, beta;
};
} // namespace parthenon

namespace ArtemisUtils {
struct EOS { // singularity::IdealGas(gm1, cv) as far as the adapter asks (gas.cpp:116-119)
  Real gm1 = 0.4, cv = 1.0;
  Real SpecificHeatFromDensityTemperature(Real, Real) const { return cv; }
};
} // namespace ArtemisUtils

// field types (artemis.hpp:39-76 declares them with Parthenon's SPARSE_VARIABLE macros)
#define MOCK_FIELD(ns1, ns2, nm)                                                       \
  namespace ns1 {                                                                      \
  namespace ns2 {                                                                      \
  struct nm {                                                                          \
    static std::string name() { return #ns1 "." #ns2 "." #nm; }                        \
  };                                                                                   \
  }                                                                                    \
  }
MOCK_FIELD(gas, prim, density)
MOCK_FIELD(gas, prim, velocity)
MOCK_FIELD(gas, prim, pressure)
MOCK_FIELD(gas, prim, sie)
MOCK_FIELD(gas, cons, density)
MOCK_FIELD(gas, cons, momentum)
MOCK_FIELD(gas, cons, total_energy)
MOCK_FIELD(gas, cons, internal_energy)
MOCK_FIELD(gas, face, velocity)
MOCK_FIELD(gas, diff, momentum)
MOCK_FIELD(gas, diff, energy)
MOCK_FIELD(dust, prim, density)
MOCK_FIELD(dust, prim, velocity)
MOCK_FIELD(dust, cons, density)
MOCK_FIELD(dust, cons, momentum)
#undef MOCK_FIELD
