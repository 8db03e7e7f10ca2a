// COMPILE-CHECK STAND-IN, tests only.  Declares -- with no behaviour -- the handful of Parthenon / Artemis names
// that integration/artemis_hip_adapter.hpp touches, so that `g++ -fsyntax-only` can check the adapter text is
// complete, well-formed C++ against the C ABI of include/artemis_hip.h.  It is NOT Parthenon, is not used to
// build or run anything of the reference, and is never shipped; in Artemis the adapter includes the real
// artemis.hpp (src/artemis.hpp:18-105) instead.
#pragma once
#include <cstddef>
#include <map>
#include <memory>
#include <string>
#include <vector>

typedef double Real;
#define KOKKOS_LAMBDA [=]
#define DEFAULT_LOOP_PATTERN 0
#define PARTHENON_REQUIRE(cond, msg) \
  do {                               \
    if (!(cond)) throw std::string(msg); \
  } while (0)

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
struct ParArray1D {
  ParArray1D() = default;
  ParArray1D(const std::string &, int) {}
  T &operator()(int) const;
  T *data() const;
};
template <class T>
void deep_copy_from_host(ParArray1D<T> &, const T *, size_t);
template <class F>
inline void par_for(int, const char *, int, int, int, int, int, F) {} // (a body only because lambdas have local types)
struct Params {
  template <class T>
  const T &Get(const std::string &) const;
};
struct StateDescriptor {
  template <class T>
  const T &Param(const std::string &) const;
};
struct Packages {
  std::shared_ptr<StateDescriptor> &Get(const std::string &);
};
struct ResolvedPackages {};
struct Coordinates_t {
  template <CoordinateDirection D>
  Real Xf(int) const;
  template <CoordinateDirection D>
  Real Dxf() const;
};
struct MeshBlock {
  Coordinates_t coords;
};
struct Mesh {
  Packages packages;
  std::shared_ptr<ResolvedPackages> resolved_packages;
};
template <class T>
struct MeshBlockData {
  MeshBlock *GetBlockPointer() const;
};
template <class T>
struct MeshData {
  Mesh *GetParentPointer() const;
  int NumBlocks() const;
  int GetPartitionId() const;
  IndexRange GetBoundsI(IndexDomain) const;
  IndexRange GetBoundsJ(IndexDomain) const;
  IndexRange GetBoundsK(IndexDomain) const;
  std::shared_ptr<MeshBlockData<T>> &GetBlockData(int) const;
};
struct SparsePackStandIn {
  Real &operator()(int b, int n, int k, int j, int i) const;
  Real &operator()(int b, TopologicalElement te, int n, int k, int j, int i) const;
  Real &flux(int b, int dir, int n, int k, int j, int i) const;
};
struct PackDescriptorStandIn {
  template <class MD>
  SparsePackStandIn GetPack(MD *) const;
};
template <class... Ts>
PackDescriptorStandIn MakePackDescriptor(ResolvedPackages *, const std::vector<int> & = {}, const std::vector<PDOpt> & = {});
struct LowStorageIntegrator {
  Real dt;
  std::vector<Real> gam0, gam1, beta;
};
} // namespace parthenon

// field types (artemis.hpp:39-76 declares them with Parthenon's SPARSE_VARIABLE macros)
namespace gas {
namespace prim { struct density {}; struct velocity {}; struct pressure {}; struct sie {}; }
namespace cons { struct density {}; struct momentum {}; struct total_energy {}; struct internal_energy {}; }
namespace face { struct velocity {}; }
namespace diff { struct momentum {}; struct energy {}; }
}
namespace dust {
namespace prim { struct density {}; struct velocity {}; }
namespace cons { struct density {}; struct momentum {}; }
}
