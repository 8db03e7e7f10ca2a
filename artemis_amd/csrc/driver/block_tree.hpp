// Logical mesh-block tree of a statically refined mesh and the box-to-box operation lists of its ghost
// exchange and flux correction.  Host-only, no device calls: the lists are uploaded once per mesh and each
// is processed by ONE launch of kernels_amr.hip (include/artemis_hip.h "multilevel block-graph data path").
//
// Stands where Parthenon's MeshBlockTree / neighbour search / boundary-buffer set-up stand for the path
// (upstream, version unpinned: the submodule is empty in the reference checkout; restated from the published
// Athena++ / Parthenon algorithm).  Artemis-side anchors: <parthenon/mesh> refinement = static with
// <parthenon/static_refinementN> regions (inputs/disk/disk_cart.in:42,68-75), the exchange at
// artemis_driver.cpp:258 (AddBoundaryExchangeTasks(..., pmesh->multilevel)), flux correction at :196-202.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <set>
#include <stdexcept>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "artemis_hip.h"

namespace artemis_host {

typedef std::array<int, 3> Loc;

struct Leaf {
  int level;
  Loc lx;
};

class BlockTree {
 public:
  int nrb[3] = {1, 1, 1}; // root grid
  int ndim = 1;
  bool periodic[3] = {false, false, false};

  int extent(int level, int d) const { return d < ndim ? (nrb[d] << level) : 1; }
  // periodic images folded back; false outside a physical boundary
  bool wrap(int level, Loc &l) const {
    for (int d = 0; d < 3; ++d) {
      const int n = extent(level, d);
      if (l[d] < 0 || l[d] >= n) {
        if (!periodic[d]) return false;
        l[d] = ((l[d] % n) + n) % n;
      }
    }
    return true;
  }
  Loc parent(const Loc &l) const {
    Loc p = {0, 0, 0};
    for (int d = 0; d < ndim; ++d) p[d] = l[d] >> 1;
    return p;
  }
  bool is_internal(int level, const Loc &l) const { return internal_.count(key(level, l)) > 0; }
  bool exists(int level, const Loc &l) const { return level == 0 || is_internal(level - 1, parent(l)); }

  // the node (level, l) exists afterwards, refining ancestors as needed
  void ensure(int level, const Loc &l) {
    if (level > 0 && !exists(level, l)) refine(level - 1, parent(l));
  }
  // split a node; every neighbour across a face, edge or corner must then exist at the node's level
  // (2:1 balance: a leaf's neighbours differ from it by at most one level)
  void refine(int level, const Loc &l) {
    if (is_internal(level, l)) return;
    ensure(level, l);
    internal_.insert(key(level, l));
    if (level + 1 > max_level_) max_level_ = level + 1;
    for (int o3 = (ndim > 2 ? -1 : 0); o3 <= (ndim > 2 ? 1 : 0); ++o3)
      for (int o2 = (ndim > 1 ? -1 : 0); o2 <= (ndim > 1 ? 1 : 0); ++o2)
        for (int o1 = -1; o1 <= 1; ++o1) {
          if (!o1 && !o2 && !o3) continue;
          Loc n = {l[0] + o1, l[1] + o2, l[2] + o3};
          if (wrap(level, n)) ensure(level, n);
        }
  }
  // <parthenon/static_refinementN>: every block of `level` that overlaps [lo, hi] exists.  Logical range as
  // upstream: first block whose upper edge lies beyond lo .. first whose upper edge reaches hi, rounded out to
  // sibling pairs; block edges from the uniform mesh generator x(r) = xmin (1 - r) + xmax r.
  void add_region(int level, const double lo[3], const double hi[3], const double xmin[3], const double xmax[3]) {
    int r0[3] = {0, 0, 0}, r1[3] = {0, 0, 0};
    for (int d = 0; d < ndim; ++d) {
      const int n = extent(level, d);
      auto edge = [&](int l) {
        if (l >= n) return xmax[d];
        const double r = static_cast<double>(l) / n;
        return xmin[d] * (1.0 - r) + xmax[d] * r;
      };
      int lmin = 0;
      while (lmin < n - 1 && !(edge(lmin + 1) > lo[d])) ++lmin;
      int lmax = lmin;
      while (lmax < n - 1 && !(edge(lmax + 1) >= hi[d])) ++lmax;
      if (lmin % 2 == 1) --lmin;
      if (lmax % 2 == 0) ++lmax;
      r0[d] = lmin, r1[d] = std::min(lmax, n - 1);
    }
    for (int l3 = r0[2]; l3 <= r1[2]; ++l3)
      for (int l2 = r0[1]; l2 <= r1[1]; ++l2)
        for (int l1 = r0[0]; l1 <= r1[0]; ++l1) ensure(level, Loc{l1, l2, l3});
  }
  int max_level() const { return max_level_; }

  // leaves in Z-order (x1 fastest among the children), the order blocks are numbered and dealt to ranks in
  std::vector<Leaf> leaves() const {
    std::vector<Leaf> out;
    for (int l3 = 0; l3 < nrb[2]; ++l3)
      for (int l2 = 0; l2 < nrb[1]; ++l2)
        for (int l1 = 0; l1 < nrb[0]; ++l1) walk(0, Loc{l1, l2, l3}, out);
    return out;
  }

 private:
  // (level, lx) packed into one word: 4 + 3 x 20 bits (a root grid of a few thousand blocks per dimension refined
  // sixteen times would not fit any memory either).  Hashed: the neighbour searches of a remesh make ~10^6 look-ups
  typedef std::uint64_t Key;
  static Key key(int level, const Loc &l) {
    return (static_cast<Key>(level) << 60) | (static_cast<Key>(l[0]) << 40) | (static_cast<Key>(l[1]) << 20) | static_cast<Key>(l[2]);
  }
  std::unordered_set<Key> internal_;
  int max_level_ = 0;
  void walk(int level, const Loc &l, std::vector<Leaf> &out) const {
    if (!is_internal(level, l)) {
      out.push_back(Leaf{level, l});
      return;
    }
    for (int c3 = 0; c3 < (ndim > 2 ? 2 : 1); ++c3)
      for (int c2 = 0; c2 < (ndim > 1 ? 2 : 1); ++c2)
        for (int c1 = 0; c1 < 2; ++c1) walk(level + 1, Loc{2 * l[0] + c1, 2 * l[1] + c2, 2 * l[2] + c3}, out);
  }
};

// ---- operation lists ------------------------------------------------------------------------------------
// One entry per (destination block, direction, source block) in a global, rank-independent order, with GLOBAL
// block ids; the driver keeps the entries that touch its rank and turns ids into local indices / message slots.
struct GlobalOp {
  artemis_ml_op_t op; // dst_block / src_block hold GLOBAL ids here
};
struct MeshOps {
  std::vector<GlobalOp> ghost;            // SAME, FROM_FINER, FROM_COARSER
  std::vector<GlobalOp> flux;             // FLUX
  std::vector<artemis_ml_box_t> prolong;  // block = GLOBAL id; coarse-buffer boxes facing coarser neighbours
  std::vector<char> has_coarser;          // per global block
};

// nx: zones per block; ng: ghost zones (even).  Index conventions: fine arrays and coarse buffers both start
// their interior at index s = ng in active dimensions (0 otherwise); a coarse buffer has nx/2 interior zones.
inline MeshOps build_mesh_ops(const BlockTree &t, const std::vector<Leaf> &leaves, const int nx[3], int ng) {
  MeshOps M;
  M.has_coarser.assign(leaves.size(), 0);
  // id lookup
  std::unordered_map<std::uint64_t, int> table;
  table.reserve(leaves.size() * 2);
  auto pack = [](int level, const Loc &l) {
    return (static_cast<std::uint64_t>(level) << 60) | (static_cast<std::uint64_t>(l[0]) << 40) |
           (static_cast<std::uint64_t>(l[1]) << 20) | static_cast<std::uint64_t>(l[2]);
  };
  for (size_t b = 0; b < leaves.size(); ++b) table.emplace(pack(leaves[b].level, leaves[b].lx), static_cast<int>(b));
  auto id_of = [&](int level, const Loc &l) {
    auto it = table.find(pack(level, l));
    if (it == table.end()) throw std::logic_error("block tree: leaf expected (2:1 balance violated?)");
    return it->second;
  };
  const int ndim = t.ndim;
  int s[3], cn[3];
  for (int d = 0; d < 3; ++d) s[d] = (d < ndim) ? ng : 0, cn[d] = (d < ndim) ? nx[d] / 2 : 1;
  const int cng = (ng + 1) / 2 + 1; // coarse zones a fine block receives from a coarser neighbour
  const int h = ng / 2;             // coarse zones covering the fine ghost zones
  for (size_t b = 0; b < leaves.size(); ++b) {
    const int level = leaves[b].level;
    const Loc &lx = leaves[b].lx;
    for (int o3 = (ndim > 2 ? -1 : 0); o3 <= (ndim > 2 ? 1 : 0); ++o3)
      for (int o2 = (ndim > 1 ? -1 : 0); o2 <= (ndim > 1 ? 1 : 0); ++o2)
        for (int o1 = -1; o1 <= 1; ++o1) {
          if (!o1 && !o2 && !o3) continue;
          const int o[3] = {o1, o2, o3};
          Loc n = {lx[0] + o1, lx[1] + o2, lx[2] + o3};
          if (!t.wrap(level, n)) continue; // physical boundary: boundary conditions fill it
          artemis_ml_op_t op;
          op.dst_block = static_cast<int>(b), op.dir = 0, op.buf = 0;
          // the fine ghost box of direction o
          for (int d = 0; d < 3; ++d) {
            if (o[d] < 0) op.lo[d] = s[d] - ng, op.n[d] = ng;
            else if (o[d] > 0) op.lo[d] = s[d] + nx[d], op.n[d] = ng;
            else op.lo[d] = s[d], op.n[d] = (d < ndim) ? nx[d] : 1;
          }
          if (!t.exists(level, n)) {
            // coarser neighbour: its interior -> my coarse buffer, cng zones deep
            const Loc p = t.parent(n);
            op.kind = ARTEMIS_ML_FROM_COARSER, op.src_block = id_of(level - 1, p);
            for (int d = 0; d < 3; ++d) {
              if (o[d] < 0) op.lo[d] = s[d] - cng, op.n[d] = cng;
              else if (o[d] > 0) op.lo[d] = s[d] + cn[d], op.n[d] = cng;
              else op.lo[d] = s[d], op.n[d] = cn[d];
              // coarse zone ci of mine = global coarse zone (n - o) cn + (ci - s) in the neighbour's frame
              op.off[d] = (d < ndim) ? (n[d] - o[d]) * cn[d] - p[d] * nx[d] : 0;
            }
            M.ghost.push_back(GlobalOp{op});
            M.has_coarser[b] = 1;
            artemis_ml_box_t bx;
            bx.block = static_cast<int>(b);
            for (int d = 0; d < 3; ++d) {
              if (o[d] < 0) bx.lo[d] = s[d] - h, bx.n[d] = h;
              else if (o[d] > 0) bx.lo[d] = s[d] + cn[d], bx.n[d] = h;
              else bx.lo[d] = s[d], bx.n[d] = cn[d];
            }
            M.prolong.push_back(bx);
          } else if (!t.is_internal(level, n)) {
            op.kind = ARTEMIS_ML_SAME, op.src_block = id_of(level, n);
            for (int d = 0; d < 3; ++d) op.off[d] = -o[d] * nx[d];
            M.ghost.push_back(GlobalOp{op});
          } else {
            // finer neighbours: the children of n that touch me, each covering a half / quarter of the box
            for (int c3 = 0; c3 < (ndim > 2 ? 2 : 1); ++c3)
              for (int c2 = 0; c2 < (ndim > 1 ? 2 : 1); ++c2)
                for (int c1 = 0; c1 < 2; ++c1) {
                  const int c[3] = {c1, c2, c3};
                  bool touches = true;
                  for (int d = 0; d < ndim; ++d)
                    if ((o[d] < 0 && c[d] != 1) || (o[d] > 0 && c[d] != 0)) touches = false;
                  if (!touches) continue;
                  const Loc child = {2 * n[0] + c1, (ndim > 1) ? 2 * n[1] + c2 : 0, (ndim > 2) ? 2 * n[2] + c3 : 0};
                  artemis_ml_op_t f = op;
                  f.kind = ARTEMIS_ML_FROM_FINER, f.src_block = id_of(level + 1, child);
                  for (int d = 0; d < 3; ++d) {
                    if (d < ndim && o[d] == 0) f.lo[d] = s[d] + c[d] * cn[d], f.n[d] = cn[d];
                    // fine index on the child of my zone i: 2 ((n - o) nx + (i - s)) - child nx + s
                    f.off[d] = (d < ndim) ? 2 * ((n[d] - o[d]) * nx[d] - s[d]) - child[d] * nx[d] + s[d] : 0;
                  }
                  M.ghost.push_back(GlobalOp{f});
                  if (std::abs(o1) + std::abs(o2) + std::abs(o3) == 1) {
                    // flux correction through this quarter of my face: my face s (o < 0) or s + nx (o > 0) <- the
                    // child's opposite boundary face
                    artemis_ml_op_t x = f;
                    x.kind = ARTEMIS_ML_FLUX;
                    for (int d = 0; d < 3; ++d)
                      if (o[d] != 0) {
                        x.dir = d;
                        x.lo[d] = (o[d] < 0) ? s[d] : s[d] + nx[d], x.n[d] = 1;
                        x.off[d] = (o[d] < 0) ? s[d] + nx[d] : s[d];
                      }
                    M.flux.push_back(GlobalOp{x});
                  }
                }
          }
        }
  }
  return M;
}

} // namespace artemis_host
