// TEST STAND-IN (see ../artemis.hpp): the data members of NBody::Particle (src/nbody/particle_base.hpp:52-67).
#pragma once
#include "artemis.hpp"
namespace parthenon {
template <class T>
struct ParArray2D {
  std::shared_ptr<std::vector<T>> v;
  int n1 = 0;
  ParArray2D() = default;
  ParArray2D(const std::string &, int n0_, int n1_) : v(std::make_shared<std::vector<T>>(static_cast<size_t>(n0_) * n1_)), n1(n1_) {}
  T &operator()(int i, int j) const { return (*v)[static_cast<size_t>(i) * n1 + j]; }
  ParArray2D GetHostMirrorAndCopy() const { return *this; }
  void DeepCopy(const ParArray2D &) {}
};
} // namespace parthenon
namespace NBody {
struct Particle {
  int id = 0;
  Real GM = 0, pos[3] = {0, 0, 0}, vel[3] = {0, 0, 0}, xf[3] = {0, 0, 0}, vf[3] = {0, 0, 0};
  Real radius = 0;
  int couple = 1, live = 0, alive = 1;
  Real live_after = 0;
  Real rs = 0, racc = 0, gamma = 0, beta = 0;
  Real target_rad = 0;
  int spline = 0;
};
} // namespace NBody
