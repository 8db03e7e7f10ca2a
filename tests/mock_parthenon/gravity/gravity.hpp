// TEST STAND-IN (see ../artemis.hpp): the names of src/gravity/gravity.hpp the adapter touches.
#pragma once
#include "artemis.hpp"
namespace Gravity {
enum class GravityType { uniform, point, binary, nbody, null }; // gravity.hpp:25
struct Orbit {                                                   // gravity.hpp:30-116 (only the call the adapter makes)
  Real pos0[3] = {0, 0, 0};
  void solve(Real, const Real, Real *pos, Real *vel) {
    for (int d = 0; d < 3; ++d) pos[d] = pos0[d], vel[d] = 0.0;
  }
};
} // namespace Gravity
