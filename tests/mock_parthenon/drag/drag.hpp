// TEST STAND-IN (see ../artemis.hpp): the names of src/drag/drag.hpp:55-142 the adapter touches.
#pragma once
#include "artemis.hpp"
namespace Drag {
enum class Coupling { simple_dust, self, null };
enum class DragModel { constant, stokes, null };
struct SelfDragParams {
  Real ix[3], ox[3];
  Real xmin[3], xmax[3];
  Real irate[3], orate[3];
  bool damp_to_visc = false;
  SelfDragParams() {
    for (int i = 0; i < 3; i++) ix[i] = -1.7976931348623157e308, ox[i] = 1.7976931348623157e308, irate[i] = orate[i] = 0.0, xmin[i] = xmax[i] = 0.0;
  }
};
struct StoppingTimeParams {
  Real scale = 1.0;
  DragModel model = DragModel::constant;
  parthenon::ParArray1D<Real> tau;
};
} // namespace Drag
