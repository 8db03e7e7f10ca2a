// TEST STAND-IN (see ../../artemis.hpp): the names of src/utils/diffusion/diffusion_coeff.hpp:23-136 the adapter touches.
#pragma once
#include "artemis.hpp"
namespace Diffusion {
enum class DiffType { viscosity_plaw, viscosity_alpha, conductivity_plaw, thermaldiff_plaw, null };
enum class DiffAvg { arithmetic, harmonic, null };
struct DiffCoeffParams {
  DiffType type = DiffType::null;
  DiffAvg avg = DiffAvg::arithmetic;
  Real nu_s = 0, eta = 0, r_exp = 0;
  Real alpha = 0, R0 = 1, Omega0 = 0;
  Real kappa_0 = 0;
  Real hcond_0 = 0, temp_exp = 0, rho_exp = 0, T0 = 1, d0 = 1;
};
} // namespace Diffusion
