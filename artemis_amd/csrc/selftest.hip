// Device-side self test of the hand-scheduled fp64 division / square root used by the fused
// kernel against the compiler's IEEE-correct `/` and sqrt() on the same operands.
#include "device_math.hpp"
#include "../../include/artemis_hip.h"

namespace artemis {
__global__ void divsqrt_kernel(long n, const double *a, const double *b, double *q_fast, double *q_ieee,
                               double *s_fast, double *s_ieee) {
  const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Recip r = recip(b[i]);
  q_fast[i] = div(a[i], r);
  q_ieee[i] = a[i] / b[i];
  const double x = fabs(b[i]);
  s_fast[i] = sqrt_pos(x);
  s_ieee[i] = sqrt(x);
}
} // namespace artemis

extern "C" int artemis_hip_selftest_divsqrt(long n, const double *a, const double *b, double *q_fast,
                                            double *q_ieee, double *s_fast, double *s_ieee,
                                            void *stream) {
  if (n <= 0 || !a || !b || !q_fast || !q_ieee || !s_fast || !s_ieee) return ARTEMIS_HIP_EINVAL;
  hipLaunchKernelGGL(artemis::divsqrt_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n, a, b, q_fast, q_ieee, s_fast, s_ieee);
  return hipGetLastError() == hipSuccess ? ARTEMIS_HIP_OK : ARTEMIS_HIP_EDEVICE;
}
