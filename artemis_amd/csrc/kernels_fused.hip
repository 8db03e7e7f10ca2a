// placeholder until the fused stage kernel lands
#include "kernels.hpp"
namespace artemis {
int launch_stage_fused(const PackView &, const artemis_stage_args_t &, int, int, hipStream_t) { return 99; }
}
