// Path-selection and tuning switches of libartemis_hip.so (device library and host driver alike): ONE table, filled from
// the environment once -- ARTEMIS_<NAME> present = its integer value, or 1 when it holds no number -- at the first use,
// and changed at run time only through artemis_hip_set_option (include/artemis_hip.h).  A launch reads an array entry;
// nothing on a launch path calls getenv.  Every path a switch selects gives the same bits: tests use them to hold one
// path against another; the tuning knobs (planes per chunk, rows per strip) are for measurements.
#pragma once

namespace artemis {

#define ARTEMIS_OPTION_LIST(X)                                                                      \
  /* path selection, device library */                                                              \
  X(NO_TUNED) X(TUNED_2D) X(NO_STAGE2D) X(NO_FUSED_CURV) X(NO_CURV_MARCH) X(NO_CURV_DUST) X(NO_CURV_DUST_MARCH) X(NO_DRAG_IN_MARCH) X(NO_STRAT_IN_KERNEL) X(NO_CART_MARCH) X(NO_IC_IN_SHELL) X(NO_IC_SKIP) \
  X(NO_ML_FUSED) X(NO_EPILOGUE) X(NO_TILED_FLUX) X(NO_VISC_SOURCE) X(NBODY_TASK) X(NBODY_GENERAL)    \
  X(NO_PLM_TABLE) X(NO_DISTANCE_TABLE) X(NO_FLAT_RANGES) X(FULL_REMESH)                              \
  /* exactness machinery (measuring what it costs) */                                                \
  X(NO_REDO) X(NO_TINY_HINT) X(NO_ML_FLOOR)                                                                   \
  /* host loop */                                                                                    \
  X(NO_GRAPH) X(SYNC_LOOP) X(FORCE_OVERLAP) X(LOOPBACK_COMM) X(WAIT_SPIN_LIMIT) X(TEST_SHELL_TARGET_BUMP)  \
  X(HOST_THREADS) X(SETUP_TIMING) X(AMR_DEBUG)                                                        \
  /* tuning knobs */                                                                                  \
  X(FUSED_KCHUNK) X(CURV_KCHUNK) X(VISC_KCHUNK) X(STAGE2D_ROWS) X(STAGE2D_RGRID) X(FUSED_NO_SWIZZLE)  \
  /* memory */                                                                                        \
  X(NO_POOL) X(POOL_GB) X(TRIM_POOL) X(POISON) X(DENSE_FLUX)

enum Opt {
#define X(name) OPT_##name,
  ARTEMIS_OPTION_LIST(X)
#undef X
      OPT_COUNT
};

extern long g_options[OPT_COUNT];
extern bool g_options_ready;
void options_load(); // (abi.hip: reads the environment, once)

inline long opt(Opt o) {
  if (!g_options_ready) options_load();
  return g_options[o];
}

} // namespace artemis
