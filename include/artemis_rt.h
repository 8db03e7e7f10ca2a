/* =======================================================================================
 * artemis_rt.h -- tiny device-runtime shim exported by libartemis_hip.so next to the compute
 * entry points of artemis_hip.h.
 *
 * The host driver (artemis_amd/csrc/driver, plain C++ built with g++) owns no HIP code: it
 * allocates, copies and orders work only through these calls, the way Artemis reaches the
 * device only through Kokkos/Parthenon (DevExecSpace(), ParArray allocations, par_for
 * launches; e.g. artemis_integrator.hpp:42, fill_derived.cpp:54).  Streams and events are
 * hipStream_t / hipEvent_t passed as void*.
 *
 * All functions returning int use the artemis_hip.h error codes; pointer-returning
 * functions return NULL on failure (message in artemis_hip_last_error()).
 * ===================================================================================== */
#ifndef ARTEMIS_RT_H_
#define ARTEMIS_RT_H_
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

int artemis_rt_set_device(int dev);
void *artemis_rt_malloc(size_t bytes);      /* device memory (HBM) */
void artemis_rt_free(void *p);
/* Optional buffer cache behind artemis_rt_malloc / artemis_rt_free, OFF unless artemis_rt_pool_limit(bytes > 0) was
 * called (the standalone driver turns it on for adaptive meshes: ARTEMIS_POOL_GB, default 64; a library host -- the
 * Parthenon adapter -- shares the device with Kokkos' allocator and never pays for it unless it asks).  When on, freed
 * buffers are kept per device in size classes (1/16 of a power of two) and handed out again to the device they live on: a
 * remesh frees and allocates tens of GB in buffers whose sizes barely change, and hipFree + hipMalloc of that costs
 * seconds.  The cache is trimmed to the limit, emptied when hipMalloc runs out of memory, and given back on request
 * (artemis_rt_pool_trim) -- a host under memory pressure calls artemis_rt_pool_trim(0) or artemis_rt_pool_limit(0).
 * artemis_rt_free of a cached buffer synchronises the device first, as hipFree does.
 * artemis_rt_device_bytes: the device footprint through artemis_rt_malloc (live and cached buffers at their
 * capacities) and its high-water mark since the start (or since the last call with reset_peak != 0) -- what a remesh,
 * which builds the new mesh next to what it keeps of the old one, costs. */
void artemis_rt_pool_limit(size_t limit_bytes);
void artemis_rt_device_bytes(size_t *current, size_t *peak, int reset_peak);
/* ... of which cached (free, ready to be handed out again): what a host could get back with artemis_rt_pool_trim(0) */
size_t artemis_rt_pool_bytes(void);
/* give cached buffers back to the device until at most keep_bytes of them are left (oldest first); the driver calls it
 * with 0 once the initial mesh is built (the smaller meshes of the initial refinement loop leave buffers nobody asks
 * for again) */
void artemis_rt_pool_trim(size_t keep_bytes);
void *artemis_rt_malloc_host(size_t bytes); /* pinned host memory */
void artemis_rt_free_host(void *p);
int artemis_rt_memcpy_h2d(void *dst, const void *src, size_t n, void *stream);
int artemis_rt_memcpy_d2h(void *dst, const void *src, size_t n, void *stream);
int artemis_rt_memcpy_d2d(void *dst, const void *src, size_t n, void *stream);
int artemis_rt_memset(void *dst, int value, size_t n, void *stream);
void *artemis_rt_stream_create(void);
void artemis_rt_stream_destroy(void *stream);
int artemis_rt_stream_sync(void *stream);
int artemis_rt_device_sync(void);
void *artemis_rt_event_create(void);
void artemis_rt_event_destroy(void *ev);
int artemis_rt_event_record(void *ev, void *stream);
int artemis_rt_stream_wait_event(void *stream, void *ev);
int artemis_rt_event_sync(void *ev);
double artemis_rt_event_elapsed_ms(void *ev0, void *ev1);
/* Kept for ABI stability: pointer tables are always read on the device, nothing is cached. */
void artemis_rt_tables_changed(void);
/* hipGraph capture of a launch sequence on a (non-default) stream: begin, issue the launches, end ->
 * an instantiated executable graph (NULL on failure), replayed with _graph_launch.  The host driver
 * captures one time step of its synchronisation-free loop, so small meshes are not bound by the
 * launch rate. */
int artemis_rt_capture_begin(void *stream);
void *artemis_rt_capture_end(void *stream);
int artemis_rt_graph_launch(void *graph_exec, void *stream);
void artemis_rt_graph_destroy(void *graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* ARTEMIS_RT_H_ */
