/* =======================================================================================
 * artemis_driver.h -- C ABI of the host driver that stands in for Parthenon + ArtemisDriver
 * when the path runs outside Artemis (bench, tests, the `artemis` command-line tool).
 *
 * It consumes the reference's own input decks (inputs/<problem>/<name>.in, Parthenon
 * `<block>` / `key = value # comment` syntax with `&` continuation) and `block/key=value`
 * overrides exactly like `artemis -i deck block/key=value ...` (tst/scripts/utils/
 * artemis.py:122-156), builds a uniform Cartesian mesh of equal mesh blocks, runs the problem
 * generator, and advances with the stage order of ArtemisDriver<GEOM>::StepTasks
 * (artemis_driver.cpp:145-273).  All device work goes through artemis_hip.h / artemis_rt.h;
 * this layer contains no HIP code.
 *
 * Decomposition: `nranks` processes, one GPU each; the mesh-block grid is split into a
 * Cartesian grid of rank bricks.  Ghost slabs between blocks of one rank are copied on the
 * device; slabs between ranks go through an `artemis_comm_t`: the native RCCL transport below
 * (artemis_comm_rccl_create; what bench.py and a C++ host use), or callbacks supplied by the launcher
 * (the CPU tests of the host logic bind them to torch.distributed / gloo).
 * ===================================================================================== */
#ifndef ARTEMIS_DRIVER_H_
#define ARTEMIS_DRIVER_H_
#ifdef __cplusplus
extern "C" {
#endif

typedef struct artemis_sim artemis_sim_t;

typedef struct artemis_msg {
  int peer;          /* rank on the other side */
  int tag;           /* unique per (destination block, destination face) */
  double *send;      /* DEVICE buffer to send (NULL if none) */
  double *recv;      /* DEVICE buffer to receive into (NULL if none) */
  long count;        /* doubles in each direction */
} artemis_msg_t;

typedef struct artemis_comm {
  void *ctx;
  int rank, nranks;
  /* Post every send/recv of one halo exchange.  Ordered after work already enqueued on
   * `stream`; returns without waiting. */
  int (*exchange_start)(void *ctx, int nmsg, const artemis_msg_t *msgs, void *stream);
  /* Make work enqueued on `stream` after this call wait for the exchange posted last. */
  int (*exchange_finish)(void *ctx, void *stream);
  int (*allreduce_min)(void *ctx, double *value);        /* host scalar, in place */
  /* optional (may be NULL): min-all-reduce one DEVICE double in place, ordered on `stream`;
   * lets the per-cycle dt reduction ride the stream instead of costing a host round trip */
  int (*allreduce_min_dev)(void *ctx, double *dev_value, void *stream);
  int (*allreduce_sum)(void *ctx, double *values, int n); /* host array, in place */
} artemis_comm_t;

/* ---- native transport: RCCL (xGMI inside a node), artemis_amd/csrc/driver/comm_rccl.cpp -------------
 * The C++ implementation of artemis_comm_t a C++ host links directly -- no Python, no GIL: one
 * ncclGroupStart/End of ncclSend / ncclRecv per ghost exchange on the driver's comm stream (posted in
 * tag order on both sides), ncclAllReduce(min) in place on the device dt scalar, ncclAllReduce(sum) for
 * history / error norms.  Stands where Parthenon's MPI boundary communication and reductions stand
 * (artemis_driver.cpp:258, :279-297; utils/history.hpp:29-100).
 *   rank 0:      artemis_comm_rccl_unique_id(buf, sizeof buf)   -> ship the bytes to every rank (any
 *                out-of-band channel: the launcher's store, MPI_Bcast, a file)
 *   every rank:  artemis_rt_set_device(local_rank); comm = artemis_comm_rccl_create(id, rank, nranks);
 *                sim = artemis_sim_create(deck, n, overrides, comm); ...; artemis_comm_rccl_destroy(comm)
 * create() is collective (ncclCommInitRank).  Returns NULL / non-zero on error
 * (artemis_comm_rccl_last_error()). */
int artemis_comm_rccl_unique_id_bytes(void);
int artemis_comm_rccl_unique_id(char *out, int capacity);
artemis_comm_t *artemis_comm_rccl_create(const char *unique_id, int rank, int nranks);
void artemis_comm_rccl_destroy(artemis_comm_t *comm);
int artemis_comm_rccl_count(const artemis_comm_t *comm);  /* ncclCommCount: ranks RCCL itself reports */
int artemis_comm_rccl_barrier(artemis_comm_t *comm);      /* all-reduce + stream sync */
const char *artemis_comm_rccl_last_error(void);

/* deck_text: contents of an input deck; overrides: `block/key=value` strings (may be NULL).
 * comm: NULL for a single process.  Returns NULL on error (artemis_sim_last_error()). */
artemis_sim_t *artemis_sim_create(const char *deck_text, int noverrides, const char *const *overrides,
                                  const artemis_comm_t *comm);
void artemis_sim_destroy(artemis_sim_t *sim);
const char *artemis_sim_last_error(void);

/* EvolutionDriver::Execute (parthenon, upstream): advance until tlim / nlim of the deck, or by
 * at most `max_cycles` further cycles if max_cycles >= 0.  Returns cycles taken, < 0 on error. */
long artemis_sim_evolve(artemis_sim_t *sim, long max_cycles);

double artemis_sim_time(const artemis_sim_t *sim);
double artemis_sim_dt(const artemis_sim_t *sim);
double artemis_sim_tlim(const artemis_sim_t *sim);
long artemis_sim_ncycle(const artemis_sim_t *sim);
/* interior cells owned by this rank / by all ranks */
long artemis_sim_local_zones(const artemis_sim_t *sim);
long artemis_sim_total_zones(const artemis_sim_t *sim);
/* Statically refined meshes (<parthenon/mesh> refinement = static + <parthenon/static_refinementN>): mesh blocks
 * of this rank carry their level; every block has the same number of zones.  The refined mesh runs the
 * per-task chain (flux correction needs the stage's face fluxes). */
int artemis_sim_block_level(const artemis_sim_t *sim, int block);
long artemis_sim_nblocks_global(const artemis_sim_t *sim);
int artemis_sim_uses_fused_path(const artemis_sim_t *sim);
/* 1 when the fused path runs the hand-tuned gas kernel (artemis_hip_stage_fused), 0 when it runs
 * the general cell-centred stage (artemis_hip_stage_general) or the per-task chain */
int artemis_sim_uses_tuned_kernel(const artemis_sim_t *sim);
/* name of the kernel family the stages run on: "stage_fused_kernel" (tuned 2.5-D), "stage2d_kernel" (2-D row march),
 * "stage_cell_kernel" (cell-centred general stage) or "per-task chain"; static storage */
const char *artemis_sim_stage_kernel(const artemis_sim_t *sim);
/* <parthenon/mesh> refinement = adaptive: how many times the block tree has changed so far (the initial
 * refinement passes included).  The block layout (artemis_sim_dims, block bounds / levels) changes with it. */
long artemis_sim_remeshes(const artemis_sim_t *sim);
/* Wall-clock cost of the remeshes DURING the run (the initial refinement passes are set-up): returns their number and
 * fills out[4] = {total seconds, of which building the new mesh's state, handing the data over, and -- over all
 * cycles, remeshed or not -- evaluating the refinement criterion and the tree}. */
long artemis_sim_remesh_seconds(const artemis_sim_t *sim, double *out);
/* The most recent remesh: leaves4 = {leaves before, leaves after, leaves created (not in the old mesh), leaves destroyed
 * (not in the new one)}, seconds3 = {total, building the new state, handing the data over} (zeros before the first
 * remesh of the run; either pointer may be NULL). */
void artemis_sim_last_remesh(const artemis_sim_t *sim, long *leaves4, double *seconds3);
/* Measurement hook (bench.py's remesh leg): split the leaf with global (Z-order) index gid as if the refinement
 * criterion had tagged it -- the ordinary remesh machinery runs (2:1 balance, new state, hand-over of the data, block
 * migration between ranks) and is timed like any other remesh.  Collective over the ranks (same gid everywhere).
 * Returns 1 if the mesh changed, 0 if not (the leaf is at the finest level), < 0 on error. */
int artemis_sim_force_refine(artemis_sim_t *sim, long gid);
/* ... many leaves at once, the way a criterion tags them: the listed leaves (global Z-order indices) get the tag +1
 * NEXT TO the tags the deck's own criterion gives every other leaf, derefinement counters included -- so leaves the
 * criterion does not want refined merge again `derefine_count` cycles later, by the ordinary path.  One remesh check is
 * run at once (collective; the same list on every rank).  Returns 1 if the mesh changed, 0 if not, < 0 on error. */
int artemis_sim_inject_refine_tags(artemis_sim_t *sim, const long *gids, int n);
/* Refined meshes: the largest rank's cost over the mean cost of the Z-order rank split (1 = perfectly even).  The cost
 * of a block is 1 by default (Parthenon's unit cost per block); the driver's own deck block
 *   <artemis_amd/loadbalance>  level_cost = c0, c1, ...   flux_face_cost = c
 * weighs blocks by refinement level and by the coarse-fine face operations they take part in; the split then evens
 * the cumulative cost of contiguous Z-order runs instead of their lengths.  Results do not depend on the split. */
double artemis_sim_load_balance(const artemis_sim_t *sim);
/* which: "fused" | "unfused"; selects the kernel path (fused only where supported). */
int artemis_sim_set_path(artemis_sim_t *sim, const char *which);
/* Halo exchange on a second stream concurrently with interior compute (fused path, remote
 * neighbours only).  0 = off; 1 = boundary-shell launch, then bulk launch; 2 = one launch whose
 * shell workgroups run first and signal a device counter the comm stream waits on. */
int artemis_sim_set_overlap(artemis_sim_t *sim, int overlap);
/* current mode (a timed-out mode-2 wait resets it to 0 and makes artemis_sim_evolve fail) */
int artemis_sim_overlap(const artemis_sim_t *sim);
/* Drop-in accounting for the tuned fused kernel (bench.py's `dropin` object): the last stage also writes
 * the conserved state (cons_out) and every stage ends with the whole-block PrimToCons a Parthenon host
 * runs as FillDerived (artemis.cpp:123, artemis_driver.cpp:261) -- what the fused path costs when the host
 * keeps `cons` as its Independent / Restart state.  on = 2: every stage stores cons of the zones it updates and only
 * the ghost zones go through PrimToCons after the exchange (artemis_hip_prim_to_cons_ghosts): `cons` is just as
 * current after every stage, at 5 % of the conversion work.  Same results; returns non-zero off the tuned path. */
int artemis_sim_set_dropin(artemis_sim_t *sim, int on);
/* <gravity/nbody> with <nbody> integrator = none: the accumulated particle_force rows [npart][7] = {mass accreted,
 * gravity force x3, accretion force x3} (nbody_gravity.hpp:129-136), summed over ranks like NBody::Advance does
 * (nbody_advance.cpp:123-131); reset != 0 zeroes them afterwards.  Returns the number of particles. */
int artemis_sim_nbody_force(artemis_sim_t *sim, double *out, int reset);
/* number of gas / dust species (sizes the buffers of artemis_sim_get_field and artemis_sim_history) */
void artemis_sim_species(const artemis_sim_t *sim, int *ns_gas, int *ns_dust);

/* Block layout of this rank. dims = {nblocks_local, ni, nj, nk, is, ie, js, je, ks, ke, ng}. */
void artemis_sim_dims(const artemis_sim_t *sim, int *dims);
/* Copy one field of one local block to the host, [nvar][nk][nj][ni] doubles (ghosts included).
 * field: "gas.prim" (6*ns), "gas.cons" (6*ns), "dust.prim" (4*ns), "dust.cons" (4*ns).
 * Conserved fields and pressure are materialised (PrimToCons) first.  Returns nvar or < 0. */
int artemis_sim_get_field(artemis_sim_t *sim, const char *field, int block, double *host_out);
/* interior bounds of a local block: out = {x1min,x1max,x2min,x2max,x3min,x3max} */
void artemis_sim_block_bounds(const artemis_sim_t *sim, int block, double *out);

/* History integrals (utils/history.hpp:29-100, gas.cpp:648-676, dust.cpp:332-352), reduced
 * over ranks: out = [gas mass, mom1..3, energy, internal energy] (species 0) then
 * [mass, mom1..3] per dust species.  Returns the number of values. */
int artemis_sim_history(artemis_sim_t *sim, double *out);
/* Post-loop error norms of the problem generator (linear_wave.hpp:267-377: out[0] rms,
 * out[1..5]; advection.hpp:224-405: out[0..2] rms gas/dust1/dust2, out[3..15]).  Returns the
 * number of values, 0 if the pgen defines none. */
int artemis_sim_errors(artemis_sim_t *sim, double *out);

/* Seconds spent inside the timed region of the last artemis_sim_evolve (device-synchronised
 * at both ends), and the average duration in ms of the dominant kernel's launches measured
 * with HIP events on the compute stream. */
void artemis_sim_set_kernel_timing(artemis_sim_t *sim, int on);
double artemis_sim_last_wall_seconds(const artemis_sim_t *sim);
double artemis_sim_kernel_ms(const artemis_sim_t *sim, long *nlaunch);

#ifdef __cplusplus
}
#endif
#endif /* ARTEMIS_DRIVER_H_ */
