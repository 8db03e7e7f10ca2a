// Host-side launchers of the HIP kernels (internal; the public surface is include/artemis_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include "pack_view.hpp"

namespace artemis {
void launch_calculate_fluxes(const PackView &P, int fluid, int riemann, int recon, hipStream_t s);
void launch_apply_update(const PackView &P, double gam0, double gam1, double beta_dt, hipStream_t s);
void launch_flux_source(const PackView &P, int fluid, double dt, hipStream_t s);
void launch_set_aux(const PackView &P, hipStream_t s);
void launch_cons_to_prim(const PackView &P, hipStream_t s);
void launch_prim_to_cons(const PackView &P, hipStream_t s, bool ghosts_only = false);
void launch_deep_copy(const PackView &P, hipStream_t s);
void launch_estimate_dt(const PackView &P, int fluid, double cfl, double *dt_dev, hipStream_t s);
int launch_apply_bc(const PackView &P, const int *bc, const artemis_bc_params_t *par, hipStream_t s);
// kernels_sources.hip
void launch_external_gravity(const PackView &P, const artemis_gravity_t &G, double dt, hipStream_t s);
int nbody_grid(const PackView &P);
void launch_nbody_gravity(const PackView &P, const artemis_nbody_particle_t *pl_dev, int npart, double omf, double dt,
                          double *partial_dev, hipStream_t s);
bool nbody_force_sums_covers(const PackView &P);
void launch_nbody_force_sums(const PackView &P, const artemis_nbody_particle_t *pl_dev, int npart, double omf, double dt,
                             const double *dt_dev, double *partial_dev, double *force_dev, hipStream_t s);
void launch_shearing_box(const PackView &P, double omega, double qshear, double dt, hipStream_t s);
void launch_rotating_frame(const PackView &P, double omega, double dt, hipStream_t s);
void launch_cooling(const PackView &P, const artemis_cooling_t &C, double dt, hipStream_t s);
// dt_dev (optional DEVICE scalar) replaces dt
void launch_drag_source(const PackView &P, const artemis_drag_t &D, double dt, const double *dt_dev,
                        hipStream_t s);
bool launch_drag_finish(const PackView &P, const artemis_drag_t &D, double dt, const double *dt_dev,
                        hipStream_t s);
bool launch_drag_finish_cells(const PackView &P, const artemis_drag_t &D, double dt, const artemis_ml_fix_cell_t *cells,
                              int ncells, hipStream_t s);
bool drag_finish_in_march(const PackView &P, const artemis_drag_t &D);
long halo_count(const PackView &P, int face, int extended = 0);
int launch_halo(const PackView &P, int block, int face, double *buf, int unpack, int extended,
                hipStream_t s);
void invalidate_table_cache();
long plm_table_count(const PackView &P);
void launch_plm_table_fill(const PackView &P, double *tab, hipStream_t s);
// kernels_fused.hip
void launch_advance_dt(double *state, double tlim, int nstages, const double *beta, hipStream_t s);
void launch_wait_counter(unsigned *counter, unsigned target, unsigned *timeout_flag, hipStream_t s);
int launch_stage_fused(const PackView &P, const artemis_stage_args_t &a, int riemann, int recon,
                       hipStream_t s);
int launch_stage_fused_redo_shell(const PackView &P, const artemis_stage_args_t &a, int riemann, int recon, hipStream_t s);
bool fused_flux_covers(const PackView &P, int recon);
int launch_flux_fused(const PackView &P, int riemann, int recon, hipStream_t s);
bool fused_curv_covers(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas);
void launch_stage_fused_curv(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas, int riemann_gas,
                             hipStream_t s);
// kernels_curv.hip
bool curv_march_covers(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas);
// Chunks of an x3 march (kernels_curv.hip, the viscous source): a launch proceeds in rounds of `slots` resident
// workgroups and a half-empty last round costs a full one; every chunk pays `prime` priming trips (in units of a full
// trip: measured, a priming trip of the stage march costs about half a trip).  The number of chunks
// (each of at most cmax planes) with the best product of round fill and march efficiency; ties go to fewer chunks.
inline int pick_march_chunks(int planes, long tiles, long slots, int cmax, double prime) {
  int best = (planes + cmax - 1) / cmax;
  double score = -1.0;
  for (int n = (planes + cmax - 1) / cmax; n <= planes && n <= 64; ++n) {
    const int kc = (planes + n - 1) / n;
    if (kc < 4 && n > (planes + cmax - 1) / cmax) break;
    const int nchunk = (planes + kc - 1) / kc;
    const long wgs = tiles * nchunk;
    const double fill = static_cast<double>(wgs) / static_cast<double>(((wgs + slots - 1) / slots) * slots);
    const double sc = fill * kc / static_cast<double>(kc + prime);
    if (sc > score + 1e-9) score = sc, best = nchunk;
  }
  return best;
}
bool curv_march_covers_dust(const PackView &P, const artemis_stage_general_args_t &g, int recon_dust, int riemann_dust);
// finish (fluid 1 only): the dust march also runs DragSource + SetAuxillaryFields + ConsToPrim of both fluids (the gas march has
// left its conserved state in P.gas.cons0) and, with g.dt_dev, both fluids' timestep limits
void launch_stage_curv(const PackView &P, const artemis_stage_general_args_t &g, int fluid, int recon, int riemann, hipStream_t s,
                       bool finish = false);
// kernels_diffusion.hip
void launch_zero_diffusion_flux(const PackView &P, hipStream_t s);
// overwrite: ZeroDiffusionFlux folded in (the flux arrays are overwritten on the face ranges)
size_t viscous_distance_count(const PackView &P);
void launch_viscous_distance_fill(const PackView &P, double *tab, hipStream_t s);
int launch_viscous_flux(const PackView &P, const artemis_diffusion_t &D, hipStream_t s, bool overwrite = false);
void launch_viscous_listed_faces(const PackView &P, const artemis_diffusion_t &D, const artemis_ml_face_box_t *boxes, int nboxes,
                                 const artemis_ml_fix_cell_t *cells, int ncells, hipStream_t s);
bool viscous_source_covers(const PackView &P);
void launch_viscous_source(const PackView &P, const artemis_diffusion_t &D, double dt, const double *dt_dev, double *const *out,
                           hipStream_t s);
void launch_thermal_flux(const PackView &P, const artemis_diffusion_t &D, hipStream_t s);
void launch_diffusion_update(const PackView &P, const artemis_diffusion_t &D, double dt, hipStream_t s);
void launch_timestep_all(const PackView &P, const artemis_diffusion_t *D, double cfl_gas, double cfl_dust, double *dt_dev,
                         hipStream_t s);
void launch_diffusion_dt(const PackView &P, const artemis_diffusion_t &D, double cfl, double *dt_dev,
                         hipStream_t s);
// kernels_stage_cell.hip
void launch_refine(const artemis_refine_t &r, int prolongate, hipStream_t s);
void launch_amr_criterion(const artemis_amr_criterion_t &a, int magnitude, hipStream_t s);
void launch_pack_criterion(const PackView &P, int var, int var_sie, int magnitude, double *maxima, hipStream_t s);
void launch_stage_epilogue(const PackView &P, const artemis_stage_general_args_t &g, hipStream_t s, bool to_cons = false);
void launch_stage_cell(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas,
                       int riemann_gas, int recon_dust, int riemann_dust, hipStream_t s);
// kernels_stage2d.hip
void launch_ml_face_fluxes(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas, int riemann_gas,
                           int recon_dust, int riemann_dust, const artemis_ml_face_box_t *boxes, int nboxes, hipStream_t s);
void launch_ml_stage_fixup(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas, int riemann_gas,
                           int recon_dust, int riemann_dust, const artemis_ml_fix_cell_t *cells, int ncells, hipStream_t s);
int stage_general_variant(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas, int riemann_gas,
                          int recon_dust, int riemann_dust);
bool stage2d_covers(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas, int riemann_gas, int recon_dust,
                    int riemann_dust);
void launch_stage2d(const PackView &P, const artemis_stage_general_args_t &g, int recon_gas, int riemann_gas, int riemann_dust,
                    hipStream_t s);
// kernels_amr.hip
void launch_ml_exchange(const PackView &P, const artemis_ml_pack_t &ml, const artemis_ml_op_t *ops, int nops, double *sbuf,
                        const double *rbuf, hipStream_t s);
void launch_ml_flux_correction(const PackView &P, const artemis_ml_op_t *ops, int nops, double *sbuf, const double *rbuf,
                               hipStream_t s);
void launch_ml_restrict_halos(const PackView &P, const artemis_ml_pack_t &ml, const int *blocks, int nblocks, hipStream_t s);
void launch_ml_prolongate(const PackView &P, const artemis_ml_pack_t &ml, const artemis_ml_box_t *boxes, int nboxes,
                          hipStream_t s);
void launch_ml_floor_ghosts(const PackView &P, const int *blocks, int nblocks, hipStream_t s);
} // namespace artemis
