"""ctypes mirror of include/artemis_hip.h and include/artemis_rt.h.

The shared library is the product; this module only declares prototypes.  If the library is
missing or no GPU is visible, calls fail loudly -- there is no Python or CPU fallback.
"""
import ctypes as C
import os

# torch bundles its own libamdhip64.so; importing it first makes the loader resolve our
# DT_NEEDED libamdhip64.so.7 to that single copy instead of loading a second HIP runtime.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libartemis_hip.so")

OK, EINVAL, EDEVICE, EUNSUPPORTED = 0, 1, 2, 3
CARTESIAN, CYLINDRICAL, SPHERICAL1D, SPHERICAL2D, SPHERICAL3D, AXISYMMETRIC = range(6)


def coord_select(sys, ndim):
    """geometry::CoordSelect (geometry.hpp:38-56)."""
    if sys == "spherical":
        return (SPHERICAL1D, SPHERICAL1D, SPHERICAL2D, SPHERICAL3D)[ndim]
    return {"cartesian": CARTESIAN, "cylindrical": CYLINDRICAL,
            "axisymmetric": AXISYMMETRIC}[sys]


HLLC, HLLE, LLF = 0, 1, 2
PCM, PLM, PPM = 0, 1, 2
GAS, DUST = 0, 1
BC_PERIODIC, BC_OUTFLOW, BC_REFLECT, BC_NONE, BC_STRAT_EXTRAP, BC_STRAT_INFLOW, BC_CONDUCTIVE, BC_IC, BC_DISK_EXTRAP, BC_DISK_VISC = range(10)
GRAVITY_UNIFORM, GRAVITY_POINT, GRAVITY_BINARY = 1, 2, 3
DRAG_SIMPLE_DUST, DRAG_SELF = 1, 2
DRAG_CONSTANT, DRAG_STOKES = 0, 1
DIFF_OFF, VISCOSITY_PLAW, VISCOSITY_ALPHA, CONDUCTIVITY_PLAW, THERMALDIFF_PLAW = range(5)
MAX_DUST_SPECIES = 16
RSOLVER = {"hllc": HLLC, "hlle": HLLE, "llf": LLF}
RECON = {"pcm": PCM, "plm": PLM, "ppm": PPM}
BCS = {"periodic": BC_PERIODIC, "outflow": BC_OUTFLOW, "reflecting": BC_REFLECT,
       "reflect": BC_REFLECT, "none": BC_NONE, "extrap": BC_STRAT_EXTRAP,
       "inflow": BC_STRAT_INFLOW, "conductive": BC_CONDUCTIVE, "ic": BC_IC, "disk_extrap": BC_DISK_EXTRAP,
       "viscous": BC_DISK_VISC}

PP = C.c_void_p  # device pointer tables are opaque to the host


class FluidPack(C.Structure):
    _fields_ = [
        ("nspecies", C.c_int), ("recon", C.c_int), ("riemann", C.c_int),
        ("dfloor", C.c_double), ("siefloor", C.c_double), ("de_switch", C.c_double),
        ("prim", PP), ("cons0", PP), ("cons1", PP),
        ("flux", PP * 3), ("pflux", PP * 3), ("vface", PP * 3), ("diff_flux", PP * 3),
    ]


class Pack(C.Structure):
    _fields_ = [
        ("nblocks", C.c_int), ("nghost", C.c_int),
        ("nx1", C.c_int), ("nx2", C.c_int), ("nx3", C.c_int),
        ("coords", C.c_int), ("gm1", C.c_double), ("geom", C.c_void_p),
        ("metric", C.c_void_p),
        ("gas", FluidPack), ("dust", FluidPack), ("omega_frame", C.c_double),
        ("plm_table", C.c_void_p),
    ]


class StageArgs(C.Structure):
    _fields_ = [
        ("gam0", C.c_double), ("gam1", C.c_double), ("beta_dt", C.c_double), ("bdt", C.c_double),
        ("pcm", C.c_int),
        ("prim_in", PP), ("prim_u1", PP), ("prim_out", PP), ("cons_out", PP),
        ("cfl", C.c_double), ("dt_dev", C.c_void_p), ("region", C.c_int),
        ("shell_done", C.c_void_p), ("shell_target", C.POINTER(C.c_uint)),
        ("beta_dt_dev", C.c_void_p), ("shell_faces", C.c_int),
        ("tiny_in", C.c_void_p), ("tiny_out", C.c_void_p), ("tiny_clear", C.c_void_p),
        ("outflow_faces", C.c_int), ("outflow_faces_by_block", C.c_void_p), ("redo_scratch", C.c_void_p),
    ]


class BcParams(C.Structure):
    _fields_ = [("qshear", C.c_double), ("omega", C.c_double), ("cond_temp", C.c_double),
                ("cond_flux", C.c_double), ("cond_g", C.c_double * 3), ("cond_coeff", C.c_double),
                ("cond_cv", C.c_double), ("cond_type", C.c_int), ("cond_temp_exp", C.c_double),
                ("cond_rho_exp", C.c_double), ("cond_T_ref", C.c_double), ("cond_rho_ref", C.c_double),
                ("ic_gas", C.c_void_p),
                ("ic_dust", C.c_void_p), ("disk_omf", C.c_double), ("disk_nu0", C.c_double),
                ("disk_nu_indx", C.c_double), ("disk_r0", C.c_double), ("disk_mdot", C.c_double),
                ("floor_ghosts", C.c_int), ("x1_interior_done", C.c_int)]


class Cooling(C.Structure):
    _fields_ = [("beta0", C.c_double), ("beta_min", C.c_double), ("exp_scale", C.c_double),
                ("tfloor", C.c_double), ("tcyl", C.c_double), ("cyl_plaw", C.c_double), ("tsph", C.c_double),
                ("sph_plaw", C.c_double), ("gm", C.c_double), ("cv", C.c_double), ("tref", C.c_void_p),
                ("beta", C.c_void_p)]


class Gravity(C.Structure):
    _fields_ = [("type", C.c_int), ("g", C.c_double * 3), ("gm", C.c_double), ("soft", C.c_double),
                ("sink", C.c_double), ("sink_rate", C.c_double), ("pos", C.c_double * 3),
                ("tstart", C.c_double), ("tstop", C.c_double), ("q", C.c_double), ("soft2", C.c_double),
                ("sink2", C.c_double), ("sink_rate2", C.c_double), ("pos2", C.c_double * 3)]


class NBodyParticle(C.Structure):  # artemis_nbody_particle_t
    _fields_ = [("gm", C.c_double), ("pos", C.c_double * 3), ("vel", C.c_double * 3), ("xf", C.c_double * 3),
                ("vf", C.c_double * 3), ("rs", C.c_double), ("racc", C.c_double), ("gamma", C.c_double),
                ("beta", C.c_double), ("spline", C.c_int), ("couple", C.c_int)]


class Damping(C.Structure):
    _fields_ = [("ix", C.c_double * 3), ("ox", C.c_double * 3), ("irate", C.c_double * 3),
                ("orate", C.c_double * 3)]


class Drag(C.Structure):
    _fields_ = [("type", C.c_int), ("model", C.c_int), ("scale", C.c_double),
                ("grain_density", C.c_double), ("tau", C.c_double * MAX_DUST_SPECIES),
                ("sizes", C.c_double * MAX_DUST_SPECIES), ("gas", Damping), ("dust", Damping),
                ("xmin", C.c_double * 3), ("xmax", C.c_double * 3), ("damp_visc", C.c_void_p)]


class DiffCoeff(C.Structure):
    _fields_ = [("type", C.c_int), ("avg", C.c_int), ("coeff", C.c_double), ("eta", C.c_double),
                ("r_exp", C.c_double), ("r0", C.c_double), ("omega0", C.c_double),
                ("temp_exp", C.c_double), ("rho_exp", C.c_double), ("rho_ref", C.c_double),
                ("T_ref", C.c_double), ("radial", C.c_void_p)]


class Diffusion(C.Structure):
    _fields_ = [("visc", DiffCoeff), ("cond", DiffCoeff), ("cv", C.c_double), ("dist", C.c_void_p)]


class Refine(C.Structure):
    _fields_ = [("coords", C.c_int), ("ndim", C.c_int), ("nvar", C.c_int),
                ("fni", C.c_int), ("fnj", C.c_int), ("fnk", C.c_int), ("cni", C.c_int), ("cnj", C.c_int),
                ("cnk", C.c_int), ("fgeom", C.c_void_p), ("fmetric", C.c_void_p), ("cgeom", C.c_void_p),
                ("cmetric", C.c_void_p), ("fine", C.c_void_p), ("coarse", C.c_void_p),
                ("cis", C.c_int), ("cie", C.c_int), ("cjs", C.c_int), ("cje", C.c_int), ("cks", C.c_int),
                ("cke", C.c_int), ("cib", C.c_int), ("cjb", C.c_int), ("ckb", C.c_int), ("fib", C.c_int),
                ("fjb", C.c_int), ("fkb", C.c_int)]


class AmrCriterion(C.Structure):
    _fields_ = [("coords", C.c_int), ("ndim", C.c_int), ("ni", C.c_int), ("nj", C.c_int), ("nk", C.c_int),
                ("geom", C.c_void_p), ("metric", C.c_void_p), ("field", C.c_void_p),
                ("is_", C.c_int), ("ie", C.c_int), ("js", C.c_int), ("je", C.c_int), ("ks", C.c_int),
                ("ke", C.c_int), ("refine_thr", C.c_double), ("deref_thr", C.c_double), ("scratch", C.c_void_p)]


class MlFaceBox(C.Structure):  # artemis_ml_face_box_t
    _fields_ = [("block", C.c_int), ("dir", C.c_int), ("lo", C.c_int * 3), ("n", C.c_int * 3)]


class MlFixCell(C.Structure):  # artemis_ml_fix_cell_t
    _fields_ = [("block", C.c_int), ("k", C.c_int), ("j", C.c_int), ("i", C.c_int), ("faces", C.c_uint)]


class StageGeneralArgs(C.Structure):
    _fields_ = [
        ("gam0", C.c_double), ("gam1", C.c_double), ("beta_dt", C.c_double), ("bdt", C.c_double),
        ("pcm", C.c_int), ("time", C.c_double),
        ("gas_in", PP), ("gas_u1", PP), ("gas_out", PP),
        ("dust_in", PP), ("dust_u1", PP), ("dust_out", PP),
        ("gravity", C.POINTER(Gravity)), ("rf_omega", C.c_double), ("rf_qshear", C.c_double),
        ("drag", C.POINTER(Drag)), ("cfl_gas", C.c_double), ("cfl_dust", C.c_double),
        ("dt_dev", C.c_void_p), ("beta_dt_dev", C.c_void_p),
        ("diffusion", C.POINTER(Diffusion)), ("cooling", C.POINTER(Cooling)),
        ("diffusion_sums", PP),
        ("nbody_dev", C.c_void_p), ("nbody_n", C.c_int), ("nbody_omf", C.c_double),
        ("defer_finish", C.c_int),
        ("strat_faces", C.c_int), ("strat_qshear", C.c_double), ("strat_omega", C.c_double),
    ]


class ArtemisHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"artemis_hip error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """dlopen libartemis_hip.so (built by __graft_entry__.build()); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if os.path.isdir(os.path.join(_HERE, "csrc")) and not os.environ.get("ARTEMIS_NO_BUILD_CHECK"):
        # In a source checkout the library that is loaded IS the working tree: bring it up to date (a no-op when every
        # object's content hash matches; hipcc cross-compiles without a GPU) and refuse a binary that still differs.
        from . import build as _build
        _build.build_hip()
        _build.verify()
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950). No fallback exists.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    PPk, vp, d, i = C.POINTER(Pack), C.c_void_p, C.c_double, C.c_int
    sig = {
        "artemis_hip_calculate_fluxes": (i, [PPk, i, i, vp]),
        "artemis_hip_apply_update": (i, [PPk, d, d, d, vp]),
        "artemis_hip_flux_source": (i, [PPk, i, d, vp]),
        "artemis_hip_set_aux": (i, [PPk, vp]),
        "artemis_hip_cons_to_prim": (i, [PPk, vp]),
        "artemis_hip_prim_to_cons": (i, [PPk, vp]),
        "artemis_hip_prim_to_cons_ghosts": (i, [PPk, vp]),
        "artemis_hip_deep_copy_conserved": (i, [PPk, vp]),
        "artemis_hip_estimate_dt": (i, [PPk, i, d, C.POINTER(d), vp]),
        "artemis_hip_estimate_dt_async": (i, [PPk, i, d, vp, vp]),
        "artemis_hip_apply_bc": (i, [PPk, C.POINTER(i), C.POINTER(BcParams), vp]),
        "artemis_hip_external_gravity": (i, [PPk, C.POINTER(Gravity), d, d, vp]),
        "artemis_hip_rotating_frame_force": (i, [PPk, d, d, d, d, vp]),
        "artemis_hip_nbody_gravity": (i, [PPk, C.POINTER(NBodyParticle), i, d, d, d, C.POINTER(d), vp]),
        "artemis_hip_nbody_force_scratch": (i, [PPk]),
        "artemis_hip_redo_scratch_bytes": (C.c_size_t, [PPk]),
        "artemis_hip_timestep_all": (i, [PPk, d, d, C.POINTER(Diffusion), vp, vp]),
        "artemis_hip_nbody_force_sums": (i, [PPk, vp, i, d, d, vp, vp, vp, vp]),
        "artemis_hip_drag_source": (i, [PPk, C.POINTER(Drag), d, d, vp]),
        "artemis_hip_cooling_source": (i, [PPk, C.POINTER(Cooling), d, d, vp]),
        "artemis_hip_cooling_table_fill": (i, [PPk, vp, vp, C.POINTER(Cooling), i, vp, vp]),
        "artemis_hip_stage_fused": (i, [PPk, C.POINTER(StageArgs), vp]),
        "artemis_hip_stage_fused_redo_shell": (i, [PPk, C.POINTER(StageArgs), vp]),
        "artemis_hip_stage_general": (i, [PPk, C.POINTER(StageGeneralArgs), vp]),
        "artemis_hip_stage_general_variant": (i, [PPk, C.POINTER(StageGeneralArgs)]),
        "artemis_hip_stage_epilogue": (i, [PPk, C.POINTER(StageGeneralArgs), vp]),
        "artemis_hip_stage_epilogue_cons": (i, [PPk, C.POINTER(StageGeneralArgs), vp]),
        "artemis_hip_plm_table_count": (C.c_long, [PPk]),
        "artemis_hip_plm_table_fill": (i, [PPk, vp, vp]),
        "artemis_hip_ml_face_fluxes": (i, [PPk, C.POINTER(StageGeneralArgs), vp, i, vp]),
        "artemis_hip_ml_stage_fixup": (i, [PPk, C.POINTER(StageGeneralArgs), vp, i, vp]),
        "artemis_hip_stage_finish": (i, [PPk, C.POINTER(Drag), d, d, vp]),
        "artemis_hip_stage_finish_cells": (i, [PPk, C.POINTER(Drag), d, d, vp, i, vp]),
        "artemis_hip_restrict_average": (i, [C.POINTER(Refine), vp]),
        "artemis_hip_prolongate_minmod": (i, [C.POINTER(Refine), vp]),
        "artemis_hip_amr_first_derivative": (i, [C.POINTER(AmrCriterion), C.POINTER(C.c_int), C.POINTER(C.c_double), vp]),
        "artemis_hip_amr_block_maxima": (i, [PPk, C.c_int, C.c_int, vp, vp]),
        "artemis_hip_amr_magnitude": (i, [C.POINTER(AmrCriterion), C.POINTER(C.c_int), C.POINTER(C.c_double), vp]),
        "artemis_hip_zero_diffusion_flux": (i, [PPk, vp]),
        "artemis_hip_viscous_distance_count": (C.c_size_t, [PPk]),
        "artemis_hip_viscous_distance_fill": (i, [PPk, vp, vp]),
        "artemis_hip_viscous_flux": (i, [PPk, C.POINTER(Diffusion), vp]),
        "artemis_hip_zero_viscous_flux": (i, [PPk, C.POINTER(Diffusion), vp]),
        "artemis_hip_thermal_flux": (i, [PPk, C.POINTER(Diffusion), vp]),
        "artemis_hip_diffusion_update": (i, [PPk, C.POINTER(Diffusion), d, vp]),
        "artemis_hip_viscous_source_covers": (i, [PPk]),
        "artemis_hip_ml_viscous_faces": (i, [PPk, C.POINTER(Diffusion), vp, i, vp, i, vp]),
        "artemis_hip_viscous_source": (i, [PPk, C.POINTER(Diffusion), d, vp, PP, vp]),
        "artemis_hip_diffusion_dt": (i, [PPk, C.POINTER(Diffusion), d, vp, vp]),
        "artemis_hip_diffusion_radial_fill": (i, [PPk, vp, vp, C.POINTER(DiffCoeff), i, vp]),
        "artemis_hip_wait_counter": (i, [vp, C.c_uint, vp, vp]),
        "artemis_hip_advance_dt": (i, [vp, d, i, C.POINTER(d), vp]),
        "artemis_hip_metric_count": (C.c_long, [PPk]),
        "artemis_hip_metric_fill": (i, [PPk, vp, vp]),
        "artemis_hip_halo_count": (C.c_long, [PPk, i]),
        "artemis_hip_halo_count_ext": (C.c_long, [PPk, i, i]),
        "artemis_hip_halo_pack_ext": (i, [PPk, i, i, i, vp, vp]),
        "artemis_hip_halo_unpack_ext": (i, [PPk, i, i, i, vp, vp]),
        "artemis_hip_halo_pack": (i, [PPk, i, i, vp, vp]),
        "artemis_hip_halo_unpack": (i, [PPk, i, i, vp, vp]),
        "artemis_hip_selftest_divsqrt": (i, [C.c_long, vp, vp, vp, vp, vp, vp, vp]),
        "artemis_hip_last_error": (C.c_char_p, []),
        "artemis_hip_device_count": (i, []),
        "artemis_hip_version": (C.c_char_p, []),
        "artemis_hip_source_sha": (C.c_char_p, []),
        "artemis_hip_object_sha": (C.c_char_p, [C.c_char_p]),
        "artemis_rt_set_device": (i, [i]),
        "artemis_rt_malloc": (vp, [C.c_size_t]),
        "artemis_rt_free": (None, [vp]),
        "artemis_rt_device_bytes": (None, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), i]),
        "artemis_rt_pool_trim": (None, [C.c_size_t]),
        "artemis_rt_pool_bytes": (C.c_size_t, []),
        "artemis_rt_pool_limit": (None, [C.c_size_t]),
        "artemis_hip_set_option": (C.c_int, [C.c_char_p, C.c_long]),
        "artemis_hip_get_option": (C.c_long, [C.c_char_p]),
        "artemis_rt_malloc_host": (vp, [C.c_size_t]),
        "artemis_rt_free_host": (None, [vp]),
        "artemis_rt_memcpy_h2d": (i, [vp, vp, C.c_size_t, vp]),
        "artemis_rt_memcpy_d2h": (i, [vp, vp, C.c_size_t, vp]),
        "artemis_rt_memcpy_d2d": (i, [vp, vp, C.c_size_t, vp]),
        "artemis_rt_memset": (i, [vp, i, C.c_size_t, vp]),
        "artemis_rt_stream_create": (vp, []),
        "artemis_rt_stream_destroy": (None, [vp]),
        "artemis_rt_stream_sync": (i, [vp]),
        "artemis_rt_device_sync": (i, []),
        "artemis_rt_event_create": (vp, []),
        "artemis_rt_event_destroy": (None, [vp]),
        "artemis_rt_event_record": (i, [vp, vp]),
        "artemis_rt_stream_wait_event": (i, [vp, vp]),
        "artemis_rt_event_sync": (i, [vp]),
        "artemis_rt_event_elapsed_ms": (d, [vp, vp]),
        "artemis_rt_tables_changed": (None, []),
        "artemis_rt_capture_begin": (i, [vp]),
        "artemis_rt_capture_end": (vp, [vp]),
        "artemis_rt_graph_launch": (i, [vp, vp]),
        "artemis_rt_graph_destroy": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the header and the library disagree
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


EXPORTS_HIP = [
    "artemis_hip_calculate_fluxes", "artemis_hip_apply_update", "artemis_hip_flux_source",
    "artemis_hip_set_aux", "artemis_hip_cons_to_prim", "artemis_hip_prim_to_cons", "artemis_hip_prim_to_cons_ghosts",
    "artemis_hip_deep_copy_conserved", "artemis_hip_estimate_dt", "artemis_hip_estimate_dt_async",
    "artemis_hip_apply_bc", "artemis_hip_stage_fused", "artemis_hip_stage_fused_redo_shell", "artemis_hip_metric_count",
    "artemis_hip_metric_fill", "artemis_hip_external_gravity", "artemis_hip_nbody_gravity", "artemis_hip_nbody_force_scratch",
    "artemis_hip_nbody_force_sums", "artemis_hip_timestep_all", "artemis_hip_redo_scratch_bytes", "artemis_hip_rotating_frame_force",
    "artemis_hip_ml_exchange", "artemis_hip_ml_flux_correction", "artemis_hip_ml_restrict_halos", "artemis_hip_ml_prolongate",
    "artemis_hip_ml_floor_ghosts",
    "artemis_hip_ml_face_fluxes", "artemis_hip_ml_stage_fixup", "artemis_hip_plm_table_count", "artemis_hip_plm_table_fill",
    "artemis_hip_drag_source", "artemis_hip_cooling_source", "artemis_hip_cooling_table_fill",
    "artemis_hip_stage_general", "artemis_hip_stage_general_variant", "artemis_hip_stage_epilogue", "artemis_hip_stage_epilogue_cons", "artemis_hip_stage_finish", "artemis_hip_stage_finish_cells", "artemis_hip_amr_block_maxima", "artemis_hip_restrict_average",
    "artemis_hip_prolongate_minmod", "artemis_hip_amr_first_derivative", "artemis_hip_amr_magnitude",
    "artemis_hip_zero_diffusion_flux", "artemis_hip_viscous_distance_count", "artemis_hip_viscous_distance_fill",
    "artemis_hip_viscous_flux", "artemis_hip_zero_viscous_flux", "artemis_hip_thermal_flux", "artemis_hip_diffusion_update",
    "artemis_hip_viscous_source_covers", "artemis_hip_viscous_source", "artemis_hip_ml_viscous_faces",
    "artemis_hip_diffusion_dt", "artemis_hip_diffusion_radial_fill", "artemis_hip_halo_count", "artemis_hip_halo_count_ext",
    "artemis_hip_halo_pack_ext", "artemis_hip_halo_unpack_ext",
    "artemis_hip_halo_pack", "artemis_hip_halo_unpack", "artemis_hip_last_error",
    "artemis_hip_device_count", "artemis_hip_version", "artemis_hip_source_sha", "artemis_hip_object_sha",
]


def source_sha():
    """sha1 of the sources the loaded library was built from (== artemis_amd.build.tree_sha() in a checkout)."""
    return load().artemis_hip_source_sha().decode()


def object_sha(unit):
    """Content hash of one translation unit of the loaded library, e.g. "kernels_fused" (None: no such unit)."""
    v = load().artemis_hip_object_sha(unit.encode())
    return v.decode() if v else None


def check(rc):
    if rc != 0:
        raise ArtemisHipError(rc, load().artemis_hip_last_error().decode())


import contextlib


@contextlib.contextmanager
def option(name, value=1):
    """`with capi.option("no_fused_curv"):` -- one of the library's switches (include/artemis_hip.h) set for the block
    and restored after it (the table is read from the environment only once, when the library first needs it)."""
    L = load()
    old = L.artemis_hip_get_option(name.encode())
    if old < 0:
        raise KeyError(name)
    check(L.artemis_hip_set_option(name.encode(), int(value)))
    try:
        yield
    finally:
        L.artemis_hip_set_option(name.encode(), old)
