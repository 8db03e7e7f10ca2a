"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/artemis_hip.h and include/artemis_rt.h declare; argument validation mirrors the
reference's PARTHENON_FAIL guards; without a GPU every compute call fails loudly."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(artemis_(?:hip|rt|sim|comm)_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from artemis_amd import capi
    L = capi.load()
    names = _declared("artemis_hip.h") + _declared("artemis_rt.h")
    assert len(names) > 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ but not exported"
    assert set(capi.EXPORTS_HIP) <= set(names)
    sim = _declared("artemis_driver.h")  # the host driver's own ABI lives in the same library
    assert len(sim) > 10
    for n in sim:
        assert hasattr(L, n), f"{n} declared in include/artemis_driver.h but not exported"


def test_the_binary_is_the_working_tree():
    """Every object of lib/ carries the content hash (source + the headers it reaches + command line + compiler) of the
    working tree, and the identities compiled into libartemis_hip.so equal them: what is loaded, tested and profiled can
    be rebuilt from this checkout (round 5 shipped one object of an uncommitted experiment; mtimes could not see it)."""
    from artemis_amd import build, capi
    L = capi.load()
    assert build.stale_objects() == []
    assert capi.source_sha() == build.tree_sha()
    want = build.object_hashes()
    assert set(want) >= {"abi", "kernels_fused", "kernels_amr", "kernels_curv", "driver_driver", "driver_comm_rccl"}
    for unit, sha in want.items():
        assert capi.object_sha(unit) == sha, unit
    assert capi.object_sha("no_such_unit") is None
    assert build.verify(L) == build.tree_sha()


def test_object_hashes_do_not_depend_on_how_the_tree_is_reached(tmp_path):
    """The GPU boxes reach the checkout through a symlink (/root/repo -> a scratch directory).  An object hash that saw the
    link in one place and its target in another called every object stale there, and every box recompiled an up-to-date
    library (round 6, four minutes of each call): through a symlinked root nothing is stale and the hashes are the same."""
    import subprocess
    import sys
    from artemis_amd import build
    link = tmp_path / "checkout"
    os.symlink(ROOT, link)
    code = ("import sys, json; sys.path.insert(0, %r); from artemis_amd import build; "
            "print(json.dumps([build.stale_objects(), build.object_hashes(), build.tree_sha()]))" % str(link))
    out = subprocess.check_output([sys.executable, "-c", code], cwd=str(tmp_path)).decode().strip().splitlines()[-1]
    import json
    stale, hashes, tree = json.loads(out)
    assert stale == [] and hashes == build.object_hashes() and tree == build.tree_sha()


def test_a_touched_header_makes_exactly_its_users_stale(tmp_path, monkeypatch):
    """The hash follows the include graph: a change in kernels_amr.hip concerns that unit alone, one in
    device_math.hpp every kernel file that reaches it and not the host driver; a reverted edit leaves nothing stale."""
    from artemis_amd import build
    base = build.object_hashes()
    real_closure = build._closure

    def edited(which):
        def closure(path, seen=None):
            out = real_closure(path, seen)
            for p in list(out):
                if os.path.basename(p) == which:
                    out[p] = out[p] + b"// edit\n"
            return out
        return closure
    monkeypatch.setattr(build, "_closure", edited("kernels_amr.hip"))
    now = build.object_hashes()
    assert [u for u in base if base[u] != now[u]] == ["kernels_amr"]
    monkeypatch.setattr(build, "_closure", edited("device_math.hpp"))
    now = build.object_hashes()
    changed = {u for u in base if base[u] != now[u]}
    assert {"kernels_fused", "kernels_curv", "kernels_unfused"} <= changed and not any(u.startswith("driver_") for u in changed)
    monkeypatch.setattr(build, "_closure", real_closure)
    assert build.object_hashes() == base
    # only= refuses to link when something outside the list is stale
    monkeypatch.setattr(build, "_closure", edited("device_math.hpp"))
    with pytest.raises(RuntimeError, match="stale too"):
        build.build_hip(only=["kernels_amr.hip"])


def test_no_gpu_fails_loudly_and_validation():
    import torch
    from artemis_amd import capi
    L = capi.load()
    p = capi.Pack()
    # null / malformed packs are EINVAL regardless of hardware
    assert L.artemis_hip_calculate_fluxes(None, 0, 0, None) == capi.EINVAL
    p.nblocks, p.nghost, p.nx1, p.nx2, p.nx3 = 1, 2, 8, 1, 8
    assert L.artemis_hip_calculate_fluxes(C.byref(p), 0, 0, None) == capi.EINVAL
    p.nx2 = 8
    p.coords = 9
    assert L.artemis_hip_set_aux(C.byref(p), None) == capi.EINVAL
    assert b"Coordinate type not recognized" in L.artemis_hip_last_error()
    p.coords = 4  # spherical3D needs the x2 metric tables (artemis_hip_metric_fill)
    assert L.artemis_hip_set_aux(C.byref(p), None) == capi.EINVAL
    assert b"metric" in L.artemis_hip_last_error()
    p.coords = 3  # spherical2D on a 3-D block: geometry::CoordSelect never produces that
    assert L.artemis_hip_set_aux(C.byref(p), None) == capi.EINVAL
    assert L.artemis_hip_metric_count(C.byref(p)) == 6 * (8 + 4 + 1) + 2 * (8 + 4 + 1)
    p.coords = 0
    p.geom = 1
    p.gas.nspecies = 1
    p.gm1 = 0.4
    if not torch.cuda.is_available():
        # a well-formed call on a box without a GPU must not silently succeed
        rc = L.artemis_hip_prim_to_cons(C.byref(p), None)
        assert rc == capi.EDEVICE
        assert b"no CPU fallback" in L.artemis_hip_last_error()
        assert L.artemis_rt_malloc(64) is None


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under artemis_amd/ may reference it."""
    for dp, _, files in os.walk(os.path.join(ROOT, "artemis_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower().replace("cpu oracle (oracle/makefile", ""), \
                    f"{os.path.join(dp, f)} mentions the oracle"


def test_ctypes_mirror_matches_the_c_header(tmp_path):
    """The ctypes structures of artemis_amd/capi.py against the C compiler's view of include/artemis_hip.h:
    same size for every struct that crosses the boundary and the same offset for the last member
    (a missing or mistyped field in the middle shifts it)."""
    import subprocess
    from artemis_amd import capi
    pairs = [("artemis_fluid_pack_t", capi.FluidPack, None), ("artemis_pack_t", capi.Pack, "omega_frame"),
             ("artemis_bc_params_t", capi.BcParams, "floor_ghosts"), ("artemis_gravity_t", capi.Gravity, "pos2"),
             ("artemis_damping_t", capi.Damping, None), ("artemis_drag_t", capi.Drag, "damp_visc"),
             ("artemis_cooling_t", capi.Cooling, "beta"), ("artemis_diffcoeff_t", capi.DiffCoeff, "radial"),
             ("artemis_diffusion_t", capi.Diffusion, "cv"), ("artemis_stage_args_t", capi.StageArgs, None),
             ("artemis_stage_general_args_t", capi.StageGeneralArgs, "cooling"),
             ("artemis_refine_t", capi.Refine, "fkb"),
             ("artemis_amr_criterion_t", capi.AmrCriterion, "scratch"),
             ("artemis_nbody_particle_t", capi.NBodyParticle, "couple")]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "artemis_hip.h"', 'int main(void) {']
    for cname, _, last in pairs:
        src.append(f'  printf("{cname} %zu %zu\\n", sizeof({cname}), '
                   + (f'offsetof({cname}, {last}));' if last else '(size_t)0);'))
    src += ['  return 0;', '}']
    c = tmp_path / "sizes.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    seen = {ln.split()[0]: (int(ln.split()[1]), int(ln.split()[2])) for ln in out if ln.strip()}
    for cname, ct, last in pairs:
        assert seen[cname][0] == C.sizeof(ct), (cname, seen[cname][0], C.sizeof(ct))
        if last:
            assert seen[cname][1] == getattr(ct, last).offset, (cname, last)


def test_metric_tables_are_plain_glibc_sin_cos():
    """artemis_hip_metric_fill (host code, no GPU needed) against Python's math.sin / math.cos, which call
    glibc's sin() and cos() one at a time: BITWISE.  All host code of the repo (this library, the driver, the
    oracle) is built with -fno-builtin-sin/-cos so no compiler merges a sin + cos pair into sincos(); if a
    toolchain ever breaks that, this test reports the ulp distance (the stated bound is 4 ulp)."""
    import math
    import numpy as np
    from artemis_amd import capi
    L = capi.load()
    p = capi.Pack()
    p.nblocks, p.nghost, p.nx1, p.nx2, p.nx3 = 1, 2, 8, 12, 6
    p.coords = capi.SPHERICAL3D
    geom = np.array([0.2, 0.1, 0.31, 0.2, -1.0, 0.41])
    n = L.artemis_hip_metric_count(C.byref(p))
    nj, nk = 16, 10
    assert n == 6 * (nj + 1) + 2 * (nk + 1)
    m = np.zeros(n)
    capi.check(L.artemis_hip_metric_fill(C.byref(p), geom.ctypes.data, m.ctypes.data))
    st = nj + 1
    want = np.zeros(n)
    for j in range(nj + 1):
        xf = geom[2] + j * geom[3]
        want[0 * st + j], want[1 * st + j] = math.cos(xf), math.sin(xf)
    for j in range(nj):  # spherical.hpp:61-68 x2v and the sines / cosine the kernels read
        x0, x1 = geom[2] + j * geom[3], geom[2] + (j + 1) * geom[3]
        ctm, ctp = want[j], want[j + 1]
        x2v = ((want[st + j + 1] - want[st + j]) - x1 * ctp + x0 * ctm) / abs(ctm - ctp)
        want[2 * st + j], want[3 * st + j] = x2v, math.sin(x2v)
        want[4 * st + j], want[5 * st + j] = math.sin(0.5 * (x0 + x1)), math.cos(x2v)
    for k in range(nk):
        x3v = 0.5 * ((geom[4] + k * geom[5]) + (geom[4] + (k + 1) * geom[5]))
        want[6 * st + k], want[6 * st + (nk + 1) + k] = math.cos(x3v), math.sin(x3v)
    if not np.array_equal(m, want):
        ulp = np.abs(m - want) / np.spacing(np.maximum(np.abs(want), 1e-300))
        assert False, "metric tables differ from glibc sin / cos by up to %.1f ulp" % ulp.max()


@pytest.mark.gpu
def test_device_buffer_cache_contract():
    """artemis_rt_malloc / artemis_rt_free (include/artemis_rt.h): a freed buffer comes back for a request of its size
    class (a remesh frees and re-requests tens of GB whose sizes barely change), a fresh buffer carries headroom so that a
    slightly larger request right after still fits it, the footprint counts cached buffers, and a trim gives them back.
    The cache is opt-in (artemis_rt_pool_limit; the standalone driver enables it for adaptive meshes): without it a free
    returns the memory to the device at once -- a library host shares the device with other allocators."""
    from artemis_amd import capi
    L = capi.load()
    cur, peak = C.c_size_t(), C.c_size_t()

    def footprint():
        L.artemis_rt_device_bytes(C.byref(cur), C.byref(peak), 0)
        return cur.value

    L.artemis_rt_pool_limit(0)
    base = footprint()
    n = 200 << 20
    a = L.artemis_rt_malloc(n)
    assert a and footprint() - base == n            # cache off: exact size, no size class
    L.artemis_rt_free(a)
    assert footprint() == base                      # ... and a free gives the memory back
    L.artemis_rt_pool_limit(8 << 30)
    a = L.artemis_rt_malloc(n)
    assert a and footprint() - base >= n
    held = footprint()
    L.artemis_rt_free(a)
    assert footprint() == held                     # cached, not returned to the device
    b = L.artemis_rt_malloc(n + (n >> 9))          # 0.2 % larger: the same buffer (3 % of headroom on a fresh allocation)
    assert b == a and footprint() == held
    c = L.artemis_rt_malloc(n)                      # a second live buffer is a new allocation
    assert c and c != b and footprint() > held
    L.artemis_rt_free(b), L.artemis_rt_free(c)
    d = L.artemis_rt_malloc(n - (n >> 4))           # 6 % smaller: a cached buffer of up to half as much again serves
    assert d in (a, c)
    L.artemis_rt_free(d)
    L.artemis_rt_pool_trim(0)
    assert footprint() == base
    L.artemis_rt_pool_limit(0)


def test_option_table_contract():
    """artemis_hip_set_option / artemis_hip_get_option (csrc/options.hpp): every switch include/artemis_hip.h documents is a
    name of the table, names are case-insensitive with or without the ARTEMIS_ prefix, unknown names are refused (EINVAL /
    -1), and the table is filled from ARTEMIS_<NAME> once, at first use -- checked in a child process so that this
    process's table (and library) stay as they are."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, os, sys
sys.path.insert(0, %r)
os.environ["ARTEMIS_VISC_KCHUNK"] = "7"
os.environ["ARTEMIS_NO_REDO"] = "yes"
from artemis_amd import capi
L = C.CDLL(capi.LIB_PATH)
L.artemis_hip_get_option.restype = C.c_long
L.artemis_hip_set_option.argtypes = [C.c_char_p, C.c_long]
assert L.artemis_hip_get_option(b"VISC_KCHUNK") == 7 and L.artemis_hip_get_option(b"artemis_visc_kchunk") == 7
assert L.artemis_hip_get_option(b"no_redo") == 1          # set, but not a number: a plain switch
assert L.artemis_hip_get_option(b"NO_GRAPH") == 0
os.environ["ARTEMIS_NO_GRAPH"] = "1"                      # (the environment is read once)
assert L.artemis_hip_get_option(b"NO_GRAPH") == 0
assert L.artemis_hip_set_option(b"no_graph", 1) == 0 and L.artemis_hip_get_option(b"NO_GRAPH") == 1
assert L.artemis_hip_set_option(b"no_such_switch", 1) != 0 and L.artemis_hip_get_option(b"no_such_switch") == -1
import re
doc = open(os.path.join(%r, "include", "artemis_hip.h")).read()
head = doc[:doc.index("#ifndef ARTEMIS_HIP_H_")]
names = set(re.findall(r"\b([A-Z][A-Z0-9_]{3,})\b", head)) - {"ARTEMIS", "NULL", "EINVAL", "ARTEMIS_HIP_EINVAL", "NAME"}
known = [n for n in names if L.artemis_hip_get_option(n.encode()) >= 0]
assert len(known) >= 30, sorted(known)
print("ok", len(known))
""" % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout[-500:], r.stderr[-1500:])


def test_object_hashes_do_not_depend_on_a_profilers_environment(monkeypatch):
    """Under `rocprofv3 -- python3 bench.py` every child process inherits the tool's LD_PRELOAD and writes its log lines to
    stderr; `hipcc --version` -- part of every object's hash -- must not see either (it did: each profiled process found
    the whole library stale and recompiled it)."""
    from artemis_amd import build
    build._versions.clear()
    want = build.object_hashes()
    monkeypatch.setenv("LD_PRELOAD", "/nonexistent/librocprofiler-sdk-tool.so")  # (the loader complains on stderr in every child)
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/nonexistent/librocprofiler-sdk-tool.so")
    build._versions.clear()
    try:
        assert build.object_hashes() == want
        assert "LD_PRELOAD" not in build._tool_env() and "ROCP_TOOL_LIBRARIES" not in build._tool_env()
    finally:
        build._versions.clear()
