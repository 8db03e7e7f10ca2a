import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cpus():
    """CPUs this process may really use: affinity mask clipped by a cgroup CPU quota (the GPU box shows 256
    logical CPUs but grants 16 CPUs of time -- an OpenMP team of 256 threads then crawls)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return n


# the CPU oracle (OpenMP) is the checker of nearly every test: size its thread team to the real CPU budget
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, _usable_cpus())))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def hiplib():
    """The product library; GPU tests must exercise it (never a fallback)."""
    import torch
    from artemis_amd import capi
    L = capi.load()
    assert torch.cuda.is_available(), "GPU test on a box without a GPU"
    assert L.artemis_hip_device_count() >= 1
    return L


@pytest.fixture
def one_openmp_thread():
    """For 1-D problems: a short row is no work for a thread team (the fork / join of every sweep costs 100x the
    arithmetic on 8 threads).  Process-wide OpenMP setting, restored afterwards."""
    import ctypes
    g = ctypes.CDLL("libgomp.so.1")
    n = g.omp_get_max_threads()
    g.omp_set_num_threads(1)
    yield
    g.omp_set_num_threads(n)


OPTION_NAMES = ["NO_TUNED", "TUNED_2D", "NO_STAGE2D", "NO_FUSED_CURV", "NO_CURV_MARCH", "NO_CURV_DUST", "NO_CURV_DUST_MARCH", "NO_DRAG_IN_MARCH", "NO_STRAT_IN_KERNEL", "NO_CART_MARCH", "NO_IC_IN_SHELL", "NO_IC_SKIP", "NO_ML_FLOOR",
                "NO_ML_FUSED", "NO_EPILOGUE", "NO_TILED_FLUX", "NO_VISC_SOURCE", "NBODY_TASK", "NBODY_GENERAL", "NO_PLM_TABLE",
                "NO_DISTANCE_TABLE", "NO_FLAT_RANGES", "FULL_REMESH", "NO_REDO", "NO_TINY_HINT", "NO_GRAPH", "SYNC_LOOP",
                "FORCE_OVERLAP", "LOOPBACK_COMM", "WAIT_SPIN_LIMIT", "TEST_SHELL_TARGET_BUMP", "HOST_THREADS", "SETUP_TIMING",
                "AMR_DEBUG", "FUSED_KCHUNK", "CURV_KCHUNK", "VISC_KCHUNK", "STAGE2D_ROWS", "STAGE2D_RGRID", "FUSED_NO_SWIZZLE",
                "NO_POOL", "POOL_GB", "TRIM_POOL", "POISON", "DENSE_FLUX"]


@pytest.fixture(autouse=True)
def _library_options_restored():
    """The library's switches (artemis_hip_set_option) are process-wide: whatever a test sets is put back after it."""
    from artemis_amd import capi
    # (never load the library on a test's behalf: it is opened RTLD_GLOBAL, and a test that runs the CPU double in this
    #  process -- tests/test_adaptive_oracle.py -- must not find the product's symbols ahead of the double's own)
    L = capi._lib
    before = {n: L.artemis_hip_get_option(n.encode()) for n in OPTION_NAMES} if L is not None else {}
    yield
    L = capi._lib
    if L is None:
        return
    for n in OPTION_NAMES:
        v = before.get(n, 0)  # (loaded during the test: every switch starts at 0)
        if v >= 0:
            L.artemis_hip_set_option(n.encode(), v)


@pytest.fixture
def option():
    """option("no_redo") / option("visc_kchunk", 5): set one of the library's switches for the rest of the test."""
    from artemis_amd import capi

    def set_(name, value=1):
        capi.check(capi.load().artemis_hip_set_option(name.encode(), int(value)))
    return set_
