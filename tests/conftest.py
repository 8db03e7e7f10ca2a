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
