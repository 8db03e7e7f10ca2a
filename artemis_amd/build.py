"""Build recipe for the HIP library and the host driver (in-tree, no JIT cache)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.environ.get("HIPCC", os.path.join(ROCM, "bin", "hipcc"))

# -ffp-contract=off: every expression tree stays unfused so results are bit-identical to the
# CPU oracle (oracle/Makefile uses the same flag).  See DESIGN.md "Floating point".
# -fno-builtin-sin/-cos (host code): sin(x) and cos(x) of one argument stay two glibc calls instead of one
# sincos() (the CPU checker under tests/ is built the same way) -- the metric tables and the pgens' trigonometry then cannot depend
# on which compiler merged what (DESIGN.md "Stated tolerance").
NO_SINCOS = ["-fno-builtin-sin", "-fno-builtin-cos"]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
             "-Wall", "-Wno-unused-function"]

HIP_SOURCES = ["abi.hip", "kernels_unfused.hip", "kernels_fused.hip", "kernels_sources.hip", "kernels_stage_cell.hip",
               "kernels_diffusion.hip", "kernels_refine.hip", "kernels_amr.hip", "kernels_stage2d.hip", "kernels_curv.hip",
               "selftest.hip"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps():
    out = []
    for d in (CSRC, os.path.join(CSRC, "driver"), os.path.join(ROOT, "include")):
        if os.path.isdir(d):
            out += [os.path.join(d, f) for f in os.listdir(d)
                    if f.endswith((".hip", ".hpp", ".h", ".cpp"))]
    return out


def build_hip(force=False, verbose=False, only=None):
    """only = ["kernels_curv.hip", ...] (development): recompile just these sources and relink with the objects of the
    last full build (the caller knows that nothing else depends on what changed)."""
    os.makedirs(LIBDIR, exist_ok=True)
    target = os.path.join(LIBDIR, "libartemis_hip.so")
    if not (force or only or _newer(target, _deps())):
        return target
    objs = []
    procs = []
    for src in HIP_SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        if only and src not in only and os.path.exists(obj):
            objs.append(obj)
            continue
        # ARTEMIS_HIPFLAGS_<STEM> (e.g. ARTEMIS_HIPFLAGS_KERNELS_FUSED="-mllvm -amdgpu-sched-strategy=max-ilp"):
        # extra flags for one source, for compiler experiments
        extra = os.environ.get("ARTEMIS_HIPFLAGS_" + src.replace(".hip", "").upper(), "").split()
        cmd = [HIPCC] + HIP_FLAGS + extra + (NO_SINCOS if src == "abi.hip" else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    drv_dir = os.path.join(CSRC, "driver")
    if os.path.isdir(drv_dir):
        for f in sorted(os.listdir(drv_dir)):
            if f.endswith(".cpp"):
                obj = os.path.join(LIBDIR, "driver_" + f.replace(".cpp", ".o"))
                if only and f not in only and os.path.exists(obj):
                    objs.append(obj)
                    continue
                cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-ffp-contract=off"] + NO_SINCOS + [
                       "-I", os.path.join(ROOT, "include"), "-c", os.path.join(drv_dir, f), "-o", obj]
                if f == "comm_rccl.cpp":  # rccl.h pulls in the HIP runtime API header (types only: no HIP calls there)
                    cmd += ["-I", os.path.join(ROCM, "include"), "-D__HIP_PLATFORM_AMD__", "-Wno-deprecated-declarations"]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-o", target] + objs + [
        "-L", os.path.join(ROCM, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return target


if __name__ == "__main__":
    print(build_hip(force="-f" in sys.argv, verbose=True, only=[a for a in sys.argv[1:] if not a.startswith("-")] or None))
