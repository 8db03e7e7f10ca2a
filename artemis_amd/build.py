"""Build recipe for the HIP library and the host driver (in-tree, no JIT cache).

The binary is the tree, as a checked fact:

* every object is keyed by a **content hash** -- sha256 over its source, the transitive closure of the quoted headers
  it includes (csrc/, csrc/driver/, include/), its exact command line and the compiler's version string -- stored next
  to the object (`lib/<name>.o.sha256`).  An object is rebuilt when, and only when, that hash differs: there is no
  mtime logic, and a source that is edited and then reverted cannot leave a stale object behind;
* the hashes are compiled into the library (`lib/source_id.cpp`, generated): `artemis_hip_source_sha()` returns the
  sha1 over every file of csrc/** and include/** plus the flags, `artemis_hip_object_sha(name)` the hash an object was
  built from.  `verify()` / tests/test_capi_load.py compare them with the working tree; bench.py and the PMC scripts
  key their records on them.
"""
import fcntl
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
DRV = os.path.join(CSRC, "driver")
INC = os.path.join(ROOT, "include")
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
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-Wall", "-ffp-contract=off"] + NO_SINCOS

HIP_SOURCES = ["abi.hip", "kernels_unfused.hip", "kernels_fused.hip", "kernels_sources.hip", "kernels_stage_cell.hip",
               "kernels_diffusion.hip", "kernels_refine.hip", "kernels_amr.hip", "kernels_stage2d.hip", "kernels_curv.hip",
               "selftest.hip"]

_INCLUDE_RE = re.compile(r'^\s*#\s*include\s*"([^"]+)"', re.M)
_versions = {}


def _tool_env():
    """The environment compilers run in: this process's without a profiler's hooks.  Under `rocprofv3 -- python3 ...` every
    child inherits the tool's LD_PRELOAD and logs timestamped lines to stderr -- `hipcc --version` then differed from call
    to call, every object looked stale and each profiled process recompiled the whole library (minutes per PMC pass)."""
    drop = ("LD_PRELOAD", "ROCP", "ROCPROF", "HSA_TOOLS", "ROCTX", "ROCTRACER")
    return {k: v for k, v in os.environ.items() if not k.startswith(drop)}


def _compiler_version(exe):
    if exe not in _versions:
        _versions[exe] = subprocess.run([exe, "--version"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                        env=_tool_env()).stdout.decode()
    return _versions[exe]


def _closure(path, seen=None):
    """The file and every quoted header it reaches (searched like the compiler does: next to the including file, then
    csrc/, csrc/driver/, include/).  System headers (<...>) belong to the image and are covered by the compiler version."""
    seen = {} if seen is None else seen
    # (abspath, not realpath: the GPU boxes reach the tree through a symlink, and a hash that saw the link's target here
    #  and the link there made every box recompile a library that was up to date)
    path = os.path.abspath(path)
    if path in seen:
        return seen
    txt = open(path, "rb").read()
    seen[path] = txt
    for inc in _INCLUDE_RE.findall(txt.decode(errors="ignore")):
        for d in (os.path.dirname(path), CSRC, DRV, INC):
            cand = os.path.join(d, inc)
            if os.path.isfile(cand):
                _closure(cand, seen)
                break
    return seen


def _hash_files(h, files):
    for p in sorted(files):
        h.update(os.path.relpath(p, ROOT).encode() + b"\0")
        h.update(files[p] if isinstance(files, dict) else open(p, "rb").read())
        h.update(b"\0")


def _units():
    """[(name, source path, object path, command without -c/-o)] for every translation unit of the library."""
    out = []
    for src in HIP_SOURCES:
        # ARTEMIS_HIPFLAGS_<STEM> (e.g. ARTEMIS_HIPFLAGS_KERNELS_FUSED="-mllvm -amdgpu-sched-strategy=max-ilp"):
        # extra flags for one source, for compiler experiments (part of the object's hash like every other flag)
        extra = os.environ.get("ARTEMIS_HIPFLAGS_" + src.replace(".hip", "").upper(), "").split()
        cmd = [HIPCC] + HIP_FLAGS + extra + (NO_SINCOS if src == "abi.hip" else [])
        out.append((src.replace(".hip", ""), os.path.join(CSRC, src), os.path.join(LIBDIR, src.replace(".hip", ".o")), cmd))
    for f in sorted(os.listdir(DRV)):
        if f.endswith(".cpp"):
            cmd = ["g++"] + CXX_FLAGS + ["-I", INC]
            if f == "comm_rccl.cpp":  # rccl.h pulls in the HIP runtime API header (types only: no HIP calls there)
                cmd += ["-I", os.path.join(ROCM, "include"), "-D__HIP_PLATFORM_AMD__", "-Wno-deprecated-declarations"]
            out.append(("driver_" + f.replace(".cpp", ""), os.path.join(DRV, f), os.path.join(LIBDIR, "driver_" + f.replace(".cpp", ".o")), cmd))
    return out


def object_hashes():
    """{unit name: sha256 hex} from the working tree: what each object of an up-to-date build was compiled from."""
    out = {}
    for name, src, _, cmd in _units():
        h = hashlib.sha256()
        h.update((" ".join(os.path.relpath(c, ROOT) if os.path.isabs(c) and c.startswith(ROOT) else c for c in cmd)).encode() + b"\0")
        h.update(_compiler_version(cmd[0]).encode() + b"\0")
        _hash_files(h, _closure(src))
        out[name] = h.hexdigest()
    return out


def tree_sha():
    """sha1 over every source of the library (csrc/**, include/**) and the flags: the identity `artemis_hip_source_sha()`
    reports for the library that was built from this tree."""
    files = []
    for d in (CSRC, DRV, INC):
        files += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".hpp", ".h", ".cpp"))]
    h = hashlib.sha1()
    h.update(" ".join(HIP_FLAGS + ["|"] + CXX_FLAGS).encode() + b"\0")
    _hash_files(h, files)
    return h.hexdigest()


def _recorded(obj):
    try:
        return open(obj + ".sha256").read().strip()
    except OSError:
        return None


def stale_objects():
    """Names of the units whose object is missing or was built from something else than the working tree."""
    want = object_hashes()
    return [name for name, _, obj, _ in _units() if not os.path.exists(obj) or _recorded(obj) != want[name]]


def _source_id_text(want):
    rows = "".join('    {"%s", "%s"},\n' % (n, want[n]) for n in sorted(want))
    return ('// generated by artemis_amd/build.py -- the identity of the sources this library was built from\n'
            '#include <cstring>\n'
            'namespace { struct Row { const char *name, *sha; }; const Row rows[] = {\n' + rows + '}; }\n'
            'extern "C" const char *artemis_hip_source_sha(void) { return "%s"; }\n'
            'extern "C" const char *artemis_hip_object_sha(const char *name) {\n'
            '  if (!name) return nullptr;\n'
            '  for (const Row &r : rows) if (std::strcmp(r.name, name) == 0) return r.sha;\n'
            '  return nullptr;\n'
            '}\n' % tree_sha())


def build_hip(force=False, verbose=False, only=None):
    """Bring lib/libartemis_hip.so up to date with the working tree and return its path.  Objects whose content hash is
    unchanged are kept.  only = ["kernels_curv.hip", ...] (development): assert that nothing else needs compiling --
    the call REFUSES (RuntimeError) if an object outside the list is stale instead of linking it."""
    os.makedirs(LIBDIR, exist_ok=True)
    target = os.path.join(LIBDIR, "libartemis_hip.so")
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)  # (several ranks / test workers may call this at once)
        want = object_hashes()
        units = _units()
        todo = [u for u in units if force or not os.path.exists(u[2]) or _recorded(u[2]) != want[u[0]]]
        if only:
            listed = {o.replace(".hip", "").replace(".cpp", "") for o in only}
            outside = [u[0] for u in todo if u[0] not in listed and u[0].replace("driver_", "") not in listed]
            if outside and not force:
                raise RuntimeError("build_hip(only=%s): %s are stale too (their sources, headers or flags changed); "
                                   "refusing to link objects that are not the tree" % (only, outside))
        id_src = os.path.join(LIBDIR, "source_id.cpp")
        id_txt = _source_id_text(want)
        id_stale = (not os.path.exists(id_src)) or open(id_src).read() != id_txt or not os.path.exists(os.path.join(LIBDIR, "source_id.o"))
        if not todo and not id_stale and os.path.exists(target):
            return target
        procs = []
        for name, src, obj, cmd in todo:
            for p in (obj, obj + ".sha256"):
                if os.path.exists(p):
                    os.remove(p)
            full = cmd + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(full), flush=True)
            procs.append((name, obj, full, subprocess.Popen(full, env=_tool_env())))
        failed = []
        for name, obj, full, p in procs:
            if p.wait() != 0:
                failed.append(" ".join(full))
            else:
                with open(obj + ".sha256", "w") as f:
                    f.write(want[name] + "\n")
        if failed:
            raise RuntimeError("compile failed: " + "\n".join(failed))
        with open(id_src, "w") as f:
            f.write(id_txt)
        id_obj = os.path.join(LIBDIR, "source_id.o")
        subprocess.check_call(["g++", "-O1", "-fPIC", "-c", id_src, "-o", id_obj], env=_tool_env())
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-o", target + ".tmp"] + [u[2] for u in units] + [id_obj] + [
            "-L", os.path.join(ROCM, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, env=_tool_env())
        os.replace(target + ".tmp", target)
    return target


def verify(lib=None):
    """Raise unless the built library is the working tree: every object's recorded hash equals the tree's, and the
    identities compiled into the .so equal them too.  (`lib` = an already loaded ctypes handle, else the file is opened.)"""
    import ctypes
    stale = stale_objects()
    if stale:
        raise RuntimeError("objects not built from the working tree: %s (python -m artemis_amd.build)" % stale)
    L = lib or ctypes.CDLL(os.path.join(LIBDIR, "libartemis_hip.so"))
    L.artemis_hip_source_sha.restype = ctypes.c_char_p
    L.artemis_hip_object_sha.restype = ctypes.c_char_p
    L.artemis_hip_object_sha.argtypes = [ctypes.c_char_p]
    got = L.artemis_hip_source_sha().decode()
    if got != tree_sha():
        raise RuntimeError("libartemis_hip.so was built from other sources (%s) than the working tree (%s)" % (got, tree_sha()))
    for name, sha in object_hashes().items():
        have = L.artemis_hip_object_sha(name.encode())
        if have is None or have.decode() != sha:
            raise RuntimeError("libartemis_hip.so: object %s is not the working tree's" % name)
    return got


if __name__ == "__main__":
    only = [a for a in sys.argv[1:] if not a.startswith("-")] or None
    print(build_hip(force="-f" in sys.argv, verbose=True, only=only))
    print("source sha", verify())
