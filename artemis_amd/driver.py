"""Simulation: Python handle on the C++ host driver (include/artemis_driver.h).

Usage mirrors `artemis -i <deck> block/key=value ...` (tst/scripts/utils/artemis.py:122-156):

    sim = Simulation("inputs/blast/blast.in", ["parthenon/mesh/nx3=256", ...])
    sim.evolve()

One process per GPU: the mesh-block grid is split over the ranks and ghost slabs travel through an
artemis_comm_t -- RcclComm = the native C++ RCCL transport (GPU runs), TorchComm = callbacks into
torch.distributed ("gloo" in the CPU tests of the host logic).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import capi


class Msg(C.Structure):
    _fields_ = [("peer", C.c_int), ("tag", C.c_int), ("send", C.c_void_p), ("recv", C.c_void_p),
                ("count", C.c_long)]


EXCH_START = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(Msg), C.c_void_p)
EXCH_FINISH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
ALLRED_MIN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double))
ALLRED_SUM = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
ALLRED_MIN_DEV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)


class Comm(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int), ("nranks", C.c_int),
                ("exchange_start", EXCH_START), ("exchange_finish", EXCH_FINISH),
                ("allreduce_min", ALLRED_MIN), ("allreduce_min_dev", ALLRED_MIN_DEV),
                ("allreduce_sum", ALLRED_SUM)]


class _DevView:
    """Zero-copy torch view of a raw device (or host) buffer of doubles."""

    def __init__(self, ptr, count, cuda):
        self.ptr, self.count, self.cuda = ptr, count, cuda
        if cuda:
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8",
                                             "data": (ptr, False), "version": 2}


def _tensor_of(ptr, count, device, cache):
    key = (ptr, count)
    t = cache.get(key)
    if t is None:
        if device.type == "cuda":
            t = torch.as_tensor(_DevView(ptr, count, True), device=device)
        else:
            buf = (C.c_double * count).from_address(ptr)
            t = torch.frombuffer(buf, dtype=torch.float64)
        cache[key] = t
    return t


class TorchComm:
    """artemis_comm_t bound to torch.distributed (nccl/RCCL on GPUs, gloo on CPUs)."""

    def __init__(self, device, group=None):
        import torch.distributed as dist
        self.dist, self.group, self.device = dist, group, torch.device(device)
        self.rank, self.nranks = dist.get_rank(group), dist.get_world_size(group)
        self._cache, self._works, self._opcache = {}, [], {}
        self._cbs = (EXCH_START(self._start), EXCH_FINISH(self._finish),
                     ALLRED_MIN(self._min), ALLRED_MIN_DEV(self._min_dev), ALLRED_SUM(self._sum))
        self.struct = Comm(None, self.rank, self.nranks, *self._cbs)

    def _stream_ctx(self, stream_ptr):
        if self.device.type != "cuda":
            import contextlib
            return contextlib.nullcontext()
        return torch.cuda.stream(torch.cuda.ExternalStream(stream_ptr, device=self.device))

    def _start(self, ctx, nmsg, msgs, stream):
        try:
            dist = self.dist
            # the same message set comes back every stage: build the P2POp list once
            key = tuple((m.peer, m.tag, m.send, m.recv, m.count) for m in (msgs[i] for i in range(nmsg)))
            ops = self._opcache.get(key)
            if ops is None:
                ops = []
                # deterministic global order (RCCL matches sends and receives by posting order,
                # not by tag): sort by tag, receives and sends alike
                items = sorted((msgs[i] for i in range(nmsg)), key=lambda m: (m.tag, m.send is None))
                for m in items:
                    if m.send:
                        ops.append(dist.P2POp(dist.isend, _tensor_of(m.send, m.count, self.device, self._cache),
                                              m.peer, group=self.group, tag=m.tag))
                    if m.recv:
                        ops.append(dist.P2POp(dist.irecv, _tensor_of(m.recv, m.count, self.device, self._cache),
                                              m.peer, group=self.group, tag=m.tag))
                self._opcache[key] = ops
            with self._stream_ctx(stream):
                self._works = dist.batch_isend_irecv(ops) if ops else []
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("TorchComm.exchange_start failed:", repr(e), flush=True)
            return 1

    def _finish(self, ctx, stream):
        try:
            with self._stream_ctx(stream):
                for w in self._works:
                    w.wait()
            self._works = []
            return 0
        except Exception as e:
            print("TorchComm.exchange_finish failed:", repr(e), flush=True)
            return 1

    def _min(self, ctx, value):
        try:
            t = torch.tensor([value[0]], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
            value[0] = float(t.item())
            return 0
        except Exception as e:
            print("TorchComm.allreduce_min failed:", repr(e), flush=True)
            return 1

    def _min_dev(self, ctx, dev_value, stream):
        try:
            t = _tensor_of(dev_value, 1, self.device, self._cache)
            with self._stream_ctx(stream):
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
            return 0
        except Exception as e:
            print("TorchComm.allreduce_min_dev failed:", repr(e), flush=True)
            return 1

    def _sum(self, ctx, values, n):
        try:
            t = torch.tensor([values[i] for i in range(n)], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            for i, v in enumerate(t.tolist()):
                values[i] = v
            return 0
        except Exception as e:
            print("TorchComm.allreduce_sum failed:", repr(e), flush=True)
            return 1


class RcclComm:
    """The native C++ transport (artemis_comm_rccl_*, csrc/driver/comm_rccl.cpp): RCCL called directly from
    the driver's streams, no Python in the loop.  `share(obj) -> obj` broadcasts rank 0's unique id to every
    rank over any out-of-band channel (bench.py: a gloo broadcast); a single rank needs none."""

    def __init__(self, rank=0, nranks=1, share=None, lib=None):
        self.L = lib if lib is not None else capi.load()
        L = self.L
        L.artemis_comm_rccl_unique_id.argtypes = [C.c_char_p, C.c_int]
        L.artemis_comm_rccl_create.restype = C.c_void_p
        L.artemis_comm_rccl_create.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.artemis_comm_rccl_destroy.argtypes = [C.c_void_p]
        L.artemis_comm_rccl_count.argtypes = [C.c_void_p]
        L.artemis_comm_rccl_barrier.argtypes = [C.c_void_p]
        L.artemis_comm_rccl_last_error.restype = C.c_char_p
        nbytes = L.artemis_comm_rccl_unique_id_bytes()
        uid = None
        if rank == 0:
            buf = C.create_string_buffer(nbytes)
            if L.artemis_comm_rccl_unique_id(buf, nbytes):
                raise RuntimeError("ncclGetUniqueId: " + L.artemis_comm_rccl_last_error().decode())
            uid = buf.raw
        if nranks > 1:
            if share is None:
                raise ValueError("RcclComm with nranks > 1 needs a `share` callable to distribute the unique id")
            uid = share(uid)
        self.h = L.artemis_comm_rccl_create(uid, rank, nranks)
        if not self.h:
            raise RuntimeError("artemis_comm_rccl_create: " + L.artemis_comm_rccl_last_error().decode())
        self.struct = Comm.from_address(self.h)
        self.rank, self.nranks = rank, nranks

    @property
    def count(self):
        return self.L.artemis_comm_rccl_count(self.h)

    def barrier(self):
        if self.L.artemis_comm_rccl_barrier(self.h):
            raise RuntimeError("RCCL barrier: " + self.L.artemis_comm_rccl_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.artemis_comm_rccl_destroy(self.h)
            self.h = None


def _declare(L):
    if getattr(L, "_sim_declared", False):
        return
    vp, d, i, l = C.c_void_p, C.c_double, C.c_int, C.c_long
    L.artemis_sim_create.restype = vp
    L.artemis_sim_create.argtypes = [C.c_char_p, i, C.POINTER(C.c_char_p), C.POINTER(Comm)]
    L.artemis_sim_destroy.argtypes = [vp]
    L.artemis_sim_last_error.restype = C.c_char_p
    L.artemis_sim_evolve.restype = l
    L.artemis_sim_evolve.argtypes = [vp, l]
    for n in ("time", "dt", "tlim", "last_wall_seconds"):
        f = getattr(L, "artemis_sim_" + n)
        f.restype, f.argtypes = d, [vp]
    for n in ("ncycle", "local_zones", "total_zones"):
        f = getattr(L, "artemis_sim_" + n)
        f.restype, f.argtypes = l, [vp]
    L.artemis_sim_uses_fused_path.argtypes = [vp]
    L.artemis_sim_uses_tuned_kernel.argtypes = [vp]
    L.artemis_sim_remeshes.argtypes = [vp]
    L.artemis_sim_remeshes.restype = C.c_long
    L.artemis_sim_force_refine.argtypes = [vp, C.c_long]
    L.artemis_sim_force_refine.restype = C.c_int
    L.artemis_sim_inject_refine_tags.argtypes = [vp, C.POINTER(C.c_long), C.c_int]
    L.artemis_sim_inject_refine_tags.restype = C.c_int
    L.artemis_sim_load_balance.argtypes = [vp]
    L.artemis_sim_load_balance.restype = C.c_double
    L.artemis_sim_remesh_seconds.argtypes = [vp, C.POINTER(C.c_double)]
    L.artemis_sim_remesh_seconds.restype = C.c_long
    L.artemis_sim_last_remesh.argtypes = [vp, C.POINTER(C.c_long), C.POINTER(C.c_double)]
    L.artemis_sim_last_remesh.restype = None
    L.artemis_rt_device_bytes.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int]
    L.artemis_rt_device_bytes.restype = None
    L.artemis_rt_pool_bytes.restype = C.c_size_t
    L.artemis_rt_pool_bytes.argtypes = []
    L.artemis_sim_stage_kernel.argtypes = [vp]
    L.artemis_sim_stage_kernel.restype = C.c_char_p
    L.artemis_sim_set_path.argtypes = [vp, C.c_char_p]
    L.artemis_sim_set_overlap.argtypes = [vp, i]
    L.artemis_sim_overlap.argtypes = [vp]
    L.artemis_sim_set_dropin.argtypes = [vp, i]
    L.artemis_sim_species.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.artemis_sim_set_kernel_timing.argtypes = [vp, i]
    L.artemis_sim_nbody_force.argtypes = [vp, C.POINTER(d), i]
    L.artemis_sim_block_level.argtypes = [vp, i]
    L.artemis_sim_nblocks_global.restype = l
    L.artemis_sim_nblocks_global.argtypes = [vp]
    L.artemis_sim_dims.argtypes = [vp, C.POINTER(i)]
    L.artemis_sim_get_field.argtypes = [vp, C.c_char_p, i, vp]
    L.artemis_sim_block_bounds.argtypes = [vp, i, C.POINTER(d)]
    L.artemis_sim_history.argtypes = [vp, C.POINTER(d)]
    L.artemis_sim_errors.argtypes = [vp, C.POINTER(d)]
    L.artemis_sim_kernel_ms.restype = d
    L.artemis_sim_kernel_ms.argtypes = [vp, C.POINTER(l)]
    L._sim_declared = True


class Simulation:
    def __init__(self, deck, overrides=(), comm=None, lib=None):
        """deck: path to an input deck or its text; overrides: 'block/key=value' strings;
        comm: a TorchComm (or any object with a .struct artemis_comm_t) for multi-rank runs;
        lib: an alternative shared library exporting the same C ABI (tests only)."""
        self.L = lib if lib is not None else capi.load()
        _declare(self.L)
        text = open(deck).read() if os.path.exists(deck) else deck
        ov = (C.c_char_p * max(len(overrides), 1))(*[o.encode() for o in overrides])
        self.comm = comm
        cptr = C.byref(comm.struct) if comm is not None else None
        self.h = self.L.artemis_sim_create(text.encode(), len(overrides), ov, cptr)
        if not self.h:
            raise RuntimeError("artemis_sim_create: " + self.L.artemis_sim_last_error().decode())
        self._refresh_dims()
        ng_, nd_ = C.c_int(0), C.c_int(0)
        self.L.artemis_sim_species(self.h, C.byref(ng_), C.byref(nd_))
        self.ns_gas, self.ns_dust = ng_.value, nd_.value

    def _refresh_dims(self):
        """Block layout of this rank (an adaptive mesh changes it between cycles)."""
        d = (C.c_int * 11)()
        self.L.artemis_sim_dims(self.h, d)
        (self.nblocks, self.ni, self.nj, self.nk, self.is_, self.ie, self.js, self.je, self.ks,
         self.ke, self.ng) = list(d)

    def close(self):
        if getattr(self, "h", None):
            self.L.artemis_sim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def evolve(self, max_cycles=-1):
        n = self.L.artemis_sim_evolve(self.h, max_cycles)
        if n < 0:
            raise RuntimeError("artemis_sim_evolve: " + self.L.artemis_sim_last_error().decode())
        self._refresh_dims()
        return n

    time = property(lambda s: s.L.artemis_sim_time(s.h))
    dt = property(lambda s: s.L.artemis_sim_dt(s.h))
    tlim = property(lambda s: s.L.artemis_sim_tlim(s.h))
    ncycle = property(lambda s: s.L.artemis_sim_ncycle(s.h))
    local_zones = property(lambda s: s.L.artemis_sim_local_zones(s.h))
    total_zones = property(lambda s: s.L.artemis_sim_total_zones(s.h))
    uses_fused_path = property(lambda s: bool(s.L.artemis_sim_uses_fused_path(s.h)))
    uses_tuned_kernel = property(lambda s: bool(s.L.artemis_sim_uses_tuned_kernel(s.h)))
    stage_kernel = property(lambda s: s.L.artemis_sim_stage_kernel(s.h).decode())
    remeshes = property(lambda s: s.L.artemis_sim_remeshes(s.h))  # adaptive meshes: tree changes so far

    def force_refine(self, gid):
        """Split leaf `gid` (global Z-order index) through the ordinary remesh machinery; True if the mesh changed."""
        rc = self.L.artemis_sim_force_refine(self.h, gid)
        if rc < 0:
            raise RuntimeError(self.L.artemis_sim_last_error().decode())
        self._refresh_dims()
        return bool(rc)

    def inject_refine_tags(self, gids):
        """Tag the listed leaves +1 next to the criterion's own tags and run one remesh check; True if the mesh changed."""
        arr = (C.c_long * len(gids))(*gids)
        rc = self.L.artemis_sim_inject_refine_tags(self.h, arr, len(gids))
        if rc < 0:
            raise RuntimeError(self.L.artemis_sim_last_error().decode())
        self._refresh_dims()
        return bool(rc)

    def remesh_seconds(self):
        """(number of remeshes during the run, total seconds, build seconds, hand-over seconds, tagging seconds)"""
        out = (C.c_double * 4)()
        n = self.L.artemis_sim_remesh_seconds(self.h, out)
        return (n,) + tuple(out)

    def last_remesh(self):
        """The most recent remesh: (leaves before, after, created, destroyed), (seconds total, build, hand-over)"""
        lv, sec = (C.c_long * 4)(), (C.c_double * 3)()
        self.L.artemis_sim_last_remesh(self.h, lv, sec)
        return tuple(lv), tuple(sec)

    def device_bytes(self, reset_peak=False):
        """(current, peak) bytes of device memory held through the library's runtime shim"""
        cur, peak = C.c_size_t(0), C.c_size_t(0)
        self.L.artemis_rt_device_bytes(C.byref(cur), C.byref(peak), int(reset_peak))
        return cur.value, peak.value

    def cached_bytes(self):
        """bytes of device_bytes()[0] that sit free in the library's buffer cache (artemis_rt_pool_trim(0) returns them)"""
        return self.L.artemis_rt_pool_bytes()
    last_wall_seconds = property(lambda s: s.L.artemis_sim_last_wall_seconds(s.h))

    @property
    def load_balance(self):
        """max over ranks / mean of the cost of the Z-order rank split of a refined mesh (1 = even)."""
        return self.L.artemis_sim_load_balance(self.h)

    def set_path(self, which):
        if self.L.artemis_sim_set_path(self.h, which.encode()):
            raise RuntimeError(self.L.artemis_sim_last_error().decode())

    def set_overlap(self, mode):
        """False/0 off, True/2 one launch with in-kernel shell signalling, 1 shell + bulk launches."""
        mode = 2 if mode is True else int(mode)
        if self.L.artemis_sim_set_overlap(self.h, mode):
            raise RuntimeError(self.L.artemis_sim_last_error().decode())

    overlap = property(lambda s: s.L.artemis_sim_overlap(s.h))

    def set_dropin(self, on):
        """Tuned path only: also write cons on the last stage and run the whole-block PrimToCons per stage."""
        if self.L.artemis_sim_set_dropin(self.h, int(on)):  # 0 off, 1 whole-block PrimToCons, 2 ghost-zone PrimToCons
            raise RuntimeError(self.L.artemis_sim_last_error().decode())

    def set_kernel_timing(self, on):
        self.L.artemis_sim_set_kernel_timing(self.h, int(on))

    def kernel_ms(self):
        n = C.c_long(0)
        ms = self.L.artemis_sim_kernel_ms(self.h, C.byref(n))
        return ms, n.value

    def field(self, name, block=0):
        nvar = {"gas.prim": 6 * self.ns_gas, "gas.cons": 6 * self.ns_gas, "dust.prim": 4 * self.ns_dust,
                "dust.cons": 4 * self.ns_dust}[name]
        buf = np.empty((max(nvar, 1), self.nk, self.nj, self.ni))  # sized from the sim's species counts
        nv = self.L.artemis_sim_get_field(self.h, name.encode(), block, buf.ctypes.data)
        if nv < 0:
            raise RuntimeError(self.L.artemis_sim_last_error().decode())
        return buf.reshape(-1)[: nv * self.nk * self.nj * self.ni].reshape(nv, self.nk, self.nj, self.ni).copy()

    def interior(self, a):
        return a[..., self.ks:self.ke + 1, self.js:self.je + 1, self.is_:self.ie + 1]

    def nbody_force(self, reset=False):
        """[npart, 7] accumulated back-reaction rows of the n-body particles (summed over ranks)."""
        buf = (C.c_double * max(7 * self.L.artemis_sim_nbody_force(self.h, None, 0), 1))()  # sized by the size query
        n = self.L.artemis_sim_nbody_force(self.h, buf, int(reset))
        if n < 0:
            raise RuntimeError("artemis_sim_nbody_force failed")
        return np.array(buf[: 7 * n]).reshape(n, 7)

    def block_level(self, block=0):
        return self.L.artemis_sim_block_level(self.h, block)

    nblocks_global = property(lambda s: s.L.artemis_sim_nblocks_global(s.h))

    def block_bounds(self, block=0):
        o = (C.c_double * 6)()
        self.L.artemis_sim_block_bounds(self.h, block, o)
        return list(o)

    def history(self):
        o = (C.c_double * (6 + 4 * self.ns_dust))()
        n = self.L.artemis_sim_history(self.h, o)
        if n < 0:
            raise RuntimeError(self.L.artemis_sim_last_error().decode())
        return np.array(o[:n])

    def errors(self):
        o = (C.c_double * 16)()
        n = self.L.artemis_sim_errors(self.h, o)
        if n < 0:
            raise RuntimeError(self.L.artemis_sim_last_error().decode())
        return np.array(o[:n])
