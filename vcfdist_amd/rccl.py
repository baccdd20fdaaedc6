"""The native collectives of the C ABI (vpr_allreduce_counts, vpr_allgather_phase: include/vcfdist_pr.h) from Python, and the
RCCL communicator they run on.  ctypes plumbing: the communicator is created with the process's own librccl (the one PyTorch
ships, when PyTorch is loaded), which is also the one the library resolves at run time."""
import ctypes as C
import os

import numpy as np

from . import _abi as A
from . import api
from . import summary

_RCCL = None
ID_BYTES = 128          # ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES)


def mapped_copies():
    """paths of the librccl copies mapped into this process (/proc/self/maps)"""
    out = []
    try:
        for line in open("/proc/self/maps"):
            i = line.find("/")
            if i < 0:
                continue
            path = line[i:].strip()
            base = path.rsplit("/", 1)[-1]
            if base.startswith("librccl") and ".so" in base and path not in out:
                out.append(path)
    except OSError:
        pass
    return out


def lib():
    """ONE RCCL per process: the copy that is already mapped (PyTorch's own librccl.so once torch is imported) is bound by its
    path -- which is also what the C side does (vpr_rccl_library) --, a soname is only opened when there is none.  Two copies, a
    communicator made by one and used through the other, would be a crash."""
    global _RCCL
    if _RCCL is None:
        err = None
        L = api.lib()
        L.vpr_rccl_available.restype = C.c_int
        L.vpr_rccl_library.restype = C.c_char_p
        have = mapped_copies()
        names = have[:1] if have else []
        if not names and L.vpr_rccl_available():      # (the library opened one: the same)
            p = L.vpr_rccl_library().decode()
            names = [p] if p else []
        cand = None
        for name in names + ["librccl.so.1", "librccl.so"]:
            try:
                cand = C.CDLL(name, mode=C.RTLD_GLOBAL)
                break
            except OSError as e:
                err = e
        if cand is None:
            raise OSError(f"no RCCL library: {err}")
        cand.ncclGetErrorString.restype = C.c_char_p
        # (the checks run before the handle is cached: a failed check must fail every later call as well -- ADVICE r5)
        if len(mapped_copies()) > 1:
            raise OSError(f"more than one RCCL mapped into the process: {mapped_copies()}")
        lp = L.vpr_rccl_library().decode() if L.vpr_rccl_available() else ""
        mc = mapped_copies()
        if lp and mc and os.path.realpath(lp) != os.path.realpath(mc[0]):
            raise OSError(f"the library bound {lp}, the process maps {mc[0]}")
        _RCCL = cand
    return _RCCL


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * ID_BYTES)]


def unique_id() -> bytes:
    uid = UniqueId()
    _chk(lib().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return bytes(C.string_at(C.byref(uid), ID_BYTES))


def _chk(rc, what):
    if rc:
        raise RuntimeError(f"{what} failed: {lib().ncclGetErrorString(rc).decode()}")


class Comm:
    """ncclComm_t of `world` ranks; `uid` = unique_id() of rank 0, handed to the other ranks by the caller (any channel: the
    bench broadcasts it over torch.distributed).  The calling thread's HIP device must be the rank's GPU."""

    def __init__(self, world: int, rank: int, uid: bytes):
        self.world, self.rank = world, rank
        u = UniqueId()
        C.memmove(C.byref(u), uid, ID_BYTES)
        self._c = C.c_void_p()
        L = lib()
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        _chk(L.ncclCommInitRank(C.byref(self._c), world, u, rank), "ncclCommInitRank")

    def count(self) -> int:
        """ranks of the communicator as RCCL itself reports them (ncclCommCount)"""
        n = C.c_int(0)
        L = lib()
        L.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _chk(L.ncclCommCount(self._c, C.byref(n)), "ncclCommCount")
        return int(n.value)

    def destroy(self):
        if self._c:
            lib().ncclCommDestroy.argtypes = [C.c_void_p]
            lib().ncclCommDestroy(self._c)
            self._c = C.c_void_p()


def available() -> bool:
    L = api.lib()
    L.vpr_rccl_available.restype = C.c_int
    try:
        lib()
    except OSError:
        return False
    return bool(L.vpr_rccl_available())


def library_paths():
    """(path bound by the C side, copies mapped into the process): tests assert there is exactly one copy and that it is that one"""
    L = api.lib()
    L.vpr_rccl_library.restype = C.c_char_p
    L.vpr_rccl_available.restype = C.c_int
    return (L.vpr_rccl_library().decode() if L.vpr_rccl_available() else None), mapped_copies()


def allreduce_counts(pr, comm: Comm, var_class_per_slot=None, pb_phase=None, min_qual=0, max_qual=60):
    """summary.pr_counts summed over the ranks of `comm`: one all-reduce of the device histogram -> int64 [2][4][3][nq]"""
    L = api.lib()
    nq = max_qual - min_qual + 1
    out = np.zeros((2, summary.VARTYPES, 3, nq), np.int64)
    pb = None if pb_phase is None else np.ascontiguousarray(pb_phase, dtype=np.int32)
    L.vpr_allreduce_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, A.P_i32, C.c_int32, C.c_int32, A.P_i64]
    arr = None
    if var_class_per_slot is not None:
        cls = [np.ascontiguousarray(c, dtype=np.uint8) for c in var_class_per_slot]
        arr = (A.P_u8 * 4)(*[A._ptr(c, C.c_uint8) for c in cls])
    rc = L.vpr_allreduce_counts(pr._h, comm._c, arr, None if pb is None else A._ptr(pb, C.c_int32), min_qual, max_qual,
                                A._ptr(out, C.c_int64))
    if rc:
        raise RuntimeError(f"vpr_allreduce_counts failed: {rc} {L.vpr_last_error(pr._h)}")
    return out


def allgather_phase(pr, comm: Comm, idx_local, n_total: int):
    """(sc_phase, orig_phase_dist, swap_phase_dist) of all n_total superclusters on every rank; idx_local[k] = global index of
    this rank's k-th supercluster"""
    L = api.lib()
    idx = np.ascontiguousarray(idx_local, dtype=np.int32)
    out = [np.zeros(n_total, np.int32) for _ in range(3)]
    L.vpr_allgather_phase.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, A.P_i32, C.c_int32, A.P_i32, A.P_i32, A.P_i32]
    rc = L.vpr_allgather_phase(pr._h, comm._c, comm.world, A._ptr(idx, C.c_int32), n_total, A._ptr(out[0], C.c_int32),
                               A._ptr(out[1], C.c_int32), A._ptr(out[2], C.c_int32))
    if rc:
        raise RuntimeError(f"vpr_allgather_phase failed: {rc} {L.vpr_last_error(pr._h)}")
    return out
