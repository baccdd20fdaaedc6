"""Python binding of the C ABI (include/vcfdist_pr.h) -- ctypes over
vcfdist_amd/lib/libvcfdist_pr.so.  Plumbing only: every result comes from the
HIP kernels behind the ABI; there is no Python or CPU fallback, and a missing
library or GPU raises."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _abi as A

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libvcfdist_pr.so")
CSRC = os.path.join(HERE, "csrc")
_LIB = None


class VprError(RuntimeError):
    pass


def build(force=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    inc = os.path.join(os.path.dirname(HERE), "include")
    srcs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    cli = os.path.join(os.path.dirname(LIB_PATH), "vcfdist_gpu")       # (the C++ command line, csrc/main.cpp: built beside the library)
    newest = min(os.path.getmtime(p) for p in (LIB_PATH, cli)) if os.path.exists(LIB_PATH) and os.path.exists(cli) else None
    stale = newest is None or any(os.path.getmtime(s) > newest and not (s.endswith("main.cpp") and os.path.getmtime(s) <= os.path.getmtime(cli))
                                  for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s"] + (["-B"] if force else []))
    return LIB_PATH


# The path keeps up to eight HIP streams busy at once (two parts of round 0, two retry ladders, two tie ladders and
# their side streams); the runtime's default of four hardware queues would serialise them pairwise.  Read by the HIP
# runtime when it initialises, so it has to be in the environment before the first HIP call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise VprError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no fallback path)")
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    L.vpr_version.restype = C.c_char_p
    L.vpr_last_error.restype = C.c_char_p
    L.vpr_last_error.argtypes = [H]
    L.vpr_create.argtypes = [C.POINTER(A.VprConfig), C.POINTER(H)]
    L.vpr_destroy.argtypes = [H]
    L.vpr_run.argtypes = [H, C.POINTER(A.VprBatch), C.POINTER(A.VprResults)]
    L.vpr_upload.argtypes = [H, C.POINTER(A.VprBatch)]
    L.vpr_upload_variants.argtypes = [H, C.POINTER(A.VprVariants)]
    L.vpr_execute.argtypes = [H]
    L.vpr_download.argtypes = [H, C.POINTER(A.VprResults)]
    L.vpr_host_alloc.restype = C.c_void_p
    L.vpr_host_alloc.argtypes = [C.c_size_t]
    L.vpr_host_free.argtypes = [C.c_void_p]
    L.vpr_results_alloc.argtypes = [H, C.POINTER(A.VprResults), C.POINTER(C.c_void_p)]
    L.vpr_get_timing.argtypes = [H, C.POINTER(A.VprTiming)]
    L.vpr_get_launch_stats.argtypes = [H, C.POINTER(A.VprLaunchStat), C.c_int32]
    L.vpr_get_tally.argtypes = [H, C.POINTER(C.c_int64)]
    L.vpr_download_path.restype = C.c_int64
    L.vpr_download_path.argtypes = [H, C.c_int32, C.c_int32, C.c_int64, A.P_u8, A.P_i32, A.P_i32, A.P_u8, A.P_u8]
    L.vpr_store_phase.restype = C.c_int32
    L.vpr_store_phase.argtypes = [C.POINTER(C.c_int32), C.c_double, A.P_i32, A.P_i32]
    L.vpr_batch_from_variants.argtypes = [C.POINTER(A.VprVariants), C.POINTER(H)]
    L.vpr_owned_batch_view.restype = C.POINTER(A.VprBatch)
    L.vpr_owned_batch_view.argtypes = [H]
    L.vpr_owned_batch_free.argtypes = [H]
    L.vpr_synth_default_params.argtypes = [C.POINTER(A.VprSynthParams)]
    L.vpr_synth_create.argtypes = [C.POINTER(A.VprSynthParams), C.POINTER(H)]
    L.vpr_synth_variants.restype = C.POINTER(A.VprVariants)
    L.vpr_synth_variants.argtypes = [H]
    L.vpr_synth_destroy.argtypes = [H]
    _LIB = L
    return L


EXPORTED = [
    "vpr_create", "vpr_destroy", "vpr_last_error", "vpr_version", "vpr_run", "vpr_upload",
    "vpr_upload_variants", "vpr_download_level_a", "vpr_select_device", "vpr_execute", "vpr_download", "vpr_host_alloc", "vpr_host_free", "vpr_get_timing", "vpr_get_launch_stats",
    "vpr_get_tally",
    "vpr_download_path", "vpr_phase", "vpr_var_class", "vpr_upload_var_class", "vpr_results_alloc", "vpr_pr_counts", "vpr_pr_summary",
    "vpr_store_phase", "vpr_batch_from_variants", "vpr_owned_batch_view", "vpr_owned_batch_free",
    "vpr_synth_default_params", "vpr_synth_create", "vpr_synth_variants", "vpr_synth_destroy",
]


def store_phase(s, thr=0.6):
    arr = (C.c_int32 * 4)(*[int(x) for x in s])
    o, w = C.c_int32(), C.c_int32()
    ph = lib().vpr_store_phase(arr, thr, C.byref(o), C.byref(w))
    return ph, o.value, w.value


def batch_from_variants(variants: A.Variants) -> A.Batch:
    """Host marshalling (generate_ptrs_strs x4 per supercluster) -> Level A Batch."""
    L = lib()
    vs = variants.as_struct()
    ob = C.c_void_p()
    rc = L.vpr_batch_from_variants(C.byref(vs), C.byref(ob))
    if rc:
        raise VprError(f"vpr_batch_from_variants failed: {rc}" + (
            " (input generate_ptrs_strs cannot process: unsorted / overlapping variants on a haplotype, a variant type other than "
            "SUB/INS/DEL, or a supercluster region that leaves its contig -- a variant ending on the last base of a contig; see "
            "include/vcfdist_pr.h)" if rc == -1 else ""))
    try:
        return A.Batch.from_struct(L.vpr_owned_batch_view(ob).contents)
    finally:
        L.vpr_owned_batch_free(ob)


def synth_params(**kw) -> A.VprSynthParams:
    p = A.VprSynthParams()
    lib().vpr_synth_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


class _Owned:
    """Keeps a vpr_owned_batch alive for numpy views into it."""

    def __init__(self, h):
        self._h = h

    def __del__(self):
        try:
            if self._h:
                lib().vpr_owned_batch_free(self._h)
                self._h = None
        except Exception:
            pass


class Synth:
    """Owns a generated synthetic workload (variants + reference) inside the library."""

    def __init__(self, **kw):
        self.params = synth_params(**kw)
        self._h = C.c_void_p()
        rc = lib().vpr_synth_create(C.byref(self.params), C.byref(self._h))
        if rc:
            raise VprError(f"vpr_synth_create failed: {rc}")

    @property
    def struct(self):
        return lib().vpr_synth_variants(self._h).contents

    def variants(self) -> A.Variants:
        return A.Variants.from_struct(self.struct)

    def var_class(self, sv_threshold=50):
        """SNP / INDEL / SV class of every variant of the four hap slots (print.cpp:362-372), from views into the
        generator's tables (variants() copies the allele pools and the contigs as well)"""
        s = self.struct
        L = lib()
        L.vpr_var_class.restype = None
        L.vpr_var_class.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        out = []
        for h in range(A.HAPS):
            nv = int(s.var_off[h][s.n_sc])
            o = np.zeros(nv, np.uint8)
            if nv:
                L.vpr_var_class(C.cast(s.var_type[h], C.c_void_p), C.cast(s.var_ref_len[h], C.c_void_p),
                                C.cast(s.var_alt_len[h], C.c_void_p), nv, sv_threshold, o.ctypes.data)
            out.append(o)
        return out

    def batch(self, copy=True) -> A.Batch:
        """Level A batch of the workload.  copy=False returns numpy views into the library-owned
        buffers (no second copy of a multi-GB batch); the Batch keeps them alive."""
        L = lib()
        ob = C.c_void_p()
        rc = L.vpr_batch_from_variants(L.vpr_synth_variants(self._h), C.byref(ob))
        if rc:
            raise VprError(f"vpr_batch_from_variants failed: {rc}" + (
            " (input generate_ptrs_strs cannot process: unsorted / overlapping variants on a haplotype, a variant type other than "
            "SUB/INS/DEL, or a supercluster region that leaves its contig -- a variant ending on the last base of a contig; see "
            "include/vcfdist_pr.h)" if rc == -1 else ""))
        if not copy:
            return A.Batch.from_struct(L.vpr_owned_batch_view(ob).contents, copy=False, owner=_Owned(ob))
        try:
            return A.Batch.from_struct(L.vpr_owned_batch_view(ob).contents)
        finally:
            L.vpr_owned_batch_free(ob)

    def close(self):
        if self._h:
            lib().vpr_synth_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PrecisionRecall:
    """Mirror of the reference's precision_recall_wrapper for a batch of superclusters."""

    def __init__(self, cfg: A.VprConfig = None, device=0, **kw):
        self.cfg = cfg or A.default_config(device=device, **kw)
        self._h = C.c_void_p()
        L = lib()
        rc = L.vpr_create(C.byref(self.cfg), C.byref(self._h))
        if rc:
            raise VprError(f"vpr_create failed ({rc}): {L.vpr_last_error(None).decode()}")
        self._batch = None

    def _chk(self, rc, what):
        if rc:
            raise VprError(f"{what} failed ({rc}): {lib().vpr_last_error(self._h).decode()}")

    def upload(self, batch: A.Batch):
        self._batch = batch
        s = batch.as_struct()
        self._chk(lib().vpr_upload(self._h, C.byref(s)), "vpr_upload")

    def execute(self):
        self._chk(lib().vpr_execute(self._h), "vpr_execute")

    def download(self, res: A.Results = None) -> A.Results:
        """Copy the results of the last execute to host memory.  Pass a previous Results to reuse its
        buffers (every field is overwritten), which avoids re-allocating hundreds of MB per call."""
        if res is None:       # page-locked buffers: the copies then run at the link rate
            L = lib()
            s = A.VprResults()
            blk = C.c_void_p()
            # one block laid out like the device's result columns: a single copy per download
            if L.vpr_results_alloc(self._h, C.byref(s), C.byref(blk)) == 0 and blk.value:
                res = A.Results.mirror(s, self._batch.n_sc, [self._batch.n_vars(h) for h in range(A.HAPS)], blk.value, L.vpr_host_free)
            else:
                res = A.Results.for_batch(self._batch, L.vpr_host_alloc, L.vpr_host_free)
        s = res.as_struct()
        self._chk(lib().vpr_download(self._h, C.byref(s)), "vpr_download")
        return res

    def run(self, batch: A.Batch) -> A.Results:
        self.upload(batch)
        self.execute()
        return self.download()

    def timing(self) -> A.VprTiming:
        t = A.VprTiming()
        self._chk(lib().vpr_get_timing(self._h, C.byref(t)), "vpr_get_timing")
        return t

    def tally(self):
        """int64[2 callsets][TP,FP,FN] accumulated on the device by the last execute."""
        out = (C.c_int64 * 6)()
        self._chk(lib().vpr_get_tally(self._h, out), "vpr_get_tally")
        return np.array(list(out), dtype=np.int64).reshape(2, 3)

    def launch_stats(self):
        L = lib()
        n = L.vpr_get_launch_stats(self._h, None, 0)
        arr = (A.VprLaunchStat * max(n, 1))()
        L.vpr_get_launch_stats(self._h, arr, n)
        return [arr[i] for i in range(n)]

    def upload_variants(self, variants_struct, batch_for_results):
        """Upload a vpr_variants struct (e.g. Synth.struct): the host sizes and checks the regions, the device writes the
        haplotype strings and pointer arrays (generate_ptrs_strs, pr_gen.hip).  `batch_for_results` only sizes the result
        buffers: anything with n_sc and n_vars(h) (a Batch, a Variants)."""
        self._batch = batch_for_results
        self._chk(lib().vpr_upload_variants(self._h, C.byref(variants_struct)), "vpr_upload_variants")

    def download_level_a(self, like: A.Batch) -> A.Batch:
        """test aid: the resident Level A arrays (written by the host marshalling or by the device generator) as a Batch
        shaped like `like` (same offsets)"""
        z = lambda a: np.zeros_like(a)
        out = A.Batch(like.n_sc, [z(a) for a in like.hap_off], [z(a) for a in like.hap_seq], [z(a) for a in like.hap_ptr],
                      [z(a) for a in like.hap_flag], z(like.ref_off), z(like.ref_seq), [z(a) for a in like.ref_ptr],
                      [z(a) for a in like.ref_flag], like.var_off, like.var_pos, like.var_qual)
        s = out.as_struct()
        L = lib()
        L.vpr_download_level_a.argtypes = [C.c_void_p, C.POINTER(A.VprBatch)]
        self._chk(L.vpr_download_level_a(self._h, C.byref(s)), "vpr_download_level_a")
        return out

    def path(self, sc, aln):
        """(plane, qri, ti, sync, edit) arrays of one alignment's walk (last workspace chunk only)."""
        lq1, lq2, lt1, lt2, lr = self._batch.lens(sc)
        cap = lq1 + lq2 + lr + lt1 + lt2 + 8
        pl = np.zeros(cap, np.uint8); q = np.zeros(cap, np.int32); t = np.zeros(cap, np.int32)
        sy = np.zeros(cap, np.uint8); ed = np.zeros(cap, np.uint8)
        n = lib().vpr_download_path(self._h, sc, aln, cap, pl.ctypes.data_as(A.P_u8), q.ctypes.data_as(A.P_i32),
                                    t.ctypes.data_as(A.P_i32), sy.ctypes.data_as(A.P_u8), ed.ctypes.data_as(A.P_u8))
        if n < 0:
            raise VprError(f"vpr_download_path failed: {n}")
        return pl[:n], q[:n], t[:n], sy[:n], ed[:n]

    def close(self):
        if self._h:
            lib().vpr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
