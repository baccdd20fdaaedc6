"""ctypes loader for the CPU oracle (oracle/libpr_oracle.so).  Test infrastructure:
imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

from vcfdist_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None


class VpoExtra(C.Structure):
    _fields_ = [
        ("swap_writes", A.P_i64), ("swap_conflict_writes", A.P_i64),
        ("swap_used_conflict", A.P_i64), ("swap_used_conflict_nonmax", A.P_i64),
        ("path_len", A.P_i64),
        ("want_sc", C.c_int32), ("want_aln", C.c_int32),
        ("path_cap", C.c_int64), ("want_len", C.c_int64),
        ("path_plane", A.P_u8), ("path_qri", A.P_i32), ("path_ti", A.P_i32),
        ("path_sync", A.P_u8), ("path_edit", A.P_u8),
        ("dump_flags", A.P_u8 * 2), ("dump_pptr", A.P_u8 * 2), ("dump_pscore", A.P_i16 * 2),
    ]


def build():
    so = os.path.join(ORACLE_DIR, "libpr_oracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("pr_oracle.cpp", "cluster_oracle.cpp", "wfa_oracle.cpp", "summary_oracle.cpp", "pr_oracle.h")] + \
           [os.path.join(ROOT, "include", "vcfdist_pr.h"), os.path.join(ROOT, "include", "vcfdist_cluster.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.vpo_edit_distance.restype = C.c_int
        L.vpo_edit_distance.argtypes = [A.P_u8, C.c_int32, A.P_u8, C.c_int32]
        L.vpo_store_phase.restype = C.c_int32
        L.vpo_store_phase.argtypes = [C.POINTER(C.c_int32), C.c_double, A.P_i32, A.P_i32]
        L.vpo_run.restype = C.c_int
        L.vpo_run.argtypes = [C.POINTER(A.VprBatch), C.POINTER(A.VprConfig), C.POINTER(A.VprResults),
                              C.POINTER(VpoExtra)]
        L.vpo_gen_create.restype = C.c_void_p
        L.vpo_gen_create.argtypes = [C.POINTER(A.VprVariants)]
        L.vpo_gen_error.restype = C.c_int
        L.vpo_gen_error.argtypes = [C.c_void_p]
        L.vpo_gen_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.vpo_gen_copy.argtypes = [C.c_void_p, C.POINTER(A.P_i64), C.POINTER(A.P_u8), C.POINTER(A.P_i32),
                                   C.POINTER(A.P_u8), A.P_i64, A.P_u8, C.POINTER(A.P_i32), C.POINTER(A.P_u8)]
        L.vpo_gen_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def edit_distance(a: bytes, b: bytes) -> int:
    aa = np.frombuffer(a, dtype=np.uint8) if len(a) else np.zeros(1, np.uint8)
    bb = np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(1, np.uint8)
    return lib().vpo_edit_distance(aa.ctypes.data_as(A.P_u8), len(a), bb.ctypes.data_as(A.P_u8), len(b))


def store_phase(s, thr=0.6):
    arr = (C.c_int32 * 4)(*s)
    o, w = C.c_int32(), C.c_int32()
    ph = lib().vpo_store_phase(arr, thr, C.byref(o), C.byref(w))
    return ph, o.value, w.value


def generate(variants: A.Variants) -> A.Batch:
    """Oracle generate_ptrs_strs over every supercluster -> Level A Batch."""
    L = lib()
    vs = variants.as_struct()
    g = L.vpo_gen_create(C.byref(vs))
    try:
        err = L.vpo_gen_error(g)
        if err:
            raise ValueError(f"oracle generate failed: {err}")
        hl = (C.c_int64 * 4)()
        rl = C.c_int64()
        L.vpo_gen_sizes(g, hl, C.byref(rl))
        n = variants.n_sc
        hap_off = [np.zeros(n + 1, np.int64) for _ in range(4)]
        hap_seq = [np.zeros(hl[h], np.uint8) for h in range(4)]
        hap_ptr = [np.zeros(hl[h], np.int32) for h in range(4)]
        hap_flag = [np.zeros(hl[h], np.uint8) for h in range(4)]
        ref_off = np.zeros(n + 1, np.int64)
        ref_seq = np.zeros(rl.value, np.uint8)
        ref_ptr = [np.zeros(rl.value, np.int32) for _ in range(2)]
        ref_flag = [np.zeros(rl.value, np.uint8) for _ in range(2)]
        L.vpo_gen_copy(g,
                       (A.P_i64 * 4)(*[a.ctypes.data_as(A.P_i64) for a in hap_off]),
                       (A.P_u8 * 4)(*[a.ctypes.data_as(A.P_u8) for a in hap_seq]),
                       (A.P_i32 * 4)(*[a.ctypes.data_as(A.P_i32) for a in hap_ptr]),
                       (A.P_u8 * 4)(*[a.ctypes.data_as(A.P_u8) for a in hap_flag]),
                       ref_off.ctypes.data_as(A.P_i64), ref_seq.ctypes.data_as(A.P_u8),
                       (A.P_i32 * 2)(*[a.ctypes.data_as(A.P_i32) for a in ref_ptr]),
                       (A.P_u8 * 2)(*[a.ctypes.data_as(A.P_u8) for a in ref_flag]))
    finally:
        L.vpo_gen_free(g)
    var_pos = []
    for h in range(4):
        vp = variants.var_pos[h].copy()
        for sc in range(n):
            vp[variants.var_off[h][sc]:variants.var_off[h][sc + 1]] -= variants.sc_beg[sc]
        var_pos.append(vp)
    return A.Batch(n, hap_off, hap_seq, hap_ptr, hap_flag, ref_off, ref_seq, ref_ptr, ref_flag,
                   variants.var_off, var_pos, variants.var_qual)


class Extra:
    """Per-alignment diagnostics of an oracle run (+ optional dump of one alignment)."""

    def __init__(self, batch, want=None, dump_matrices=False):
        n = batch.n_sc * 4
        self.swap_writes = np.zeros(n, np.int64)
        self.swap_conflict_writes = np.zeros(n, np.int64)
        self.swap_used_conflict = np.zeros(n, np.int64)
        self.swap_used_conflict_nonmax = np.zeros(n, np.int64)
        self.path_len = np.zeros(n, np.int64)
        self.s = VpoExtra()
        for name in ("swap_writes", "swap_conflict_writes", "swap_used_conflict",
                     "swap_used_conflict_nonmax", "path_len"):
            setattr(self.s, name, getattr(self, name).ctypes.data_as(A.P_i64))
        self.s.want_sc, self.s.want_aln = (-1, -1) if want is None else want
        self.path = None
        if want is not None:
            sc, aln = want
            lq1, lq2, lt1, lt2, lr = batch.lens(sc)
            lq = (lq1, lq2)[aln >> 1]
            lt = (lt1, lt2)[aln & 1]
            cap = max(lq, lr) + lt + 4
            self.path_plane = np.zeros(cap, np.uint8)
            self.path_qri = np.zeros(cap, np.int32)
            self.path_ti = np.zeros(cap, np.int32)
            self.path_sync = np.zeros(cap + 1, np.uint8)
            self.path_edit = np.zeros(cap + 1, np.uint8)
            self.s.path_cap = cap
            self.s.path_plane = self.path_plane.ctypes.data_as(A.P_u8)
            self.s.path_qri = self.path_qri.ctypes.data_as(A.P_i32)
            self.s.path_ti = self.path_ti.ctypes.data_as(A.P_i32)
            self.s.path_sync = self.path_sync.ctypes.data_as(A.P_u8)
            self.s.path_edit = self.path_edit.ctypes.data_as(A.P_u8)
            if dump_matrices:
                self.flags = [np.zeros((lq, lt), np.uint8), np.zeros((lr, lt), np.uint8)]
                self.pptr = [np.zeros((lq, lt), np.uint8), np.zeros((lr, lt), np.uint8)]
                self.pscore = [np.zeros((lq, lt), np.int16), np.zeros((lr, lt), np.int16)]
                for p in range(2):
                    self.s.dump_flags[p] = self.flags[p].ctypes.data_as(A.P_u8)
                    self.s.dump_pptr[p] = self.pptr[p].ctypes.data_as(A.P_u8)
                    self.s.dump_pscore[p] = self.pscore[p].ctypes.data_as(A.P_i16)

    def path_arrays(self):
        n = int(self.s.want_len)
        return (self.path_plane[:n], self.path_qri[:n], self.path_ti[:n],
                self.path_sync[:n + 1], self.path_edit[:n + 1])


def run(batch: A.Batch, cfg=None, extra: Extra = None) -> A.Results:
    cfg = cfg or A.default_config()
    res = A.Results.for_batch(batch)
    bs, rs = batch.as_struct(), res.as_struct()
    rc = lib().vpo_run(C.byref(bs), C.byref(cfg), C.byref(rs), C.byref(extra.s) if extra else None)
    if rc:
        raise RuntimeError(f"oracle run failed: {rc}")
    return res


def oracle_pr_counts(OL, batch_var_off, res, var_class_per_slot, pb_phase=None, min_qual=0, max_qual=60):
    """the reference's nested counting loops (oracle/summary_oracle.cpp) on downloaded results"""
    nq = max_qual - min_qual + 1
    n_sc = len(res.sc_phase)
    out = np.zeros((2, 4, 3, nq), np.int64)
    voff = [np.ascontiguousarray(v, dtype=np.int64) for v in batch_var_off]
    cls = [np.ascontiguousarray(c, dtype=np.uint8) for c in var_class_per_slot]
    err = [[np.ascontiguousarray(res.errtype[s][w], dtype=np.uint8) for w in range(2)] for s in range(4)]
    cq = [[np.ascontiguousarray(res.callq[s][w], dtype=np.float32) for w in range(2)] for s in range(4)]
    P_f32 = C.POINTER(C.c_float)
    a_voff = (A.P_i64 * 4)(*[A._ptr(v, C.c_int64) for v in voff])
    a_err = ((A.P_u8 * 2) * 4)(*[(A.P_u8 * 2)(*[A._ptr(err[s][w], C.c_uint8) for w in range(2)]) for s in range(4)])
    a_cq = ((P_f32 * 2) * 4)(*[(P_f32 * 2)(*[A._ptr(cq[s][w], C.c_float) for w in range(2)]) for s in range(4)])
    a_cls = (A.P_u8 * 4)(*[A._ptr(c, C.c_uint8) for c in cls])
    scp = np.ascontiguousarray(res.sc_phase, dtype=np.int32)
    pb = None if pb_phase is None else np.ascontiguousarray(pb_phase, dtype=np.int32)
    OL.vso_pr_counts.argtypes = [C.c_int32, A.P_i64 * 4, (A.P_u8 * 2) * 4, (P_f32 * 2) * 4, A.P_u8 * 4, A.P_i32, A.P_i32,
                                 C.c_int32, C.c_int32, A.P_i64]
    rc = OL.vso_pr_counts(n_sc, a_voff, a_err, a_cq, a_cls, A._ptr(scp, C.c_int32), None if pb is None else A._ptr(pb, C.c_int32),
                          min_qual, max_qual, A._ptr(out, C.c_int64))
    assert rc == 0
    return out
