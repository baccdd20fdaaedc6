"""ctypes mirror of the phasing / precision-recall summary entry points of include/vcfdist_pr.h (SURVEY 8(e), 8(f)
rank 4) and of their oracle counterparts (prefix vso_)."""
import ctypes as C

import numpy as np

from . import _abi as A
from . import api

VARTYPES = 4
NAMES = ["SNP", "INDEL", "SV", "ALL"]


class PrRow(C.Structure):
    _fields_ = [("vartype", C.c_int32), ("best", C.c_int32), ("qual", C.c_int32),
                ("truth_tp", C.c_int32), ("query_tp", C.c_int32), ("truth_fn", C.c_int32), ("query_fp", C.c_int32),
                ("precision", C.c_float), ("recall", C.c_float), ("f1_score", C.c_float), ("f1_qscore", C.c_float)]

    def key(self):
        f = lambda x: np.float32(x).view(np.uint32).item()
        return (self.vartype, self.best, self.qual, self.truth_tp, self.query_tp, self.truth_fn, self.query_fp,
                f(self.precision), f(self.recall), f(self.f1_score), f(self.f1_qscore))


def var_class(var_type, ref_len, alt_len, sv_threshold=50):
    """SNP / INDEL / SV class of every variant (print.cpp:362-372)"""
    t = np.asarray(var_type); r = np.asarray(ref_len); a = np.asarray(alt_len)
    small = ((t == 2) & (a < sv_threshold)) | ((t == 3) & (r < sv_threshold))
    return np.where(t == 1, 0, np.where(small, 1, 2)).astype(np.uint8)


def phase(sc_phase, phase_set, L=None, prefix="vpr"):
    L = L or api.lib()
    sc_phase = np.ascontiguousarray(sc_phase, dtype=np.int32)
    phase_set = np.ascontiguousarray(phase_set, dtype=np.int32)
    n = len(sc_phase)
    pb = np.zeros(n, np.int32); sw = np.zeros(max(n, 1), np.int32); fl = np.zeros(max(n, 1), np.int32)
    ns = C.c_int32(); nf = C.c_int32()
    f = getattr(L, prefix + "_phase")
    f.argtypes = [A.P_i32, A.P_i32, C.c_int32, A.P_i32, A.P_i32, C.POINTER(C.c_int32), A.P_i32, C.POINTER(C.c_int32)]
    rc = f(A._ptr(sc_phase, C.c_int32), A._ptr(phase_set, C.c_int32), n, A._ptr(pb, C.c_int32), A._ptr(sw, C.c_int32),
           C.byref(ns), A._ptr(fl, C.c_int32), C.byref(nf))
    if rc:
        raise ValueError(f"{prefix}_phase failed: {rc}")
    return pb, sw[:ns.value].copy(), fl[:nf.value].copy()


def pr_summary(counts, min_qual=0, max_qual=60, L=None, prefix="vpr"):
    L = L or api.lib()
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    rows = (PrRow * (2 * VARTYPES))()
    f = getattr(L, prefix + "_pr_summary")
    f.argtypes = [A.P_i64, C.c_int32, C.c_int32, C.POINTER(PrRow)]
    rc = f(A._ptr(counts, C.c_int64), min_qual, max_qual, rows)
    if rc:
        raise ValueError(f"{prefix}_pr_summary failed: {rc}")
    return list(rows)


def pr_counts(pr, var_class_per_slot, pb_phase=None, min_qual=0, max_qual=60):
    """device counters of the last execute of a PrecisionRecall handle -> int64 [2][4][3][nq]"""
    L = api.lib()
    nq = max_qual - min_qual + 1
    out = np.zeros((2, VARTYPES, 3, nq), np.int64)
    pb = None if pb_phase is None else np.ascontiguousarray(pb_phase, dtype=np.int32)
    L.vpr_pr_counts.argtypes = [C.c_void_p, C.c_void_p, A.P_i32, C.c_int32, C.c_int32, A.P_i64]
    arr = None
    if var_class_per_slot is not None:      # None: classes already resident (upload_var_class)
        cls = [np.ascontiguousarray(c, dtype=np.uint8) for c in var_class_per_slot]
        arr = (A.P_u8 * 4)(*[A._ptr(c, C.c_uint8) for c in cls])
    rc = L.vpr_pr_counts(pr._h, arr, None if pb is None else A._ptr(pb, C.c_int32), min_qual, max_qual, A._ptr(out, C.c_int64))
    if rc:
        raise ValueError(f"vpr_pr_counts failed: {rc} {L.vpr_last_error(pr._h)}")
    return out


def upload_var_class(pr, var_class_per_slot):
    L = api.lib()
    cls = [np.ascontiguousarray(c, dtype=np.uint8) for c in var_class_per_slot]
    arr = (A.P_u8 * 4)(*[A._ptr(c, C.c_uint8) for c in cls])
    L.vpr_upload_var_class.argtypes = [C.c_void_p, A.P_u8 * 4]
    rc = L.vpr_upload_var_class(pr._h, arr)
    if rc:
        raise ValueError(f"vpr_upload_var_class failed: {rc}")
