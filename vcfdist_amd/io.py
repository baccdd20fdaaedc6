"""ctypes mirror of include/vcfdist_io.h (VCF / BED / FASTA readers)."""
import ctypes as C

import numpy as np

from . import _abi as A
from . import api

P_f32 = C.POINTER(C.c_float)


class VioParams(C.Structure):
    _fields_ = [("min_qual", C.c_int32), ("max_qual", C.c_int32), ("max_size", C.c_int32), ("cluster_min_gap", C.c_int32)]


class VioHapVars(C.Structure):
    _fields_ = [("n", C.c_int32), ("pos", A.P_i32), ("rlen", A.P_i32), ("type", A.P_u8), ("orig_gt", A.P_u8),
                ("var_qual", P_f32), ("gt_qual", P_f32), ("phase_set", A.P_i32), ("ref_len", A.P_i32), ("alt_len", A.P_i32),
                ("ref_off", A.P_i64), ("alt_off", A.P_i64), ("pool", A.P_u8), ("pool_len", C.c_int64)]


COUNTERS = ["n_records", "n_failed_filter", "n_low_qual", "n_unphased", "n_overlap", "n_bed_outside", "n_bed_border",
            "n_bed_offctg", "n_large", "n_complex", "n_ref_call", "n_spanning_del", "n_unknown_allele", "n_wrong_ploidy",
            "n_ps_missing"]


class VioCallset(C.Structure):
    _fields_ = [("n_ctg", C.c_int32), ("ctg_name", C.POINTER(C.c_char_p)), ("ctg_len", A.P_i64), ("ploidy", A.P_i32),
                ("vars", C.POINTER(VioHapVars)), ("sample", C.c_char_p)] + [(k, C.c_int64) for k in COUNTERS]


class VioFasta(C.Structure):
    _fields_ = [("n_ctg", C.c_int32), ("ctg_name", C.POINTER(C.c_char_p)), ("ctg_off", A.P_i64), ("seq", A.P_u8)]


EXPORTED = ["vio_read_bed", "vio_bed_free", "vio_bed_contains", "vio_read_vcf", "vio_callset_free", "vio_read_fasta",
            "vio_fasta_free", "vio_last_error"]


def _err():
    L = api.lib()
    L.vio_last_error.restype = C.c_char_p
    return L.vio_last_error().decode()


class Bed:
    def __init__(self, path):
        L = api.lib()
        self._h = C.c_void_p()
        L.vio_read_bed.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        rc = L.vio_read_bed(path.encode(), C.byref(self._h))
        if rc:
            raise IOError(f"vio_read_bed failed ({rc}): {_err()}")
        # contigs in the order the file first names them (bedData::contigs, bed.cpp:22-26)
        import gzip
        self.contigs = []
        with (gzip.open(path, "rt") if path.endswith(".gz") else open(path)) as fh:
            for line in fh:
                c = line.split("\t", 1)[0].strip()
                if c and c not in self.contigs:
                    self.contigs.append(c)

    def contains(self, ctg, start, stop, typ):
        L = api.lib()
        L.vio_bed_contains.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32]
        return L.vio_bed_contains(self._h, ctg.encode(), start, stop, typ)

    def __del__(self):
        try:
            L = api.lib()
            L.vio_bed_free.argtypes = [C.c_void_p]
            if self._h:
                L.vio_bed_free(self._h)
                self._h = None
        except Exception:
            pass


def read_vcf(path, bed=None, min_qual=0, max_qual=60, max_size=5000, cluster_min_gap=50, filters=()):
    """-> dict(contigs=[...], lengths=[...], ploidy=[...], sample, stats={...}, vars[ctg][hap] = dict of numpy columns)"""
    L = api.lib()
    prm = VioParams(min_qual, max_qual, max_size, cluster_min_gap)
    out = C.POINTER(VioCallset)()
    flt = (C.c_char_p * max(len(filters), 1))(*[f.encode() for f in filters])
    L.vio_read_vcf.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(VioParams), C.POINTER(C.c_char_p), C.c_int32,
                               C.POINTER(C.POINTER(VioCallset))]
    rc = L.vio_read_vcf(path.encode(), bed._h if bed is not None else None, C.byref(prm), flt, len(filters), C.byref(out))
    if rc:
        raise IOError(f"vio_read_vcf failed ({rc}): {_err()}")
    c = out.contents
    res = dict(contigs=[c.ctg_name[k].decode() for k in range(c.n_ctg)], lengths=[int(c.ctg_len[k]) for k in range(c.n_ctg)],
               ploidy=[int(c.ploidy[k]) for k in range(c.n_ctg)], sample=c.sample.decode(),
               stats={k: int(getattr(c, k)) for k in COUNTERS}, vars=[])
    for k in range(c.n_ctg):
        per = []
        for hp in range(2):
            V = c.vars[k * 2 + hp]
            n = V.n
            per.append(dict(pos=A._from_ptr(V.pos, n, np.int32).copy(), rlen=A._from_ptr(V.rlen, n, np.int32).copy(),
                            type=A._from_ptr(V.type, n, np.uint8).copy(), orig_gt=A._from_ptr(V.orig_gt, n, np.uint8).copy(),
                            var_qual=A._from_ptr(V.var_qual, n, np.float32).copy(), gt_qual=A._from_ptr(V.gt_qual, n, np.float32).copy(),
                            phase_set=A._from_ptr(V.phase_set, n, np.int32).copy(), ref_len=A._from_ptr(V.ref_len, n, np.int32).copy(),
                            alt_len=A._from_ptr(V.alt_len, n, np.int32).copy(), ref_off=A._from_ptr(V.ref_off, n, np.int64).copy(),
                            alt_off=A._from_ptr(V.alt_off, n, np.int64).copy(),
                            pool=A._from_ptr(V.pool, int(V.pool_len), np.uint8).copy() if V.pool_len else np.zeros(1, np.uint8)))
        res["vars"].append(per)
    L.vio_callset_free.argtypes = [C.POINTER(VioCallset)]
    L.vio_callset_free(out)
    return res


def read_fasta(path):
    """-> {name: numpy uint8 (upper-case)}"""
    L = api.lib()
    out = C.POINTER(VioFasta)()
    L.vio_read_fasta.argtypes = [C.c_char_p, C.POINTER(C.POINTER(VioFasta))]
    rc = L.vio_read_fasta(path.encode(), C.byref(out))
    if rc:
        raise IOError(f"vio_read_fasta failed ({rc}): {_err()}")
    f = out.contents
    offs = A._from_ptr(f.ctg_off, f.n_ctg + 1, np.int64)
    seq = A._from_ptr(f.seq, int(offs[-1]), np.uint8)
    res = {f.ctg_name[k].decode(): seq[offs[k]:offs[k + 1]].copy() for k in range(f.n_ctg)}
    L.vio_fasta_free.argtypes = [C.POINTER(VioFasta)]
    L.vio_fasta_free(out)
    return res
