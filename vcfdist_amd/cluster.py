"""ctypes mirror of include/vcfdist_cluster.h (distance clustering + superclustering).
The same structs are used by the CPU oracle (oracle/cluster_oracle.cpp, prefix vco_) in the tests."""
import ctypes as C

import numpy as np

from . import _abi as A
from . import api

SENTINEL = 0x7fffffff


class VclHap(C.Structure):
    _fields_ = [("n_var", C.c_int32), ("pos", A.P_i32), ("rlen", A.P_i32), ("type", A.P_u8),
                ("ref_len", A.P_i32), ("alt_len", A.P_i32)]


class VclClusters(C.Structure):
    _fields_ = [("n", C.c_int32), ("var_beg", A.P_i32), ("left_reach", A.P_i32), ("right_reach", A.P_i32)]


class VclSuperclusters(C.Structure):
    _fields_ = [("n", C.c_int32), ("brk", A.P_i32 * 4), ("beg", A.P_i32), ("end", A.P_i32),
                ("n_oversize", C.c_int32), ("n_unsplittable", C.c_int32), ("clusters", C.POINTER(VclClusters) * 4)]


class VclHapSeq(C.Structure):
    _fields_ = [("cols", VclHap), ("ref_off", A.P_i64), ("alt_off", A.P_i64), ("pool", A.P_u8)]


class VclWfaStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("align_calls", C.c_int64), ("reach_calls", C.c_int64), ("ms_device", C.c_double)]


EXPORTED = ["vcl_simple_cluster", "vcl_clusters_free", "vcl_supercluster", "vcl_superclusters_free",
            "vcl_supercluster_cells", "vcl_supercluster_cells_all", "vcl_wfa_cluster"]


class Hap:
    """Variant columns of one (callset, hap) on one contig."""

    def __init__(self, pos, rlen, type=None, ref_len=None, alt_len=None):
        self.pos = np.ascontiguousarray(pos, dtype=np.int32)
        self.rlen = np.ascontiguousarray(rlen, dtype=np.int32)
        n = len(self.pos)
        self.type = np.ascontiguousarray(type if type is not None else np.ones(n), dtype=np.uint8)
        self.ref_len = np.ascontiguousarray(ref_len if ref_len is not None else self.rlen, dtype=np.int32)
        self.alt_len = np.ascontiguousarray(alt_len if alt_len is not None else np.ones(n), dtype=np.int32)

    def as_struct(self):
        s = VclHap()
        s.n_var = len(self.pos)
        s.pos = A._ptr(self.pos, C.c_int32)
        s.rlen = A._ptr(self.rlen, C.c_int32)
        s.type = A._ptr(self.type, C.c_uint8)
        s.ref_len = A._ptr(self.ref_len, C.c_int32)
        s.alt_len = A._ptr(self.alt_len, C.c_int32)
        return s


class Clusters:
    def __init__(self, var_beg, left_reach, right_reach):
        self.var_beg = np.ascontiguousarray(var_beg, dtype=np.int32)
        self.left_reach = np.ascontiguousarray(left_reach, dtype=np.int32)
        self.right_reach = np.ascontiguousarray(right_reach, dtype=np.int32)

    @property
    def n(self):
        return max(len(self.var_beg) - 1, 0)

    @classmethod
    def from_struct(cls, s):
        m = s.n + 1 if s.n > 0 else 0
        return cls(A._from_ptr(s.var_beg, m, np.int32).copy(), A._from_ptr(s.left_reach, m, np.int32).copy(),
                   A._from_ptr(s.right_reach, m, np.int32).copy())

    def as_struct(self):
        s = VclClusters()
        s.n = self.n
        s.var_beg = A._ptr(self.var_beg, C.c_int32)
        s.left_reach = A._ptr(self.left_reach, C.c_int32)
        s.right_reach = A._ptr(self.right_reach, C.c_int32)
        return s

    def __eq__(self, o):
        return (np.array_equal(self.var_beg, o.var_beg) and np.array_equal(self.left_reach, o.left_reach)
                and np.array_equal(self.right_reach, o.right_reach))


class Superclusters:
    def __init__(self, s):
        n = s.n
        self.n = n
        self.brk = [A._from_ptr(s.brk[i], n + 1, np.int32).copy() for i in range(4)]
        self.beg = A._from_ptr(s.beg, n, np.int32).copy()
        self.end = A._from_ptr(s.end, n, np.int32).copy()
        self.n_oversize = s.n_oversize
        self.n_unsplittable = s.n_unsplittable
        self.clusters = [Clusters.from_struct(s.clusters[i].contents) for i in range(4)]

    def var_off(self, i):
        """first variant of every supercluster on hap slot i (+ total): vpr_variants.var_off[i]"""
        c = self.clusters[i]
        if c.n == 0:
            return np.zeros(self.n + 1, dtype=np.int64)
        return c.var_beg[self.brk[i]].astype(np.int64)

    def __eq__(self, o):
        return (self.n == o.n and all(np.array_equal(a, b) for a, b in zip(self.brk, o.brk))
                and np.array_equal(self.beg, o.beg) and np.array_equal(self.end, o.end)
                and self.n_oversize == o.n_oversize and self.n_unsplittable == o.n_unsplittable
                and all(a == b for a, b in zip(self.clusters, o.clusters)))


def _bind(L, prefix):
    f = getattr(L, prefix + "_simple_cluster")
    f.argtypes = [C.POINTER(VclHap), C.c_int, C.c_int32, C.c_int32, C.POINTER(C.POINTER(VclClusters))]
    g = getattr(L, prefix + "_supercluster")
    g.argtypes = [VclHap * 4, C.POINTER(VclClusters) * 4, C.c_int32, C.POINTER(C.POINTER(VclSuperclusters))]
    return f, g


def simple_cluster(hap: Hap, size_mode=0, cluster_min_gap=50, reach_min_gap=0, L=None, prefix="vcl"):
    """defaults: vcfdist -c gap 50 ... (globals.h)"""
    L = L or api.lib()
    f, _ = _bind(L, prefix)
    out = C.POINTER(VclClusters)()
    hs = hap.as_struct()
    rc = f(C.byref(hs), size_mode, cluster_min_gap, reach_min_gap, C.byref(out))
    if rc:
        raise ValueError(f"{prefix}_simple_cluster failed: {rc}")
    res = Clusters.from_struct(out.contents)
    if prefix == "vcl":
        L.vcl_clusters_free.argtypes = [C.POINTER(VclClusters)]
        L.vcl_clusters_free(out)
    return res


def supercluster(haps, clusters, max_supercluster_size=10000, L=None, prefix="vcl"):
    L = L or api.lib()
    _, g = _bind(L, prefix)
    hs = (VclHap * 4)(*[h.as_struct() for h in haps])
    keep = [c.as_struct() for c in clusters]
    cs = (C.POINTER(VclClusters) * 4)(*[C.pointer(k) for k in keep])
    out = C.POINTER(VclSuperclusters)()
    rc = g(hs, cs, max_supercluster_size, C.byref(out))
    if rc:
        raise ValueError(f"{prefix}_supercluster failed: {rc}")
    res = Superclusters(out.contents)
    if prefix == "vcl":
        L.vcl_supercluster_cells_all.argtypes = [VclHap * 4, C.POINTER(VclSuperclusters), A.P_i64]
        res.cells = np.zeros(res.n, dtype=np.int64)
        L.vcl_supercluster_cells_all(hs, out, A._ptr(res.cells, C.c_int64))
        L.vcl_superclusters_free.argtypes = [C.POINTER(VclSuperclusters)]
        L.vcl_superclusters_free(out)
    return res


class HapSeq(Hap):
    """A hap's variants with allele strings (for the biWFA clustering)."""

    def __init__(self, pos, type, refs, alts):
        refs = [r.encode() if isinstance(r, str) else bytes(r) for r in refs]
        alts = [a.encode() if isinstance(a, str) else bytes(a) for a in alts]
        ref_len = [len(r) for r in refs]
        alt_len = [len(a) for a in alts]
        super().__init__(pos, ref_len, type, ref_len, alt_len)
        pool = bytearray()
        self.ref_off, self.alt_off = [], []
        for r, a in zip(refs, alts):
            self.ref_off.append(len(pool)); pool += r
            self.alt_off.append(len(pool)); pool += a
        self.ref_off = np.ascontiguousarray(self.ref_off, dtype=np.int64)
        self.alt_off = np.ascontiguousarray(self.alt_off, dtype=np.int64)
        self.pool = np.frombuffer(bytes(pool) + b"\0", dtype=np.uint8).copy()

    def as_seq_struct(self):
        s = VclHapSeq()
        s.cols = self.as_struct()
        s.ref_off = A._ptr(self.ref_off, C.c_int64)
        s.alt_off = A._ptr(self.alt_off, C.c_int64)
        s.pool = A._ptr(self.pool, C.c_uint8)
        return s


def wfa_cluster(hap: HapSeq, ctg, sub=5, open=6, extend=2, max_cluster_itrs=4, reach_min_gap=10, device=0, L=None,
                prefix="vcl"):
    """defaults: vcfdist -c biwfa (globals.h:33-43).  Returns (Clusters, stats dict)."""
    L = L or api.lib()
    ctg = np.frombuffer(ctg.encode() if isinstance(ctg, str) else bytes(ctg), dtype=np.uint8).copy()
    hs = hap.as_seq_struct()
    out = C.POINTER(VclClusters)()
    st = VclWfaStats()
    f = getattr(L, prefix + "_wfa_cluster")
    if prefix == "vcl":
        f.argtypes = [C.POINTER(VclHapSeq), A.P_u8, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                      C.c_int32, C.POINTER(C.POINTER(VclClusters)), C.POINTER(VclWfaStats)]
        rc = f(C.byref(hs), A._ptr(ctg, C.c_uint8), len(ctg), sub, open, extend, max_cluster_itrs, reach_min_gap, device,
               C.byref(out), C.byref(st))
    else:
        f.argtypes = [C.POINTER(VclHapSeq), A.P_u8, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                      C.POINTER(C.POINTER(VclClusters)), C.POINTER(VclWfaStats)]
        rc = f(C.byref(hs), A._ptr(ctg, C.c_uint8), len(ctg), sub, open, extend, max_cluster_itrs, reach_min_gap,
               C.byref(out), C.byref(st))
    if rc:
        raise ValueError(f"{prefix}_wfa_cluster failed: {rc}")
    res = Clusters.from_struct(out.contents)
    if prefix == "vcl":
        L.vcl_clusters_free.argtypes = [C.POINTER(VclClusters)]
        L.vcl_clusters_free(out)
    return res, dict(iterations=st.iterations, align_calls=st.align_calls, reach_calls=st.reach_calls, ms_device=st.ms_device)
