"""ctypes mirror of include/vcfdist_pr.h (plain-data structs only) and numpy
containers for the flat batch / result layouts.  No compute lives here."""
import ctypes as C

import numpy as np

HAPS = 4
ALNS = 4

PTR_VARIANT, PTR_VAR_BEG, PTR_VAR_END, PTR_INS_LOC = 1, 2, 4, 8
TYPE_SUB, TYPE_INS, TYPE_DEL = 1, 2, 3
ERRTYPE_TP, ERRTYPE_FP, ERRTYPE_FN, ERRTYPE_UN = 0, 1, 2, 5
PHASE_ORIG, PHASE_SWAP, PHASE_NONE = 0, 1, 2
PLANE_QUERY, PLANE_REF = 0, 1

ST_SWAP_TIE = 1
ST_WARN_REF_ED = 2
ST_WARN_QUERY_ED = 4
ST_WARN_EXCEEDS = 8
ST_WARN_ZERO_ED = 16
ST_ERR_NO_PTR = 32
ST_ERR_UNFINISHED = 64
ST_ERR_LIMIT = 128
ST_WARN_MASK = ST_WARN_REF_ED | ST_WARN_QUERY_ED | ST_WARN_EXCEEDS | ST_WARN_ZERO_ED

P_i64 = C.POINTER(C.c_int64)
P_i32 = C.POINTER(C.c_int32)
P_u32 = C.POINTER(C.c_uint32)
P_u8 = C.POINTER(C.c_uint8)
P_i16 = C.POINTER(C.c_int16)
P_f32 = C.POINTER(C.c_float)


class VprBatch(C.Structure):
    _fields_ = [
        ("n_sc", C.c_int32),
        ("hap_off", P_i64 * HAPS), ("hap_seq", P_u8 * HAPS),
        ("hap_ptr", P_i32 * HAPS), ("hap_flag", P_u8 * HAPS),
        ("ref_off", P_i64), ("ref_seq", P_u8),
        ("ref_ptr", P_i32 * 2), ("ref_flag", P_u8 * 2),
        ("var_off", P_i64 * HAPS), ("var_pos", P_i32 * HAPS), ("var_qual", P_f32 * HAPS),
    ]


class VprVariants(C.Structure):
    _fields_ = [
        ("n_sc", C.c_int32), ("n_ctg", C.c_int32),
        ("ctg_off", P_i64), ("ctg_seq", P_u8),
        ("sc_ctg", P_i32), ("sc_beg", P_i32), ("sc_end", P_i32),
        ("var_off", P_i64 * HAPS), ("var_pos", P_i32 * HAPS), ("var_type", P_u8 * HAPS),
        ("var_qual", P_f32 * HAPS), ("var_ref_off", P_i64 * HAPS), ("var_ref_len", P_i32 * HAPS),
        ("var_alt_off", P_i64 * HAPS), ("var_alt_len", P_i32 * HAPS),
        ("allele_pool", P_u8 * HAPS),
    ]


class VprConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("max_qual", C.c_float),
        ("credit_threshold", C.c_double), ("phase_threshold", C.c_double),
        ("workspace_bytes", C.c_int64), ("band_mode", C.c_int32), ("flags", C.c_int32),
    ]


class VprResults(C.Structure):
    _fields_ = [
        ("aln_dist", P_i32), ("aln_end_plane", P_u8), ("aln_beg_plane", P_u8), ("aln_status", P_u32),
        ("sc_phase", P_i32), ("orig_phase_dist", P_i32), ("swap_phase_dist", P_i32),
        ("errtype", (P_u8 * 2) * HAPS), ("sync_group", (P_i32 * 2) * HAPS),
        ("credit", (P_f32 * 2) * HAPS), ("ref_ed", (P_i32 * 2) * HAPS),
        ("query_ed", (P_i32 * 2) * HAPS), ("callq", (P_f32 * 2) * HAPS),
    ]


CFG_DENSE_S16 = 1   # VPR_CFG_DENSE_S16
CFG_TIE_SMALL_LOGS = 2   # VPR_CFG_TIE_SMALL_LOGS
CFG_GUARD_ALLOC = 4      # VPR_CFG_GUARD_ALLOC
CFG_KEEP_PATHS = 8       # VPR_CFG_KEEP_PATHS
CFG_HAP_DEDUP = 16       # VPR_CFG_HAP_DEDUP


class VprTiming(C.Structure):
    _fields_ = [
        ("ms_total", C.c_double), ("ms_prep", C.c_double), ("ms_fwd", C.c_double),
        ("ms_bwd", C.c_double), ("ms_walk", C.c_double), ("ms_ed", C.c_double),
        ("n_fwd_launches", C.c_int64), ("cells_dense", C.c_int64), ("cells_touched", C.c_int64),
        ("bytes_algorithmic", C.c_int64), ("n_band_retries", C.c_int64),
        ("n_tie_replays", C.c_int64), ("ms_tie", C.c_double),
        ("ms_wall", C.c_double), ("ms_wall_phase", C.c_double * 6), ("ms_host_alloc", C.c_double),
        ("ms_host_blocked", C.c_double), ("n_alignments_computed", C.c_int64), ("n_device_allocs", C.c_int64), ("n_device_frees", C.c_int64),
        ("n_host_allocs", C.c_int64), ("n_lane1_seen", C.c_int64), ("n_lane1_finished", C.c_int64), ("n_lane1_waves_dropped", C.c_int64),
    ]


class VprLaunchStat(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("threads", C.c_int32), ("cells_per_thread", C.c_int32), ("n_units", C.c_int32),
        ("cells", C.c_int64), ("bytes_algorithmic", C.c_int64), ("ms", C.c_double),
        ("cells_dense", C.c_int64), ("kernel", C.c_char * 32),
    ]


class VprSynthParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("n_sc", C.c_int32), ("len_mode", C.c_int32),
        ("len_a", C.c_double), ("len_b", C.c_double),
        ("len_min", C.c_int32), ("len_max", C.c_int32),
        ("p_repeat", C.c_double), ("var_per_base", C.c_double), ("p_snp", C.c_double),
        ("indel_mean", C.c_double), ("p_hom", C.c_double), ("p_keep", C.c_double),
        ("p_drop", C.c_double), ("max_qual", C.c_int32), ("reserved", C.c_int32),
        ("p_sv", C.c_double), ("sv_min", C.c_int32), ("sv_max", C.c_int32),
    ]


def default_config(device=0, band_mode=1, workspace_bytes=0, flags=0):
    """The reference's defaults: globals.h:27 (max_qual), :49 (credit), :46 (phase)."""
    return VprConfig(device=device, max_qual=60.0, credit_threshold=0.7, phase_threshold=0.6,
                     workspace_bytes=workspace_bytes, band_mode=band_mode, flags=flags)


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def _arr(x, dtype):
    return np.ascontiguousarray(np.asarray(x, dtype=dtype))


_COPY = True   # module switch used by from_struct(copy=False): view C memory instead of copying


def _from_ptr(p, n, dtype):
    """n elements behind a ctypes pointer as a numpy array (a copy unless _COPY is off)."""
    if n == 0:
        return np.zeros(0, dtype=dtype)
    addr = C.cast(p, C.c_void_p).value
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
    a = np.frombuffer(buf, dtype=dtype, count=n)
    return a.copy() if _COPY else a


class Batch:
    """Level A batch (outputs of generate_ptrs_strs for every supercluster)."""

    def __init__(self, n_sc, hap_off, hap_seq, hap_ptr, hap_flag, ref_off, ref_seq, ref_ptr,
                 ref_flag, var_off, var_pos, var_qual):
        self.n_sc = int(n_sc)
        self.hap_off = [_arr(a, np.int64) for a in hap_off]
        self.hap_seq = [_arr(a, np.uint8) for a in hap_seq]
        self.hap_ptr = [_arr(a, np.int32) for a in hap_ptr]
        self.hap_flag = [_arr(a, np.uint8) for a in hap_flag]
        self.ref_off = _arr(ref_off, np.int64)
        self.ref_seq = _arr(ref_seq, np.uint8)
        self.ref_ptr = [_arr(a, np.int32) for a in ref_ptr]
        self.ref_flag = [_arr(a, np.uint8) for a in ref_flag]
        self.var_off = [_arr(a, np.int64) for a in var_off]
        self.var_pos = [_arr(a, np.int32) for a in var_pos]
        self.var_qual = [_arr(a, np.float32) for a in var_qual]

    @classmethod
    def from_struct(cls, s, copy=True, owner=None):
        """Numpy view of a vpr_batch (e.g. of a vpr_owned_batch).  copy=False keeps views into the
        C memory; `owner` is then stored on the Batch to keep that memory alive."""
        global _COPY
        _COPY = copy
        try:
            b = cls._from_struct(s)
        finally:
            _COPY = True
        b._owner = owner
        return b

    @classmethod
    def _from_struct(cls, s):
        n = s.n_sc
        hap_off = [_from_ptr(s.hap_off[h], n + 1, np.int64) for h in range(HAPS)]
        ref_off = _from_ptr(s.ref_off, n + 1, np.int64)
        var_off = [_from_ptr(s.var_off[h], n + 1, np.int64) for h in range(HAPS)]
        hl = [int(o[-1]) for o in hap_off]
        rl = int(ref_off[-1])
        vl = [int(o[-1]) for o in var_off]
        return cls(n, hap_off,
                   [_from_ptr(s.hap_seq[h], hl[h], np.uint8) for h in range(HAPS)],
                   [_from_ptr(s.hap_ptr[h], hl[h], np.int32) for h in range(HAPS)],
                   [_from_ptr(s.hap_flag[h], hl[h], np.uint8) for h in range(HAPS)],
                   ref_off, _from_ptr(s.ref_seq, rl, np.uint8),
                   [_from_ptr(s.ref_ptr[h], rl, np.int32) for h in range(2)],
                   [_from_ptr(s.ref_flag[h], rl, np.uint8) for h in range(2)],
                   var_off,
                   [_from_ptr(s.var_pos[h], vl[h], np.int32) for h in range(HAPS)],
                   [_from_ptr(s.var_qual[h], vl[h], np.float32) for h in range(HAPS)])

    def as_struct(self):
        s = VprBatch()
        s.n_sc = self.n_sc
        for h in range(HAPS):
            s.hap_off[h] = _ptr(self.hap_off[h], C.c_int64)
            s.hap_seq[h] = _ptr(self.hap_seq[h], C.c_uint8)
            s.hap_ptr[h] = _ptr(self.hap_ptr[h], C.c_int32)
            s.hap_flag[h] = _ptr(self.hap_flag[h], C.c_uint8)
            s.var_off[h] = _ptr(self.var_off[h], C.c_int64)
            s.var_pos[h] = _ptr(self.var_pos[h], C.c_int32)
            s.var_qual[h] = _ptr(self.var_qual[h], C.c_float)
        s.ref_off = _ptr(self.ref_off, C.c_int64)
        s.ref_seq = _ptr(self.ref_seq, C.c_uint8)
        for h in range(2):
            s.ref_ptr[h] = _ptr(self.ref_ptr[h], C.c_int32)
            s.ref_flag[h] = _ptr(self.ref_flag[h], C.c_uint8)
        return s

    def n_vars(self, h):
        return int(self.var_off[h][-1])

    def lens(self, sc):
        """(Lq1, Lq2, Lt1, Lt2, Lr) of one supercluster."""
        l = [int(self.hap_off[h][sc + 1] - self.hap_off[h][sc]) for h in range(HAPS)]
        return l + [int(self.ref_off[sc + 1] - self.ref_off[sc])]

    def dense_cells(self):
        """sum_i (Lq_i + Lr) * Lt_i  (SURVEY.md 8(d))."""
        lq = [np.diff(self.hap_off[h]) for h in range(2)]
        lt = [np.diff(self.hap_off[2 + h]) for h in range(2)]
        lr = np.diff(self.ref_off)
        tot = 0
        for i in range(4):
            tot += int(((lq[i >> 1] + lr) * lt[i & 1]).sum())
        return tot

    def subset(self, idx):
        """A new Batch holding the superclusters `idx` (any order)."""
        idx = np.asarray(idx, dtype=np.int64)

        def take(off, arrs):
            lens = off[idx + 1] - off[idx]
            new_off = np.zeros(len(idx) + 1, dtype=np.int64)
            np.cumsum(lens, out=new_off[1:])
            sel = np.concatenate([np.arange(off[i], off[i + 1]) for i in idx]) if len(idx) else np.zeros(0, np.int64)
            return new_off, [a[sel] for a in arrs]

        hap_off, hap_seq, hap_ptr, hap_flag, var_off, var_pos, var_qual = [], [], [], [], [], [], []
        for h in range(HAPS):
            o, (s, p, f) = take(self.hap_off[h], [self.hap_seq[h], self.hap_ptr[h], self.hap_flag[h]])
            hap_off.append(o); hap_seq.append(s); hap_ptr.append(p); hap_flag.append(f)
            o, (vp, vq) = take(self.var_off[h], [self.var_pos[h], self.var_qual[h]])
            var_off.append(o); var_pos.append(vp); var_qual.append(vq)
        ref_off, (rs, rp0, rp1, rf0, rf1) = take(
            self.ref_off, [self.ref_seq, self.ref_ptr[0], self.ref_ptr[1], self.ref_flag[0], self.ref_flag[1]])
        return Batch(len(idx), hap_off, hap_seq, hap_ptr, hap_flag, ref_off, rs, [rp0, rp1], [rf0, rf1],
                     var_off, var_pos, var_qual)


class Variants:
    """Level B input (variants + reference)."""

    def __init__(self, ctg_off, ctg_seq, sc_ctg, sc_beg, sc_end, var_off, var_pos, var_type, var_qual,
                 var_ref_off, var_ref_len, var_alt_off, var_alt_len, allele_pool):
        self.n_sc = len(sc_beg)
        self.ctg_off = _arr(ctg_off, np.int64)
        self.ctg_seq = _arr(ctg_seq, np.uint8)
        self.sc_ctg = _arr(sc_ctg, np.int32)
        self.sc_beg = _arr(sc_beg, np.int32)
        self.sc_end = _arr(sc_end, np.int32)
        self.var_off = [_arr(a, np.int64) for a in var_off]
        self.var_pos = [_arr(a, np.int32) for a in var_pos]
        self.var_type = [_arr(a, np.uint8) for a in var_type]
        self.var_qual = [_arr(a, np.float32) for a in var_qual]
        self.var_ref_off = [_arr(a, np.int64) for a in var_ref_off]
        self.var_ref_len = [_arr(a, np.int32) for a in var_ref_len]
        self.var_alt_off = [_arr(a, np.int64) for a in var_alt_off]
        self.var_alt_len = [_arr(a, np.int32) for a in var_alt_len]
        self.allele_pool = [_arr(a, np.uint8) for a in allele_pool]

    @classmethod
    def from_struct(cls, s):
        n = s.n_sc
        ctg_off = _from_ptr(s.ctg_off, s.n_ctg + 1, np.int64)
        var_off = [_from_ptr(s.var_off[h], n + 1, np.int64) for h in range(HAPS)]
        nv = [int(o[-1]) for o in var_off]
        ref_off = [_from_ptr(s.var_ref_off[h], nv[h], np.int64) for h in range(HAPS)]
        alt_off = [_from_ptr(s.var_alt_off[h], nv[h], np.int64) for h in range(HAPS)]
        ref_len = [_from_ptr(s.var_ref_len[h], nv[h], np.int32) for h in range(HAPS)]
        alt_len = [_from_ptr(s.var_alt_len[h], nv[h], np.int32) for h in range(HAPS)]
        pool_len = [int(max((ref_off[h] + ref_len[h]).max(initial=0), (alt_off[h] + alt_len[h]).max(initial=0)))
                    for h in range(HAPS)]
        return cls(ctg_off, _from_ptr(s.ctg_seq, int(ctg_off[-1]), np.uint8),
                   _from_ptr(s.sc_ctg, n, np.int32), _from_ptr(s.sc_beg, n, np.int32),
                   _from_ptr(s.sc_end, n, np.int32), var_off,
                   [_from_ptr(s.var_pos[h], nv[h], np.int32) for h in range(HAPS)],
                   [_from_ptr(s.var_type[h], nv[h], np.uint8) for h in range(HAPS)],
                   [_from_ptr(s.var_qual[h], nv[h], np.float32) for h in range(HAPS)],
                   ref_off, ref_len, alt_off, alt_len,
                   [_from_ptr(s.allele_pool[h], pool_len[h], np.uint8) for h in range(HAPS)])

    @classmethod
    def from_sites(cls, contigs, superclusters):
        """Small-case constructor.  contigs: list of str.  superclusters: list of
        dicts {ctg, beg, end, vars: [list per hap slot of (pos, type, ref, alt, qual)]}."""
        ctg_off = np.zeros(len(contigs) + 1, dtype=np.int64)
        for i, c in enumerate(contigs):
            ctg_off[i + 1] = ctg_off[i] + len(c)
        ctg_seq = np.frombuffer("".join(contigs).encode(), dtype=np.uint8)
        var_off = [[0] for _ in range(HAPS)]
        pos = [[] for _ in range(HAPS)]; typ = [[] for _ in range(HAPS)]; qual = [[] for _ in range(HAPS)]
        roff = [[] for _ in range(HAPS)]; aoff = [[] for _ in range(HAPS)]; pool = [bytearray() for _ in range(HAPS)]
        rlen = [[] for _ in range(HAPS)]; alen = [[] for _ in range(HAPS)]
        for sc in superclusters:
            for h in range(HAPS):
                for (p, t, r, a, q) in sc["vars"][h]:
                    pos[h].append(p); typ[h].append(t); qual[h].append(q)
                    roff[h].append(len(pool[h])); rlen[h].append(len(r)); pool[h] += r.encode()
                    aoff[h].append(len(pool[h])); alen[h].append(len(a)); pool[h] += a.encode()
                var_off[h].append(len(pos[h]))
        return cls(ctg_off, ctg_seq, [s["ctg"] for s in superclusters], [s["beg"] for s in superclusters],
                   [s["end"] for s in superclusters], var_off, pos, typ, qual, roff, rlen, aoff, alen,
                   [np.frombuffer(bytes(p), dtype=np.uint8) for p in pool])

    def n_vars(self, h):
        return int(self.var_off[h][-1])

    def as_struct(self):
        s = VprVariants()
        s.n_sc = self.n_sc
        s.n_ctg = len(self.ctg_off) - 1
        s.ctg_off = _ptr(self.ctg_off, C.c_int64)
        s.ctg_seq = _ptr(self.ctg_seq, C.c_uint8)
        s.sc_ctg = _ptr(self.sc_ctg, C.c_int32)
        s.sc_beg = _ptr(self.sc_beg, C.c_int32)
        s.sc_end = _ptr(self.sc_end, C.c_int32)
        for h in range(HAPS):
            s.var_off[h] = _ptr(self.var_off[h], C.c_int64)
            s.var_pos[h] = _ptr(self.var_pos[h], C.c_int32)
            s.var_type[h] = _ptr(self.var_type[h], C.c_uint8)
            s.var_qual[h] = _ptr(self.var_qual[h], C.c_float)
            s.var_ref_off[h] = _ptr(self.var_ref_off[h], C.c_int64)
            s.var_alt_off[h] = _ptr(self.var_alt_off[h], C.c_int64)
            s.var_ref_len[h] = _ptr(self.var_ref_len[h], C.c_int32)
            s.var_alt_len[h] = _ptr(self.var_alt_len[h], C.c_int32)
            s.allele_pool[h] = _ptr(self.allele_pool[h], C.c_uint8)
        return s


class _HostBlock:
    """one page-locked allocation, released when the last array viewing it goes away"""

    def __init__(self, ptr, free):
        self.ptr, self.free = ptr, free

    def __del__(self):
        if self.ptr and self.free is not None:
            self.free(self.ptr)
            self.ptr = None


class Results:
    """Result buffers, initialised to the reference's initial values
    (errtype UN, everything else 0; variant.cpp:45-52, cluster.cpp:26-29)."""

    PER_VAR = (("errtype", np.uint8), ("sync_group", np.int32), ("credit", np.float32),
               ("ref_ed", np.int32), ("query_ed", np.int32), ("callq", np.float32))

    def __init__(self, n_sc, n_vars, host_alloc=None, host_free=None):
        """host_alloc / host_free: optional allocator of page-locked memory (vpr_host_alloc / vpr_host_free of the library);
        without it the buffers are ordinary numpy arrays"""
        self.n_sc = n_sc

        def full(n, value, dt):
            n = int(n)
            if host_alloc is None or n == 0:
                return np.full(n, value, dt)
            p = host_alloc(n * np.dtype(dt).itemsize)
            if not p:
                return np.full(n, value, dt)
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(p)
            buf._block = _HostBlock(p, host_free)     # the array (through its base) keeps the block alive, not this object
            a = np.frombuffer(buf, dtype=dt, count=n)
            a[:] = value
            return a
        self.aln_dist = full(n_sc * 4, 0, np.int32)
        self.aln_end_plane = full(n_sc * 4, 0, np.uint8)
        self.aln_beg_plane = full(n_sc * 4, 0, np.uint8)
        self.aln_status = full(n_sc * 4, 0, np.uint32)
        self.sc_phase = full(n_sc, PHASE_NONE, np.int32)
        self.orig_phase_dist = full(n_sc, -1, np.int32)
        self.swap_phase_dist = full(n_sc, -1, np.int32)
        for name, dt in self.PER_VAR:
            init = ERRTYPE_UN if name == "errtype" else 0
            setattr(self, name, [[full(n_vars[h], init, dt) for _ in range(2)] for h in range(HAPS)])

    @classmethod
    def for_batch(cls, batch, host_alloc=None, host_free=None):
        return cls(batch.n_sc, [batch.n_vars(h) for h in range(HAPS)], host_alloc, host_free)

    @classmethod
    def mirror(cls, s, n_sc, n_vars, block, host_free):
        """Results whose arrays are views into ONE page-locked block laid out like the device's result columns
        (vpr_results_alloc filled the struct `s`): vpr_download then moves everything with a single copy.  Every array keeps
        the block alive."""
        self = cls.__new__(cls)
        self.n_sc = n_sc
        owner = _HostBlock(block, host_free)

        def view(p, n, dt):
            n = int(n)
            if n == 0:
                return np.zeros(0, dt)
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(C.cast(p, C.c_void_p).value)
            buf._block = owner
            return np.frombuffer(buf, dtype=dt, count=n)
        self.aln_dist = view(s.aln_dist, n_sc * 4, np.int32)
        self.aln_end_plane = view(s.aln_end_plane, n_sc * 4, np.uint8)
        self.aln_beg_plane = view(s.aln_beg_plane, n_sc * 4, np.uint8)
        self.aln_status = view(s.aln_status, n_sc * 4, np.uint32)
        self.sc_phase = view(s.sc_phase, n_sc, np.int32)
        self.orig_phase_dist = view(s.orig_phase_dist, n_sc, np.int32)
        self.swap_phase_dist = view(s.swap_phase_dist, n_sc, np.int32)
        for name, dt in cls.PER_VAR:
            field = getattr(s, name)
            setattr(self, name, [[view(field[h][w], n_vars[h], dt) for w in range(2)] for h in range(HAPS)])
        return self

    def as_struct(self):
        s = VprResults()
        s.aln_dist = _ptr(self.aln_dist, C.c_int32)
        s.aln_end_plane = _ptr(self.aln_end_plane, C.c_uint8)
        s.aln_beg_plane = _ptr(self.aln_beg_plane, C.c_uint8)
        s.aln_status = _ptr(self.aln_status, C.c_uint32)
        s.sc_phase = _ptr(self.sc_phase, C.c_int32)
        s.orig_phase_dist = _ptr(self.orig_phase_dist, C.c_int32)
        s.swap_phase_dist = _ptr(self.swap_phase_dist, C.c_int32)
        ct = {"errtype": C.c_uint8, "sync_group": C.c_int32, "credit": C.c_float,
              "ref_ed": C.c_int32, "query_ed": C.c_int32, "callq": C.c_float}
        for name, _ in self.PER_VAR:
            field = getattr(s, name)
            arrs = getattr(self, name)
            for h in range(HAPS):
                for w in range(2):
                    field[h][w] = _ptr(arrs[h][w], ct[name])
        return s

    def diff(self, other, skip_status_mask=0):
        """List of human-readable mismatches against another Results (bit-exact compare)."""
        out = []
        for name in ("aln_dist", "aln_end_plane", "aln_beg_plane", "sc_phase", "orig_phase_dist",
                     "swap_phase_dist"):
            a, b = getattr(self, name), getattr(other, name)
            bad = np.nonzero(a != b)[0]
            if len(bad):
                out.append(f"{name}: {len(bad)} mismatches, first at {bad[0]}: {a[bad[0]]} vs {b[bad[0]]}")
        sa = self.aln_status & ~np.uint32(skip_status_mask)
        sb = other.aln_status & ~np.uint32(skip_status_mask)
        bad = np.nonzero(sa != sb)[0]
        if len(bad):
            out.append(f"aln_status: {len(bad)} mismatches, first at {bad[0]}: {sa[bad[0]]} vs {sb[bad[0]]}")
        for name, dt in self.PER_VAR:
            for h in range(HAPS):
                for w in range(2):
                    a, b = getattr(self, name)[h][w], getattr(other, name)[h][w]
                    if dt == np.float32:  # bit-exact float compare
                        a, b = a.view(np.uint32), b.view(np.uint32)
                    bad = np.nonzero(a != b)[0]
                    if len(bad):
                        out.append(f"{name}[hap {h}][swap {w}]: {len(bad)} mismatches, first at var {bad[0]}: "
                                   f"{getattr(self, name)[h][w][bad[0]]} vs {getattr(other, name)[h][w][bad[0]]}")
        return out
