"""ctypes mirror of include/vcfdist_report.h: the TSV tables and the summary VCF behind the path."""
import ctypes as C

import numpy as np

from . import _abi as A
from . import api

P_f32 = C.POINTER(C.c_float)


class VrpHap(C.Structure):
    _fields_ = [("n_var", C.c_int32), ("pos", A.P_i32), ("type", A.P_u8), ("loc", A.P_u8), ("var_qual", P_f32),
                ("phase_set", A.P_i32), ("ref_len", A.P_i32), ("alt_len", A.P_i32), ("ref_off", A.P_i64), ("alt_off", A.P_i64),
                ("pool", A.P_u8), ("n_cluster", C.c_int32), ("cluster_beg", A.P_i32),
                ("errtype", A.P_u8 * 2), ("credit", P_f32 * 2), ("sync_group", A.P_i32 * 2), ("ref_ed", A.P_i32 * 2),
                ("query_ed", A.P_i32 * 2)]


class VrpContig(C.Structure):
    _fields_ = [("name", C.c_char_p), ("length", C.c_int32), ("ploidy", C.c_int32), ("seq", A.P_u8), ("seq_len", C.c_int64),
                ("hap", VrpHap * 4), ("n_sc", C.c_int32), ("sc_beg", A.P_i32), ("sc_end", A.P_i32), ("sc_brk", A.P_i32 * 4),
                ("sc_phase", A.P_i32), ("pb_phase", A.P_i32), ("orig_phase_dist", A.P_i32), ("swap_phase_dist", A.P_i32),
                ("sc_phase_set", A.P_i32), ("n_pb", C.c_int32), ("phase_block", A.P_i32), ("n_switches", C.c_int32),
                ("n_flips", C.c_int32), ("switches", A.P_i32), ("flips", A.P_i32)]


EXPORTED = ["vrp_phase_blocks", "vrp_write_precision_recall", "vrp_write_phase_blocks", "vrp_write_superclusters",
            "vrp_write_switchflips", "vrp_write_phasing_summary", "vrp_ng50",
            "vrp_write_variants", "vrp_write_summary_vcf", "vrp_last_error"]


class ReportError(RuntimeError):
    pass


def _check(rc, what):
    if rc:
        L = api.lib()
        L.vrp_last_error.restype = C.c_char_p
        raise ReportError(f"{what} failed ({rc}): {L.vrp_last_error().decode()}")


def phase_blocks(sc_phase_set):
    """first supercluster of each phase block + n_sc (phaseblockData ctor, phase.cpp:229-262)"""
    ps = np.ascontiguousarray(sc_phase_set, np.int32)
    out = np.zeros(len(ps) + 1, np.int32)
    L = api.lib()
    L.vrp_phase_blocks.argtypes = [A.P_i32, C.c_int32, A.P_i32]
    n = L.vrp_phase_blocks(A._ptr(ps, C.c_int32), len(ps), A._ptr(out, C.c_int32))
    if n < 0:
        raise ReportError(f"vrp_phase_blocks failed ({n})")
    return out[:n + 1].copy()


_C32 = ("pos", "phase_set", "ref_len", "alt_len")


class Contig:
    """Everything the writers need of one contig.  slots: the four column dicts of vcfdist_amd.io (Q1, Q2, T1, T2);
    clusters: the four cluster tables after superclustering (Superclusters.clusters); sc: Superclusters; res: the
    Results of the path (or any object with the same per-variant arrays); pb / switches / flips: summary.phase()."""

    def __init__(self, name, length, ploidy, seq, slots, clusters, sc, res, sc_phase_set, pb_phase, switches, flips):
        self.keep = []
        k = self._own
        c = VrpContig()
        self.name = name.encode()
        c.name = self.name
        c.length, c.ploidy = int(length), int(ploidy)
        seq = np.frombuffer(seq, np.uint8) if isinstance(seq, (bytes, bytearray)) else np.asarray(seq, np.uint8)
        c.seq, c.seq_len = A._ptr(k(seq, np.uint8), C.c_uint8), len(seq)
        for i, s in enumerate(slots):
            h = c.hap[i]
            n = len(s["pos"])
            h.n_var = n
            for f in _C32:
                setattr(h, f, A._ptr(k(s[f], np.int32), C.c_int32))
            h.type = A._ptr(k(s["type"], np.uint8), C.c_uint8)
            h.var_qual = A._ptr(k(s["var_qual"], np.float32), C.c_float)
            h.ref_off = A._ptr(k(s["ref_off"], np.int64), C.c_int64)
            h.alt_off = A._ptr(k(s["alt_off"], np.int64), C.c_int64)
            h.pool = A._ptr(k(s["pool"] if len(s["pool"]) else np.zeros(1, np.uint8), np.uint8), C.c_uint8)
            cl = clusters[i]
            h.n_cluster = cl.n
            if cl.n:
                h.cluster_beg = A._ptr(k(cl.var_beg, np.int32), C.c_int32)
            for w in range(2):
                h.errtype[w] = A._ptr(k(res.errtype[i][w], np.uint8), C.c_uint8)
                h.credit[w] = A._ptr(k(res.credit[i][w], np.float32), C.c_float)
                h.sync_group[w] = A._ptr(k(res.sync_group[i][w], np.int32), C.c_int32)
                h.ref_ed[w] = A._ptr(k(res.ref_ed[i][w], np.int32), C.c_int32)
                h.query_ed[w] = A._ptr(k(res.query_ed[i][w], np.int32), C.c_int32)
        c.n_sc = sc.n
        i32 = lambda a: A._ptr(k(a, np.int32), C.c_int32)
        c.sc_beg, c.sc_end = i32(sc.beg), i32(sc.end)
        for i in range(4):
            c.sc_brk[i] = i32(sc.brk[i])
        c.sc_phase, c.pb_phase = i32(res.sc_phase), i32(pb_phase)
        c.orig_phase_dist, c.swap_phase_dist = i32(res.orig_phase_dist), i32(res.swap_phase_dist)
        c.sc_phase_set = i32(sc_phase_set)
        self.phase_block = phase_blocks(sc_phase_set)
        c.n_pb = len(self.phase_block) - 1
        c.phase_block = i32(self.phase_block)
        c.n_switches, c.n_flips = len(switches), len(flips)
        c.switches, c.flips = i32(switches), i32(flips)
        self.struct = c

    def _own(self, a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        if a.size == 0:
            a = np.zeros(1, dt)
        self.keep.append(a)
        return a


def _array(contigs):
    arr = (VrpContig * max(len(contigs), 1))()
    for i, c in enumerate(contigs):
        arr[i] = c.struct
    return arr


def write_precision_recall(prefix, counts, min_qual, max_qual):
    cnt = np.ascontiguousarray(counts, np.int64)
    L = api.lib()
    L.vrp_write_precision_recall.argtypes = [C.c_char_p, A.P_i64, C.c_int32, C.c_int32]
    _check(L.vrp_write_precision_recall(prefix.encode(), A._ptr(cnt, C.c_int64), min_qual, max_qual), "vrp_write_precision_recall")


def write_results(prefix, contigs, cmd="", file_date=None, credit_threshold=0.7):
    """phase-blocks.tsv, superclusters.tsv, query.tsv, truth.tsv (write_results, print.cpp:575-878), switchflips.tsv and
    phasing-summary.tsv (phase.cpp:406-528) and summary.vcf"""
    L = api.lib()
    arr, n = _array(contigs), len(contigs)
    P = C.POINTER(VrpContig)
    L.vrp_write_phase_blocks.argtypes = [C.c_char_p, P, C.c_int32]
    L.vrp_write_superclusters.argtypes = [C.c_char_p, P, C.c_int32]
    L.vrp_write_variants.argtypes = [C.c_char_p, P, C.c_int32, C.c_int32]
    L.vrp_write_summary_vcf.argtypes = [C.c_char_p, P, C.c_int32, C.c_char_p, C.c_char_p, C.c_float]
    L.vrp_write_switchflips.argtypes = [C.c_char_p, P, C.c_int32]
    L.vrp_write_phasing_summary.argtypes = [C.c_char_p, P, C.c_int32]
    _check(L.vrp_write_phase_blocks((prefix + "phase-blocks.tsv").encode(), arr, n), "vrp_write_phase_blocks")
    _check(L.vrp_write_switchflips((prefix + "switchflips.tsv").encode(), arr, n), "vrp_write_switchflips")
    _check(L.vrp_write_phasing_summary((prefix + "phasing-summary.tsv").encode(), arr, n), "vrp_write_phasing_summary")
    _check(L.vrp_write_superclusters((prefix + "superclusters.tsv").encode(), arr, n), "vrp_write_superclusters")
    _check(L.vrp_write_variants((prefix + "query.tsv").encode(), arr, n, 0), "vrp_write_variants")
    _check(L.vrp_write_variants((prefix + "truth.tsv").encode(), arr, n, 1), "vrp_write_variants")
    _check(L.vrp_write_summary_vcf((prefix + "summary.vcf").encode(), arr, n, cmd.encode(),
                                   file_date.encode() if file_date else None, credit_threshold), "vrp_write_summary_vcf")


def write_parameters(prefix, args, cmd):
    """parameters.txt (write_params, print.cpp:30-56): the run's settings, one `key = value` per line, in the reference's order and
    formats (strings quoted, booleans true / false, the two thresholds and max_ram with %f).  Keys of stages this implementation
    does not have keep the reference's defaults (realignment off, eval penalties 3 / 2 / 1, distance false, globals.h:41-59)."""
    b2s = lambda b: "true" if b else "false"
    L = api.lib()
    L.vpr_version.restype = C.c_char_p
    text = (
        "program = '%s'\nversion = '%s'\nout_prefix = '%s'\ncommand = '%s'\nreference_fasta = '%s'\n"
        "query_vcf = '%s'\ntruth_vcf = '%s'\nbed_file = '%s'\nwrite_outputs = %s\nfilters = '%s'\n"
        "min_var_qual = %d\nmax_var_qual = %d\nmax_var_size = %d\nsv_threshold = %d\n"
        "phase_threshold = %f\ncredit_threshold = %f\nrealign_truth = %s\nrealign_query = %s\n"
        "realign_only = %s\ncluster_method = '%s'\ncluster_min_gap = %d\n"
        "reach_min_gap = %d\nmax_cluster_itrs = %d\nmax_threads = %d\nmax_ram = %f\n"
        "sub = %d\nopen = %d\nextend = %d\neval_sub = %d\neval_open = %d\neval_extend = %d\ndistance = %s" % (
            "vcfdist_amd", L.vpr_version().decode(), prefix, cmd, args.fasta, args.query, args.truth, args.bed or "",
            b2s(not args.no_output_files), args.filter, args.min_qual, args.max_qual, args.max_size, args.sv_threshold,
            args.phase_threshold, args.credit_threshold, b2s(False), b2s(False), b2s(False), args.cluster, args.cluster_gap,
            args.reach_min_gap, args.max_iterations, 64, 64.0, args.sub, args.open, args.extend, 3, 2, 1, b2s(False)))
    with open(prefix + "parameters.txt", "w") as f:
        f.write(text)
