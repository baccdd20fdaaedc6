"""Test infrastructure: a pure-Python restatement of the reference's output writers, used only to check
vcfdist_amd/csrc/report.cpp byte for byte.  Follows write_precision_recall / write_results (src/print.cpp:441-878),
phaseblockData::write_summary_vcf (src/phase.cpp:8-222) and ctgVariants::print_var_* (src/variant.cpp:229-286).
Python's % formatting of a float32 widened to double prints the same digits as C's printf.

PARITY UNPINNED for the file formats: the reference ships no example of these files and cannot be built here, so the
format strings, merge order and header text below are pinned only by reading the cited lines; the one piece of
reference output that touches them (demo/output.txt, SNP rows) agrees with precision-recall-summary.tsv.
"""
import math

import numpy as np

TYPE_STR = ["REF", "SNP", "INS", "DEL", "CPX"]
VARTYPE_STR = ["SNP", "INDEL", "SV", "ALL"]
ERROR_STR = ["TP", "FP", "FN", "PE", "GE", "??"]
REGION_STR = ["OUTSIDE", "INSIDE ", "BORDER ", "OFF CTG"]
PHASE_STR = ["=", "X", "?"]
TP, FP, FN = 0, 1, 2
f32 = np.float32
INT_MAX = 2**31 - 1


def qscore(p):
    return float(f32(min(100.0, max(0.0, -10 * math.log10(p) if p > 0 else float("inf")))))


def _metrics(counts, t, qi):
    qtp, qfp = int(counts[0, t, TP, qi]), int(counts[0, t, FP, qi])
    ttp, tfn = int(counts[1, t, TP, qi]), int(counts[1, t, FN, qi])
    prec = f32(1) if qtp + qfp == 0 else f32(qtp) / f32(qtp + qfp)
    rec = f32(1) if ttp + tfn == 0 else f32(ttp) / f32(ttp + tfn)
    f1 = f32(2) * prec * rec / (prec + rec) if prec + rec > 0 else f32(0)
    return qtp, qfp, ttp, tfn, prec, rec, f1


def precision_recall(counts, min_qual, max_qual):
    """-> (precision-recall.tsv text, precision-recall-summary.tsv text)"""
    a = ["VAR_TYPE\tMIN_QUAL\tPREC\tRECALL\tF1_SCORE\tF1_QSCORE\tTRUTH_TOTAL\tTRUTH_TP\tTRUTH_FN\tQUERY_TOTAL\tQUERY_TP\tQUERY_FP\n"]
    best_q = [0] * 4
    for t in range(4):
        best = f32(0)
        for q in range(min_qual, max_qual + 1):
            qtp, qfp, ttp, tfn, prec, rec, f1 = _metrics(counts, t, q - min_qual)
            if f1 > best:
                best, best_q[t] = f1, q
            a.append("%s\t%d\t%f\t%f\t%f\t%f\t%d\t%d\t%d\t%d\t%d\t%d\n" % (VARTYPE_STR[t], q, prec, rec, f1, qscore(float(f32(1) - f1)),
                                                                     ttp + tfn, ttp, tfn, qtp + qfp, qtp, qfp))
        if not min_qual <= best_q[t] <= max_qual:
            best_q[t] = min_qual
    s = ["VAR_TYPE\tTHRESHOLD\tMIN_QUAL\tTRUTH_TP\tQUERY_TP\tTRUTH_FN\tQUERY_FP\tPREC\tRECALL\tF1_SCORE\tF1_QSCORE\n"]
    for t in range(4):
        for thr, q in (("NONE", min_qual), ("BEST", best_q[t])):
            qtp, qfp, ttp, tfn, prec, rec, f1 = _metrics(counts, t, q - min_qual)
            s.append("%s\t%s\t%d\t%d\t%d\t%d\t%d\t%f\t%f\t%f\t%f\n" % (VARTYPE_STR[t], thr, q, ttp, qtp, tfn, qfp, prec, rec, f1,
                                                                 qscore(float(f32(1) - f1))))
    return "".join(a), "".join(s)


def phase_blocks(sc_phase_set):
    out, cur = [], -1
    for k, ps in enumerate(sc_phase_set):
        if ps != cur:
            out.append(k)
            cur = ps
    out.append(len(sc_phase_set))
    return out


class Ctg:
    """plain-data view of one contig: name, length, ploidy, seq (bytes), slots (4 column dicts incl. 'ref'/'alt' lists of
    str), clusters (4 lists: var_beg incl. sentinel, [] for an empty hap), sc_beg/sc_end/sc_brk, sc_phase, pb_phase,
    orig/swap dist, sc_phase_set, switches, flips, res (errtype/credit/sync_group/ref_ed/query_ed [slot][swap])"""

    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.pbs = phase_blocks(self.sc_phase_set)


def phase_blocks_tsv(ctgs):
    o = ["CONTIG\tPHASE_BLOCK\tSTART\tSTOP\tSIZE\tSUPERCLUSTERS\tFLIP_ERRORS\tSWITCH_ERRORS\n"]
    for c in ctgs:
        n_sc = len(c.sc_beg)
        for i in range(len(c.pbs) - 1):
            if n_sc == 0:
                break
            b, e = c.pbs[i], c.pbs[i + 1] - 1
            nsw = sum(1 for s in c.switches if b < s <= e)
            nfl = sum(1 for s in c.flips if b < s <= e)
            o.append("%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n" % (c.name, i, c.sc_beg[b], c.sc_end[e], c.sc_end[e] - c.sc_beg[b], e - b + 1, nfl, nsw))
    return "".join(o)


SW_FLIP, SW_SWITCH, SW_BOTH, SW_ERR, SW_NONE = 0, 1, 2, 3, 6     # defs.h:74-80


def _breaks(c, on_switch, on_flip):
    """the loop skeleton shared by write_switchflips (phase.cpp:424-505) and calculate_ng50 (phase.cpp:561-612):
    yields (type, next_sc, pb_idx - 1); raises where the reference's ERROR() fires"""
    n = len(c.sc_beg)
    switch_idx = flip_idx = 0
    pb_idx, sc = 1, 0
    n_pb = len(c.pbs) - 1
    while True:
        next_sc, typ = n, SW_NONE
        if pb_idx < n_pb and c.pbs[pb_idx] <= next_sc:
            typ, next_sc = SW_SWITCH, c.pbs[pb_idx]
        if on_switch and switch_idx < len(c.switches) and c.switches[switch_idx] <= next_sc:
            typ, next_sc = SW_ERR, c.switches[switch_idx]
        if on_flip and flip_idx < len(c.flips) and c.flips[flip_idx] <= next_sc:
            typ = SW_BOTH if typ == SW_SWITCH else SW_FLIP
            next_sc = c.flips[flip_idx]
        if typ == SW_NONE:
            return
        if next_sc <= sc:
            raise ValueError("Next supercluster is not after current supercluster")
        yield typ, int(next_sc), pb_idx - 1
        if typ in (SW_FLIP, SW_BOTH):
            flip_idx += 1
            if typ == SW_BOTH:
                pb_idx += 1
        elif typ == SW_SWITCH:
            pb_idx += 1
        else:
            switch_idx += 1
        sc = next_sc


def switchflips_tsv(ctgs):
    """phaseblockData::write_switchflips, phase.cpp:406-509"""
    o = ["CONTIG\tSTART\tSTOP\tSWITCH_TYPE\tSUPERCLUSTER\tPHASE_BLOCK\n"]
    for c in ctgs:
        n = len(c.sc_beg)
        if n == 0:
            continue
        for typ, next_sc, pb in _breaks(c, True, True):
            if typ in (SW_FLIP, SW_BOTH):
                left = next_sc - 1
                while left > 0 and c.sc_phase[left] == 2:      # PHASE_NONE
                    left -= 1
                if left >= 0:
                    o.append("%s\t%d\t%d\t%s\t%d\t%d\n" % (c.name, c.sc_end[left], c.sc_beg[next_sc], "FLIP_BEG", next_sc, pb))
                right = next_sc + 1
                while right < n - 1 and c.sc_phase[right] == 2:
                    right += 1
                if right < n:
                    o.append("%s\t%d\t%d\t%s\t%d\t%d\n" % (c.name, c.sc_end[next_sc], c.sc_beg[right], "FLIP_END", next_sc, pb))
            elif typ == SW_ERR:
                left = next_sc - 1
                while left > 0 and c.sc_phase[left] == 2:
                    left -= 1
                right = next_sc
                while right < n - 1 and c.sc_phase[right] == 2:
                    right += 1
                if left >= 0 and right < n:
                    o.append("%s\t%d\t%d\t%s\t%d\t%d\n" % (c.name, c.sc_end[left], c.sc_beg[right], "SWITCH_ERR", next_sc, pb))
    return "".join(o)


def ng50(ctgs, on_switch, on_flip):
    """phaseblockData::calculate_ng50, phase.cpp:534-626 (a flip on a contig's last supercluster reads begs[n] there:
    undefined; the library and this restatement start the following, empty stretch at the last supercluster's end)"""
    total = sum(int(c.length) for c in ctgs)
    blocks = []
    for c in ctgs:
        n = len(c.sc_beg)
        if n == 0:
            continue
        beg = c.sc_beg[0]
        nxt = lambda k: c.sc_beg[k] if k < n else c.sc_end[n - 1]
        for typ, next_sc, _ in _breaks(c, on_switch, on_flip):
            blocks.append(c.sc_end[next_sc - 1] - beg)
            beg = nxt(next_sc)
            if typ in (SW_FLIP, SW_BOTH):
                blocks.append(c.sc_end[next_sc] - beg)
                beg = nxt(next_sc + 1)
        blocks.append(c.sc_end[n - 1] - beg)
    acc = 0
    for b in sorted(blocks, reverse=True):
        acc += int(b)
        if acc >= total // 2:
            return int(b)
    return 0


def phasing_summary_tsv(ctgs):
    """phaseblockData::write_phasing_summary, phase.cpp:515-528, with the totals of phase.cpp:363-386"""
    nb = sum(len(c.pbs) - 1 for c in ctgs)
    return ("PHASE_BLOCKS\tSWITCH_ERRORS\tFLIP_ERRORS\tNG_50\tSWITCH_NGC50\tSWITCHFLIP_NGC50\n%d\t%d\t%d\t%d\t%d\t%d" %
            (nb, sum(len(c.switches) for c in ctgs), sum(len(c.flips) for c in ctgs), ng50(ctgs, False, False), ng50(ctgs, True, False),
             ng50(ctgs, True, True)))


def superclusters_tsv(ctgs):
    o = ["CONTIG\tSUPERCLUSTER\tSTART\tSTOP\tSIZE\tQUERY1_VARS\tQUERY2_VARS\tTRUTH1_VARS\tTRUTH2_VARS\tORIG_ED\tSWAP_ED"
         "\tPHASE_STATE\tSC_PHASE\tPHASE_SET\tPHASE_BLOCK\tFLIP_ERROR\n"]
    for c in ctgs:
        pb = 0
        for i in range(len(c.sc_beg)):
            if i >= c.pbs[pb + 1]:
                pb += 1
            sw, psc = bool(c.pb_phase[i]), int(c.sc_phase[i])
            flip = (psc == 0) if sw else (psc == 1)
            nv = [(c.clusters[h][c.sc_brk[h][i + 1]] - c.clusters[h][c.sc_brk[h][i]]) if len(c.clusters[h]) else 0 for h in range(4)]
            o.append("%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%s\t%d\t%d\t%d\n" % (
                c.name, i, c.sc_beg[i], c.sc_end[i], c.sc_end[i] - c.sc_beg[i], nv[0], nv[1], nv[2], nv[3], c.orig_dist[i],
                c.swap_dist[i], int(sw), PHASE_STR[psc], c.sc_phase_set[i], pb, int(flip)))
    return "".join(o)


def variants_tsv(ctgs, callset):
    o = ["CONTIG\tPOS\tHAP\tREF\tALT\tQUAL\tTYPE\tERRTYPE\tCREDIT\tCLUSTER\tSUPERCLUSTER\tSYNC_GROUP\tREF_DIST\tQUERY_DIST\tLOCATION\n"]
    for c in ctgs:
        V = [c.slots[2 * callset], c.slots[2 * callset + 1]]
        CL = [c.clusters[2 * callset], c.clusters[2 * callset + 1]]
        R = lambda name, h, s: getattr(c.res, name)[2 * callset + h][s]
        n = [len(V[0]["pos"]), len(V[1]["pos"])]
        v, cl, sci = [0, 0], [0, 0], 0
        while v[0] < n[0] or v[1] < n[1]:
            if v[1] >= n[1] or (v[0] < n[0] and V[0]["pos"][v[0]] < V[1]["pos"][v[1]]):
                h = 0
            else:
                h = 1
            i = v[h]
            assert cl[h] + 1 < len(CL[h])
            if CL[h][cl[h] + 1] <= i:
                cl[h] += 1
            while V[h]["pos"][i] >= c.sc_end[sci]:
                sci += 1
            psc = int(c.sc_phase[sci])
            s = 0 if psc == 0 else 1 if psc == 1 else int(bool(c.pb_phase[sci]))
            o.append("%s\t%d\t%d\t%s\t%s\t%.2f\t%s\t%s\t%f\t%d\t%d\t%d\t%d\t%d\t%s\n" % (
                c.name, V[h]["pos"][i], h, V[h]["ref"][i], V[h]["alt"][i], float(V[h]["var_qual"][i]), TYPE_STR[V[h]["type"][i]],
                ERROR_STR[R("errtype", h, s)[i]], float(R("credit", h, s)[i]), cl[h], sci, R("sync_group", h, s)[i],
                R("ref_ed", h, s)[i], R("query_ed", h, s)[i], REGION_STR[1]))
            v[h] += 1
    return "".join(o)


VCF_HEADER_TAIL = (
    "##FILTER=<ID=PASS,Description=\"All filters passed\">\n"
    "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"GenoType\">\n"
    "##FORMAT=<ID=BD,Number=1,Type=String,Description=\"Benchmark Decision for call (TP/FP/FN).\">\n"
    "##FORMAT=<ID=BC,Number=1,Type=Float,Description=\"Benchmark Credit (on the interval [0,1], based on sync group edit distance)\">\n"
    "##FORMAT=<ID=RD,Number=1,Type=Integer,Description=\"Reference edit Distance from truth within current sync group\">\n"
    "##FORMAT=<ID=QD,Number=1,Type=Integer,Description=\"Query edit Distance from truth within current sync group\">\n"
    "##FORMAT=<ID=BK,Number=1,Type=String,Description=\"BenchmarK category ('gm' if credit == 1, 'lm' if credit > 0, else '.')\">\n"
    "##FORMAT=<ID=QQ,Number=1,Type=Float,Description=\"variant Quality\">\n"
    "##FORMAT=<ID=SC,Number=1,Type=Integer,Description=\"SuperCluster (index in contig)\">\n"
    "##FORMAT=<ID=SG,Number=1,Type=Integer,Description=\"Sync Group (index in supercluster, for credit assignment)\">\n"
    "##FORMAT=<ID=PS,Number=1,Type=Integer,Description=\"Phase Set identifier (input, per-variant)\">\n"
    "##FORMAT=<ID=PB,Number=1,Type=Integer,Description=\"Phase Block (output, per-supercluster, index in contig)\">\n"
    "##FORMAT=<ID=BS,Number=1,Type=Integer,Description=\"Block State (phaseblock truth-to-query mapping state; 0 = T1Q1:T2Q2, 1 = T1Q2:T2Q1)\">\n"
    "##FORMAT=<ID=FE,Number=1,Type=Integer,Description=\"Flip Error (a per-supercluster error)\">\n"
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tTRUTH\tQUERY\n")
FMT = "GT:BD:BC:RD:QD:BK:QQ:SC:SG:PS:PB:BS:FE"


def summary_vcf(ctgs, cmd, file_date, credit_threshold):
    o = ["##fileformat=VCFv4.2\n", "##fileDate=%s\n" % file_date, "##CL=%s\n" % cmd]
    for c in ctgs:
        o.append("##contig=<ID=%s,length=%d,ploidy=%d>\n" % (c.name, c.length, c.ploidy))
    o.append(VCF_HEADER_TAIL)
    thr = f32(credit_threshold)
    for c in ctgs:
        if len(c.sc_beg) == 0:
            continue
        QUERY, TRUTH = 0, 1
        vars_ = [[c.slots[0], c.slots[1]], [c.slots[2], c.slots[3]]]
        size = lambda cs, h: len(vars_[cs][h]["pos"])
        ptrs = [[0, 0], [0, 0]]
        sc_idx, phase_block = 0, 0

        def state(k):
            sw, psc = bool(c.pb_phase[k]), int(c.sc_phase[k])
            if sw:
                return (sw, True, False) if psc == 0 else (sw, False, True)
            return (sw, True, True) if psc == 1 else (sw, False, False)

        phase_switch, phase_flip, swap = state(0)

        def info(cs, h, idx):
            V = vars_[cs][h]
            if V["type"][idx] == 1:
                o.append("%s\t%d\t.\t%s\t%s\t.\tPASS\t.\t%s" % (c.name, V["pos"][idx] + 1, V["ref"][idx], V["alt"][idx], FMT))
            else:
                base = chr(c.seq[V["pos"][idx] - 1])
                o.append("%s\t%d\t.\t%s\t%s\t.\tPASS\t.\t%s" % (c.name, V["pos"][idx], base + V["ref"][idx], base + V["alt"][idx], FMT))

        def empty(query):
            o.append("\t.:.:.:.:.:.:.:%d:.:.:%d:.:.%s" % (sc_idx, phase_block, "\n" if query else ""))

        def sample(cs, h, idx, gt, query):
            V = vars_[cs][h]
            s = int(phase_switch ^ phase_flip)
            slot = 2 * cs + h
            credit = f32(c.res.credit[slot][s][idx])
            if credit == 1:
                bd, bk = "TP", "gm"
            elif credit == 0:
                bd, bk = ("FP" if query else "FN"), "."
            elif credit >= thr:
                bd, bk = "TP", "lm"
            else:
                bd, bk = ("FP" if query else "FN"), "lm"
            red, qed = int(c.res.ref_ed[slot][s][idx]), int(c.res.query_ed[slot][s][idx])
            o.append("\t%s:%s:%f:%s:%s:%s:%d:%d:%d:%d:%d:%s:%s%s" % (
                gt, bd, float(credit), "." if red == 0 else str(red), "." if red == 0 else str(qed), bk, int(V["var_qual"][idx]),
                sc_idx, int(c.res.sync_group[slot][s][idx]), int(V["phase_set"][idx]), phase_block,
                ("1" if phase_switch else "0") if query else ".", ("1" if phase_flip else "0") if query else ".", "\n" if query else ""))

        while any(ptrs[cs][h] < size(cs, h) for cs in range(2) for h in range(2)):
            poss = [[INT_MAX, INT_MAX], [INT_MAX, INT_MAX]]
            for cs in range(2):
                for h in range(2):
                    p = ptrs[cs][h]
                    if p < size(cs, h):
                        poss[cs][h] = int(vars_[cs][h]["pos"][p]) - (1 if vars_[cs][h]["type"][p] in (2, 3) else 0)
            pos = min(poss[0] + poss[1])
            nxt = [[poss[cs][h] == pos for h in range(2)] for cs in range(2)]
            if pos >= c.sc_end[sc_idx]:
                sc_idx += 1
                if sc_idx >= c.pbs[phase_block + 1]:
                    phase_block += 1
                phase_switch, phase_flip, swap = state(sc_idx)
            sw = int(swap)
            pair = [False, False]
            for h in range(2):
                if nxt[QUERY][h] and nxt[TRUTH][sw ^ h]:
                    a, b = vars_[QUERY][h], vars_[TRUTH][sw ^ h]
                    ia, ib = ptrs[QUERY][h], ptrs[TRUTH][sw ^ h]
                    pair[h] = a["ref"][ia] == b["ref"][ib] and a["alt"][ia] == b["alt"][ib]
            hap_gt = lambda h: "0|1" if h else "1|0"
            one_gt = lambda h: "1" if c.ploidy == 1 else hap_gt(h)
            if nxt[QUERY][0] and nxt[QUERY][1]:
                for h in range(2):
                    info(QUERY, h, ptrs[QUERY][h])
                    if pair[h]:
                        sample(TRUTH, h ^ sw, ptrs[TRUTH][h ^ sw], hap_gt(h ^ sw), False)
                        sample(QUERY, h, ptrs[QUERY][h], hap_gt(h), True)
                        ptrs[QUERY][h] += 1
                        ptrs[TRUTH][h ^ sw] += 1
                    else:
                        empty(False)
                        sample(QUERY, h, ptrs[QUERY][h], hap_gt(h), True)
                        ptrs[QUERY][h] += 1
            elif nxt[QUERY][0] or nxt[QUERY][1]:
                h = 0 if nxt[QUERY][0] else 1
                info(QUERY, h, ptrs[QUERY][h])
                if pair[h]:
                    sample(TRUTH, h ^ sw, ptrs[TRUTH][h ^ sw], one_gt(h ^ sw), False)
                    sample(QUERY, h, ptrs[QUERY][h], one_gt(h), True)
                    ptrs[QUERY][h] += 1
                    ptrs[TRUTH][h ^ sw] += 1
                else:
                    empty(False)
                    sample(QUERY, h, ptrs[QUERY][h], one_gt(h), True)
                    ptrs[QUERY][h] += 1
            elif nxt[TRUTH][0] and nxt[TRUTH][1]:
                for h in range(2):
                    info(TRUTH, h ^ sw, ptrs[TRUTH][h ^ sw])
                    sample(TRUTH, h ^ sw, ptrs[TRUTH][h ^ sw], hap_gt(h ^ sw), False)
                    empty(True)
                    ptrs[TRUTH][h ^ sw] += 1
            else:
                th = (0 ^ sw) if nxt[TRUTH][0 ^ sw] else (1 ^ sw)
                assert nxt[TRUTH][th]
                info(TRUTH, th, ptrs[TRUTH][th])
                sample(TRUTH, th, ptrs[TRUTH][th], one_gt(th), False)
                empty(True)
                ptrs[TRUTH][th] += 1
    return "".join(o)
