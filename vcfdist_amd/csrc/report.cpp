// report.cpp -- the output tables behind the precision/recall path (include/vcfdist_report.h): host formatting of the
// device results into the reference's files.  Every format string follows the reference's fprintf at the cited line
// so that the files compare byte for byte; the control flow is organised around one merge iterator per file instead
// of the reference's per-haplotype copies.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/vcfdist_pr.h"
#include "../../include/vcfdist_report.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }

// string tables of src/main.cpp:14-24
const char *const TYPE_STR[] = {"REF", "SNP", "INS", "DEL", "CPX"};
const char *const VARTYPE_STR[] = {"SNP", "INDEL", "SV", "ALL"};
const char *const ERROR_STR[] = {"TP", "FP", "FN", "PE", "GE", "??"};
const char *const REGION_STR[] = {"OUTSIDE", "INSIDE ", "BORDER ", "OFF CTG"};
const char *const PHASE_STR[] = {"=", "X", "?"};

struct File {
    FILE *f = nullptr;
    explicit File(const char *path) { if (path) f = fopen(path, "w"); }
    ~File() { if (f) fclose(f); }
    operator FILE *() const { return f; }
    // flush and close now; false if any write failed (disk full, ...): the caller reports it instead of VRP_OK
    bool finish() {
        if (!f) return false;
        const bool bad = fflush(f) != 0 || ferror(f) != 0;
        const bool closed = fclose(f) == 0;
        f = nullptr;
        return !bad && closed;
    }
};

struct Metrics { int query_tp, query_fp, truth_tp, truth_fn; float precision, recall, f1, f1_q; };

Metrics metrics(const int64_t *counts, int nq, int type, int qidx) {
    auto at = [&](int callset, int err) { return int(counts[((size_t(callset) * VPR_VARTYPES + type) * 3 + err) * nq + qidx]); };
    Metrics m;
    m.query_tp = at(0, VPR_ERRTYPE_TP); m.query_fp = at(0, VPR_ERRTYPE_FP);
    m.truth_tp = at(1, VPR_ERRTYPE_TP); m.truth_fn = at(1, VPR_ERRTYPE_FN);
    const int query_tot = m.query_tp + m.query_fp, truth_tot = m.truth_tp + m.truth_fn;
    m.precision = query_tot == 0 ? 1 : float(m.query_tp) / query_tot;
    m.recall = truth_tot == 0 ? 1 : float(m.truth_tp) / truth_tot;
    m.f1 = m.precision + m.recall > 0 ? 2 * m.precision * m.recall / (m.precision + m.recall) : 0;
    m.f1_q = float(std::min(100.0, std::max(0.0, -10 * std::log10(double(1 - m.f1)))));   // qscore(), edit.cpp:102
    return m;
}

// the evaluation (ORIG / SWAP) a supercluster's variants are reported under: print.cpp:699-704
inline int swap_of(const vrp_contig &c, int sci) {
    const int p = c.sc_phase[sci];
    return p == VPR_PHASE_ORIG ? 0 : p == VPR_PHASE_SWAP ? 1 : (c.pb_phase[sci] != 0);
}

bool contig_ok(const vrp_contig &c) {
    if (!c.name) return false;
    if (c.n_sc > 0 && (!c.sc_beg || !c.sc_end || !c.sc_phase || !c.pb_phase)) return false;
    for (int i = 0; i < 4; i++) {
        const vrp_hap &h = c.hap[i];
        if (h.n_var > 0 && (!h.pos || !h.type || !h.cluster_beg || !h.pool || !h.ref_len || !h.alt_len || !h.ref_off || !h.alt_off))
            return false;
    }
    return true;
}

}  // namespace

extern "C" const char *vrp_last_error(void) { return g_err.c_str(); }

extern "C" int32_t vrp_phase_blocks(const int32_t *sc_phase_set, int32_t n_sc, int32_t *phase_block) {
    if (n_sc < 0 || !phase_block || (n_sc && !sc_phase_set)) return VRP_ERR_ARG;
    int32_t n_pb = 0, cur = -1;   // a new block wherever the phase set changes (phase.cpp:249-258)
    for (int32_t k = 0; k < n_sc; k++)
        if (sc_phase_set[k] != cur) { phase_block[n_pb++] = k; cur = sc_phase_set[k]; }
    phase_block[n_pb] = n_sc;
    return n_pb;
}

extern "C" int vrp_write_precision_recall(const char *prefix, const int64_t *counts, int32_t min_qual, int32_t max_qual) {
    if (!prefix || !counts || max_qual < min_qual) return fail(VRP_ERR_ARG, "vrp_write_precision_recall: bad argument");
    const int nq = max_qual - min_qual + 1;
    const std::string fn_all = std::string(prefix) + "precision-recall.tsv", fn_sum = std::string(prefix) + "precision-recall-summary.tsv";
    File all(fn_all.c_str());
    if (!all) return fail(VRP_ERR_OPEN, "cannot create " + fn_all);
    File sum(fn_sum.c_str());
    if (!sum) return fail(VRP_ERR_OPEN, "cannot create " + fn_sum);
    fprintf(all, "VAR_TYPE\tMIN_QUAL\tPREC\tRECALL\tF1_SCORE\tF1_QSCORE\tTRUTH_TOTAL\tTRUTH_TP\tTRUTH_FN\tQUERY_TOTAL\tQUERY_TP\tQUERY_FP\n");
    int best_qual[VPR_VARTYPES];
    for (int type = 0; type < VPR_VARTYPES; type++) {
        float best = 0;
        best_qual[type] = 0;
        for (int qual = min_qual; qual <= max_qual; qual++) {
            const Metrics m = metrics(counts, nq, type, qual - min_qual);
            if (m.f1 > best) { best = m.f1; best_qual[type] = qual; }   // first maximum wins (print.cpp:470)
            fprintf(all, "%s\t%d\t%f\t%f\t%f\t%f\t%d\t%d\t%d\t%d\t%d\t%d\n", VARTYPE_STR[type], qual, m.precision, m.recall,
                    m.f1, m.f1_q, m.truth_tp + m.truth_fn, m.truth_tp, m.truth_fn, m.query_tp + m.query_fp, m.query_tp, m.query_fp);
        }
        // (an all-zero F1 column leaves quality 0 in the reference, which it then indexes; keep the index in range)
        if (best_qual[type] < min_qual || best_qual[type] > max_qual) best_qual[type] = min_qual;
    }
    fprintf(sum, "VAR_TYPE\tTHRESHOLD\tMIN_QUAL\tTRUTH_TP\tQUERY_TP\tTRUTH_FN\tQUERY_FP\tPREC\tRECALL\tF1_SCORE\tF1_QSCORE\n");
    for (int type = 0; type < VPR_VARTYPES; type++) {
        const int quals[2] = {min_qual, best_qual[type]};
        const char *const thresh[2] = {"NONE", "BEST"};
        for (int i = 0; i < 2; i++) {
            const Metrics m = metrics(counts, nq, type, quals[i] - min_qual);
            fprintf(sum, "%s\t%s\t%d\t%d\t%d\t%d\t%d\t%f\t%f\t%f\t%f\n", VARTYPE_STR[type], thresh[i], quals[i], m.truth_tp,
                    m.query_tp, m.truth_fn, m.query_fp, m.precision, m.recall, m.f1, m.f1_q);
        }
    }
    if (!all.finish() || !sum.finish()) return fail(VRP_ERR_OPEN, "write error on " + fn_all + " / " + fn_sum);
    return VRP_OK;
}

extern "C" int vrp_write_phase_blocks(const char *path, const vrp_contig *ctgs, int32_t n_ctg) {
    if (!path || n_ctg < 0 || (n_ctg && !ctgs)) return fail(VRP_ERR_ARG, "vrp_write_phase_blocks: bad argument");
    File out(path);
    if (!out) return fail(VRP_ERR_OPEN, std::string("cannot create ") + path);
    fprintf(out, "CONTIG\tPHASE_BLOCK\tSTART\tSTOP\tSIZE\tSUPERCLUSTERS\tFLIP_ERRORS\tSWITCH_ERRORS\n");
    for (int32_t ci = 0; ci < n_ctg; ci++) {
        const vrp_contig &c = ctgs[ci];
        if (!contig_ok(c) || (c.n_pb > 0 && !c.phase_block)) return fail(VRP_ERR_ARG, "vrp_write_phase_blocks: incomplete contig");
        for (int32_t i = 0; i < c.n_pb && c.n_sc > 0; i++) {
            const int32_t first = c.phase_block[i], last = c.phase_block[i + 1] - 1;
            // errors strictly inside the block: an event on its first supercluster belongs to the border (print.cpp:596-603)
            auto inside = [&](const int32_t *ev, int32_t n) {
                int32_t k = 0;
                for (int32_t e = 0; e < n; e++) k += ev[e] > first && ev[e] <= last;
                return k;
            };
            const int32_t beg = c.sc_beg[first], end = c.sc_end[last];
            fprintf(out, "%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", c.name, i, beg, end, end - beg, last - first + 1,
                    inside(c.flips, c.n_flips), inside(c.switches, c.n_switches));
        }
    }
    if (!out.finish()) return fail(VRP_ERR_OPEN, std::string("write error on ") + path);
    return VRP_OK;
}

// ---- phasing errors: switchflips.tsv (phaseblockData::write_switchflips, phase.cpp:406-509), phasing-summary.tsv
// (write_phasing_summary, phase.cpp:515-528) and the three NG50 figures behind it (calculate_ng50, phase.cpp:534-626).
// All of them sweep a contig's superclusters from break to break.  A break is the start of a phase block (a new phase
// set: not an error), a switch error or a flip error; the next one is the pending event with the smallest supercluster,
// where on equal superclusters a flip takes precedence over a switch error over a block start -- and a flip that is met
// while a block start is the candidate is reported as "switch + flip" and uses up that block start, wherever the block
// start lies (the reference's comparison order; the tables must come out the same).
namespace {
enum { BRK_FLIP = 0, BRK_BLOCK = 1, BRK_BLOCK_FLIP = 2, BRK_SWITCH_ERR = 3, BRK_NONE = 6 };

// f(kind, supercluster, index of the phase block the break is counted in); false = events out of order
template <class F>
bool for_each_break(const vrp_contig &c, bool with_switches, bool with_flips, F f) {
    int32_t i_pb = 1, i_sw = 0, i_fl = 0, at = 0;
    for (;;) {
        int kind = BRK_NONE;
        int32_t nxt = c.n_sc;
        if (i_pb < c.n_pb && c.phase_block[i_pb] <= nxt) { kind = BRK_BLOCK; nxt = c.phase_block[i_pb]; }
        if (with_switches && i_sw < c.n_switches && c.switches[i_sw] <= nxt) { kind = BRK_SWITCH_ERR; nxt = c.switches[i_sw]; }
        if (with_flips && i_fl < c.n_flips && c.flips[i_fl] <= nxt) { kind = kind == BRK_BLOCK ? BRK_BLOCK_FLIP : BRK_FLIP; nxt = c.flips[i_fl]; }
        if (kind == BRK_NONE) return true;
        if (nxt <= at) return false;        // the reference's ERROR("Next supercluster ... is not after current supercluster")
        f(kind, nxt, i_pb - 1);
        if (kind == BRK_FLIP || kind == BRK_BLOCK_FLIP) i_fl++;
        if (kind == BRK_BLOCK || kind == BRK_BLOCK_FLIP) i_pb++;
        if (kind == BRK_SWITCH_ERR) i_sw++;
        at = nxt;
    }
}

bool phasing_ok(const vrp_contig &c) {
    if (!contig_ok(c) || c.n_pb < 0 || c.n_switches < 0 || c.n_flips < 0) return false;
    if (c.n_sc > 0 && !c.phase_block) return false;
    return (c.n_switches == 0 || c.switches) && (c.n_flips == 0 || c.flips);
}
}  // namespace

extern "C" int vrp_write_switchflips(const char *path, const vrp_contig *ctgs, int32_t n_ctg) {
    if (!path || n_ctg < 0 || (n_ctg && !ctgs)) return fail(VRP_ERR_ARG, "vrp_write_switchflips: bad argument");
    File out(path);
    if (!out) return fail(VRP_ERR_OPEN, std::string("cannot create ") + path);
    fprintf(out, "CONTIG\tSTART\tSTOP\tSWITCH_TYPE\tSUPERCLUSTER\tPHASE_BLOCK\n");
    for (int32_t ci = 0; ci < n_ctg; ci++) {
        const vrp_contig &c = ctgs[ci];
        if (!phasing_ok(c)) return fail(VRP_ERR_ARG, "vrp_write_switchflips: incomplete contig");
        if (c.n_sc == 0) continue;
        // nearest supercluster with a phase of its own at or left of `k` (stops at 0) / at or right of `k` (stops at the last)
        auto phased_left = [&](int32_t k) { while (k > 0 && c.sc_phase[k] == VPR_PHASE_NONE) k--; return k; };
        auto phased_right = [&](int32_t k) { while (k < c.n_sc - 1 && c.sc_phase[k] == VPR_PHASE_NONE) k++; return k; };
        auto row = [&](int32_t from_sc, int32_t to_sc, const char *what, int32_t sc, int32_t pb) {
            fprintf(out, "%s\t%d\t%d\t%s\t%d\t%d\n", c.name, c.sc_end[from_sc], c.sc_beg[to_sc], what, sc, pb);
        };
        const bool ordered = for_each_break(c, true, true, [&](int kind, int32_t sc, int32_t pb) {
            if (kind == BRK_FLIP || kind == BRK_BLOCK_FLIP) {
                // the phasing may have changed anywhere between the last phased supercluster in front of the flipped one and
                // the flipped one, and back anywhere between it and the next phased one
                const int32_t l = phased_left(sc - 1), r = phased_right(sc + 1);
                if (l >= 0) row(l, sc, "FLIP_BEG", sc, pb);
                if (r < c.n_sc) row(sc, r, "FLIP_END", sc, pb);
            } else if (kind == BRK_SWITCH_ERR) {
                const int32_t l = phased_left(sc - 1), r = phased_right(sc);
                if (l >= 0 && r < c.n_sc) row(l, r, "SWITCH_ERR", sc, pb);
            }       // (the start of a phase block is not an error: no row)
        });
        if (!ordered) return fail(VRP_ERR_ARG, "vrp_write_switchflips: phase blocks / switches / flips are not ascending");
    }
    if (!out.finish()) return fail(VRP_ERR_OPEN, std::string("write error on ") + path);
    return VRP_OK;
}

// NG50 of the correctly phased stretches: break at phase-block starts, optionally at switch errors and around flipped
// superclusters; the genome size is the sum of the contigs' header lengths.  Returns -1 for inconsistent tables.
extern "C" int32_t vrp_ng50(const vrp_contig *ctgs, int32_t n_ctg, int32_t break_on_switch, int32_t break_on_flip) {
    if (n_ctg < 0 || (n_ctg && !ctgs)) return -1;
    uint64_t genome = 0;
    std::vector<int32_t> stretch;
    for (int32_t ci = 0; ci < n_ctg; ci++) {
        const vrp_contig &c = ctgs[ci];
        if (!phasing_ok(c)) return -1;
        genome += uint64_t(uint32_t(c.length));
        if (c.n_sc == 0) continue;
        int32_t from = c.sc_beg[0];
        auto cut = [&](int32_t last_sc, int32_t next_sc) {      // a stretch ends with last_sc, the next one starts at next_sc
            stretch.push_back(c.sc_end[last_sc] - from);
            // (a flip on the contig's last supercluster makes the reference read begs[n]: the stretch behind it is empty here)
            from = next_sc < c.n_sc ? c.sc_beg[next_sc] : c.sc_end[c.n_sc - 1];
        };
        const bool ordered = for_each_break(c, break_on_switch != 0, break_on_flip != 0, [&](int kind, int32_t sc, int32_t) {
            cut(sc - 1, sc);
            if (kind == BRK_FLIP || kind == BRK_BLOCK_FLIP) cut(sc, sc + 1);     // the flipped supercluster is a stretch of its own
        });
        if (!ordered) return -1;
        stretch.push_back(c.sc_end[c.n_sc - 1] - from);
    }
    std::sort(stretch.begin(), stretch.end(), std::greater<int32_t>());
    uint64_t covered = 0;
    for (int32_t len : stretch) {
        covered += uint64_t(int64_t(len));      // (size_t arithmetic, as in the reference)
        if (covered >= genome / 2) return len;
    }
    return 0;
}

extern "C" int vrp_write_phasing_summary(const char *path, const vrp_contig *ctgs, int32_t n_ctg) {
    if (!path || n_ctg < 0 || (n_ctg && !ctgs)) return fail(VRP_ERR_ARG, "vrp_write_phasing_summary: bad argument");
    int blocks = 0, switches = 0, flips = 0;
    for (int32_t ci = 0; ci < n_ctg; ci++) {
        if (!phasing_ok(ctgs[ci])) return fail(VRP_ERR_ARG, "vrp_write_phasing_summary: incomplete contig");
        blocks += ctgs[ci].n_pb; switches += ctgs[ci].n_switches; flips += ctgs[ci].n_flips;
    }
    const int32_t ng50 = vrp_ng50(ctgs, n_ctg, 0, 0), sw50 = vrp_ng50(ctgs, n_ctg, 1, 0), sf50 = vrp_ng50(ctgs, n_ctg, 1, 1);
    if (ng50 < 0 || sw50 < 0 || sf50 < 0) return fail(VRP_ERR_ARG, "vrp_write_phasing_summary: phase blocks / switches / flips are not ascending");
    File out(path);
    if (!out) return fail(VRP_ERR_OPEN, std::string("cannot create ") + path);
    fprintf(out, "PHASE_BLOCKS\tSWITCH_ERRORS\tFLIP_ERRORS\tNG_50\tSWITCH_NGC50\tSWITCHFLIP_NGC50\n");
    fprintf(out, "%d\t%d\t%d\t%d\t%d\t%d", blocks, switches, flips, ng50, sw50, sf50);      // (no newline behind the row, phase.cpp:526)
    if (!out.finish()) return fail(VRP_ERR_OPEN, std::string("write error on ") + path);
    return VRP_OK;
}

extern "C" int vrp_write_superclusters(const char *path, const vrp_contig *ctgs, int32_t n_ctg) {
    if (!path || n_ctg < 0 || (n_ctg && !ctgs)) return fail(VRP_ERR_ARG, "vrp_write_superclusters: bad argument");
    File out(path);
    if (!out) return fail(VRP_ERR_OPEN, std::string("cannot create ") + path);
    fprintf(out, "CONTIG\tSUPERCLUSTER\tSTART\tSTOP\tSIZE\tQUERY1_VARS\tQUERY2_VARS\tTRUTH1_VARS\tTRUTH2_VARS\tORIG_ED\tSWAP_ED"
                 "\tPHASE_STATE\tSC_PHASE\tPHASE_SET\tPHASE_BLOCK\tFLIP_ERROR\n");
    for (int32_t ci = 0; ci < n_ctg; ci++) {
        const vrp_contig &c = ctgs[ci];
        if (!contig_ok(c)) return fail(VRP_ERR_ARG, "vrp_write_superclusters: incomplete contig");
        if (c.n_sc > 0 && (!c.phase_block || !c.orig_phase_dist || !c.swap_phase_dist || !c.sc_phase_set))
            return fail(VRP_ERR_ARG, "vrp_write_superclusters: incomplete contig");
        int32_t pb = 0;
        for (int32_t k = 0; k < c.n_sc; k++) {
            if (k >= c.phase_block[pb + 1]) pb++;
            const int phase_switch = c.pb_phase[k] != 0, phase_sc = c.sc_phase[k];
            if (phase_sc < 0 || phase_sc > VPR_PHASE_NONE) return fail(VRP_ERR_ARG, "vrp_write_superclusters: unexpected phase");
            const int flip_error = phase_switch ? phase_sc == VPR_PHASE_ORIG : phase_sc == VPR_PHASE_SWAP;
            int32_t nv[4];
            for (int i = 0; i < 4; i++) {   // an empty hap has no cluster table at all
                const vrp_hap &h = c.hap[i];
                nv[i] = (h.n_var > 0 && c.sc_brk[i]) ? h.cluster_beg[c.sc_brk[i][k + 1]] - h.cluster_beg[c.sc_brk[i][k]] : 0;
            }
            fprintf(out, "%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%s\t%d\t%d\t%d\n", c.name, k, c.sc_beg[k], c.sc_end[k],
                    c.sc_end[k] - c.sc_beg[k], nv[0], nv[1], nv[2], nv[3], c.orig_phase_dist[k], c.swap_phase_dist[k], phase_switch,
                    PHASE_STR[phase_sc], c.sc_phase_set[k], pb, flip_error);
        }
    }
    if (!out.finish()) return fail(VRP_ERR_OPEN, std::string("write error on ") + path);
    return VRP_OK;
}

extern "C" int vrp_write_variants(const char *path, const vrp_contig *ctgs, int32_t n_ctg, int32_t callset) {
    if (!path || n_ctg < 0 || (n_ctg && !ctgs) || callset < 0 || callset > 1) return fail(VRP_ERR_ARG, "vrp_write_variants: bad argument");
    File out(path);
    if (!out) return fail(VRP_ERR_OPEN, std::string("cannot create ") + path);
    fprintf(out, "CONTIG\tPOS\tHAP\tREF\tALT\tQUAL\tTYPE\tERRTYPE\tCREDIT\tCLUSTER\tSUPERCLUSTER\tSYNC_GROUP\tREF_DIST\tQUERY_DIST\tLOCATION\n");
    for (int32_t ci = 0; ci < n_ctg; ci++) {
        const vrp_contig &c = ctgs[ci];
        if (!contig_ok(c)) return fail(VRP_ERR_ARG, "vrp_write_variants: incomplete contig");
        const vrp_hap *H[2] = {&c.hap[2 * callset], &c.hap[2 * callset + 1]};
        int32_t v[2] = {0, 0}, cl[2] = {0, 0}, sci = 0;
        while (v[0] < H[0]->n_var || v[1] < H[1]->n_var) {
            // the two haplotypes merge by position; on equal positions the second haplotype goes first (print.cpp:692-693)
            const int h = (v[1] >= H[1]->n_var || (v[0] < H[0]->n_var && H[0]->pos[v[0]] < H[1]->pos[v[1]])) ? 0 : 1;
            const vrp_hap &V = *H[h];
            const int32_t i = v[h];
            if (cl[h] + 1 > V.n_cluster) return fail(VRP_ERR_ARG, "Out of bounds cluster during write_results()");
            if (V.cluster_beg[cl[h] + 1] <= i) cl[h]++;
            while (sci < c.n_sc && V.pos[i] >= c.sc_end[sci]) sci++;
            if (sci >= c.n_sc) return fail(VRP_ERR_ARG, "Out of bounds supercluster during write_results()");
            const int s = swap_of(c, sci);
            if (!V.errtype[s] || !V.credit[s] || !V.sync_group[s] || !V.ref_ed[s] || !V.query_ed[s] || !V.var_qual)
                return fail(VRP_ERR_ARG, "vrp_write_variants: result columns missing");
            const int et = V.errtype[s][i], ty = V.type[i], loc = V.loc ? V.loc[i] : 1;
            if (et > VPR_ERRTYPE_UN || ty > 4 || loc > 3) return fail(VRP_ERR_ARG, "vrp_write_variants: value out of range");
            fprintf(out, "%s\t%d\t%d\t%.*s\t%.*s\t%.2f\t%s\t%s\t%f\t%d\t%d\t%d\t%d\t%d\t%s\n", c.name, V.pos[i], h,
                    int(V.ref_len[i]), reinterpret_cast<const char *>(V.pool + V.ref_off[i]),
                    int(V.alt_len[i]), reinterpret_cast<const char *>(V.pool + V.alt_off[i]),
                    V.var_qual[i], TYPE_STR[ty], ERROR_STR[et], V.credit[s][i], cl[h], sci, V.sync_group[s][i],
                    V.ref_ed[s][i], V.query_ed[s][i], REGION_STR[loc]);
            v[h]++;
        }
    }
    if (!out.finish()) return fail(VRP_ERR_OPEN, std::string("write error on ") + path);
    return VRP_OK;
}

namespace {

// GA4GH-style record pieces (variant.cpp:229-286)
struct VcfOut {
    FILE *f;
    const vrp_contig *c;
    float credit_threshold;

    bool info(const vrp_hap &V, int32_t i) const {
        const char *ref = reinterpret_cast<const char *>(V.pool + V.ref_off[i]);
        const char *alt = reinterpret_cast<const char *>(V.pool + V.alt_off[i]);
        static const char FMT[] = "GT:BD:BC:RD:QD:BK:QQ:SC:SG:PS:PB:BS:FE";
        if (V.type[i] == VPR_TYPE_SUB) {
            fprintf(f, "%s\t%d\t.\t%.*s\t%.*s\t.\tPASS\t.\t%s", c->name, V.pos[i] + 1, int(V.ref_len[i]), ref, int(V.alt_len[i]), alt, FMT);
        } else if (V.type[i] == VPR_TYPE_INS || V.type[i] == VPR_TYPE_DEL) {
            if (!c->seq || V.pos[i] < 1 || V.pos[i] > c->seq_len) return false;
            const char anchor = char(c->seq[V.pos[i] - 1]);   // indels are written with the base in front of them
            fprintf(f, "%s\t%d\t.\t%c%.*s\t%c%.*s\t.\tPASS\t.\t%s", c->name, V.pos[i], anchor, int(V.ref_len[i]), ref, anchor,
                    int(V.alt_len[i]), alt, FMT);
        } else {
            return false;
        }
        return true;
    }
    void empty(int sci, int pb, bool query) const { fprintf(f, "\t.:.:.:.:.:.:.:%d:.:.:%d:.:.%s", sci, pb, query ? "\n" : ""); }
    void sample(const vrp_hap &V, int32_t i, const char *gt, int sci, int pb, bool phase_switch, bool phase_flip, bool query) const {
        const int s = phase_switch ^ phase_flip;
        const float credit = V.credit[s][i];
        const char *bd, *bk;
        if (credit == 1) { bd = "TP"; bk = "gm"; }
        else if (credit == 0) { bd = query ? "FP" : "FN"; bk = "."; }
        else if (credit >= credit_threshold) { bd = "TP"; bk = "lm"; }
        else { bd = query ? "FP" : "FN"; bk = "lm"; }
        // RD and QD are both blanked by a zero reference distance (variant.cpp:279-280)
        const bool dot = V.ref_ed[s][i] == 0;
        const std::string rd = dot ? "." : std::to_string(V.ref_ed[s][i]), qd = dot ? "." : std::to_string(V.query_ed[s][i]);
        fprintf(f, "\t%s:%s:%f:%s:%s:%s:%d:%d:%d:%d:%d:%s:%s%s", gt, bd, credit, rd.c_str(), qd.c_str(), bk, int(V.var_qual[i]), sci,
                V.sync_group[s][i], V.phase_set ? V.phase_set[i] : 0, pb, query ? (phase_switch ? "1" : "0") : ".",
                query ? (phase_flip ? "1" : "0") : ".", query ? "\n" : "");
    }
};

inline bool same_allele(const vrp_hap &A, int32_t a, const vrp_hap &B, int32_t b) {
    return A.ref_len[a] == B.ref_len[b] && A.alt_len[a] == B.alt_len[b] &&
           std::equal(A.pool + A.ref_off[a], A.pool + A.ref_off[a] + A.ref_len[a], B.pool + B.ref_off[b]) &&
           std::equal(A.pool + A.alt_off[a], A.pool + A.alt_off[a] + A.alt_len[a], B.pool + B.alt_off[b]);
}

}  // namespace

extern "C" int vrp_write_summary_vcf(const char *path, const vrp_contig *ctgs, int32_t n_ctg, const char *cmd,
                                     const char *file_date, float credit_threshold) {
    if (!path || n_ctg < 0 || (n_ctg && !ctgs)) return fail(VRP_ERR_ARG, "vrp_write_summary_vcf: bad argument");
    File out(path);
    if (!out) return fail(VRP_ERR_OPEN, std::string("cannot create ") + path);
    fprintf(out, "##fileformat=VCFv4.2\n");
    if (file_date) {
        fprintf(out, "##fileDate=%s\n", file_date);
    } else {
        const time_t tt = time(nullptr);
        const tm lt = *localtime(&tt);
        fprintf(out, "##fileDate=%04d%02d%02d\n", lt.tm_year + 1900, lt.tm_mon + 1, lt.tm_mday);
    }
    fprintf(out, "##CL=%s\n", cmd ? cmd : "");
    for (int32_t ci = 0; ci < n_ctg; ci++)
        fprintf(out, "##contig=<ID=%s,length=%d,ploidy=%d>\n", ctgs[ci].name, ctgs[ci].length, ctgs[ci].ploidy);
    // header text of phase.cpp:24-39
    fputs("##FILTER=<ID=PASS,Description=\"All filters passed\">\n"
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
          "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tTRUTH\tQUERY\n", out);

    for (int32_t ci = 0; ci < n_ctg; ci++) {
        const vrp_contig &c = ctgs[ci];
        if (!contig_ok(c)) return fail(VRP_ERR_ARG, "vrp_write_summary_vcf: incomplete contig");
        if (c.n_sc == 0) continue;
        if (!c.phase_block) return fail(VRP_ERR_ARG, "vrp_write_summary_vcf: incomplete contig");
        const VcfOut w{out, &c, credit_threshold};
        const vrp_hap *Q[2] = {&c.hap[0], &c.hap[1]}, *T[2] = {&c.hap[2], &c.hap[3]};
        for (int i = 0; i < 4; i++)
            if (c.hap[i].n_var > 0 && (!c.hap[i].credit[0] || !c.hap[i].credit[1] || !c.hap[i].ref_ed[0] || !c.hap[i].ref_ed[1] ||
                                       !c.hap[i].query_ed[0] || !c.hap[i].query_ed[1] || !c.hap[i].sync_group[0] ||
                                       !c.hap[i].sync_group[1] || !c.hap[i].var_qual))
                return fail(VRP_ERR_ARG, "vrp_write_summary_vcf: result columns missing");
        int32_t ptr[4] = {0, 0, 0, 0};
        int32_t sci = 0, pb = 0;
        bool phase_switch = false, phase_flip = false;
        int swap = 0;
        // a supercluster whose own phase contradicts its block's state is a flip error and keeps its own evaluation
        auto enter = [&](int32_t k) {
            phase_switch = c.pb_phase[k] != 0;
            const int psc = c.sc_phase[k];
            phase_flip = phase_switch ? psc == VPR_PHASE_ORIG : psc == VPR_PHASE_SWAP;
            swap = int(phase_switch) ^ int(phase_flip);
        };
        enter(0);
        auto left = [&]() { for (int i = 0; i < 4; i++) if (ptr[i] < c.hap[i].n_var) return true; return false; };
        const bool haploid = c.ploidy == 1;
        while (left()) {
            // next record position: indels are keyed by their anchor base
            int32_t at[4];
            for (int i = 0; i < 4; i++) {
                const vrp_hap &V = c.hap[i];
                at[i] = INT_MAX;
                if (ptr[i] < V.n_var) at[i] = V.pos[ptr[i]] - (V.type[ptr[i]] == VPR_TYPE_INS || V.type[ptr[i]] == VPR_TYPE_DEL);
            }
            const int32_t pos = std::min(std::min(at[0], at[1]), std::min(at[2], at[3]));
            bool nq[2] = {at[0] == pos, at[1] == pos}, nt[2] = {at[2] == pos, at[3] == pos};
            if (pos >= c.sc_end[sci]) {   // one step at most, as in the reference (phase.cpp:104)
                if (++sci >= c.n_sc) return fail(VRP_ERR_ARG, "vrp_write_summary_vcf: variant behind the last supercluster");
                if (sci >= c.phase_block[pb + 1]) pb++;
                enter(sci);
            }
            bool pair[2];
            for (int h = 0; h < 2; h++)
                pair[h] = nq[h] && nt[swap ^ h] && same_allele(*Q[h], ptr[h], *T[swap ^ h], ptr[2 + (swap ^ h)]);
            auto gt_of = [&](int hap) { return haploid ? "1" : hap ? "0|1" : "1|0"; };
            auto emit_query = [&](int h, const char *gq, const char *gtr) {   // one record led by query hap h
                const int th = h ^ swap;
                if (!w.info(*Q[h], ptr[h])) return false;
                if (pair[h]) { w.sample(*T[th], ptr[2 + th], gtr, sci, pb, phase_switch, phase_flip, false); }
                else w.empty(sci, pb, false);
                w.sample(*Q[h], ptr[h], gq, sci, pb, phase_switch, phase_flip, true);
                ptr[h]++;
                if (pair[h]) ptr[2 + th]++;
                return true;
            };
            auto emit_truth = [&](int h, const char *gtr) {   // truth-only record on the truth hap paired with query hap h
                const int th = h ^ swap;
                if (!w.info(*T[th], ptr[2 + th])) return false;
                w.sample(*T[th], ptr[2 + th], gtr, sci, pb, phase_switch, phase_flip, false);
                w.empty(sci, pb, true);
                ptr[2 + th]++;
                return true;
            };
            bool ok = true;
            if (nq[0] && nq[1]) {          // homozygous-looking pairs stay two records (their credit may differ)
                for (int h = 0; h < 2 && ok; h++) ok = emit_query(h, h ? "0|1" : "1|0", (h ^ swap) ? "0|1" : "1|0");
            } else if (nq[0]) {
                ok = emit_query(0, gt_of(0), gt_of(0 ^ swap));
            } else if (nq[1]) {
                ok = emit_query(1, gt_of(1), gt_of(1 ^ swap));
            } else if (nt[0] && nt[1]) {
                for (int h = 0; h < 2 && ok; h++) ok = emit_truth(h, (h ^ swap) ? "0|1" : "1|0");
            } else if (nt[0 ^ swap]) {
                ok = emit_truth(0, gt_of(0 ^ swap));
            } else if (nt[1 ^ swap]) {
                ok = emit_truth(1, gt_of(1 ^ swap));
            } else {
                return fail(VRP_ERR_ARG, "No variants are selected next.");
            }
            if (!ok) return fail(VRP_ERR_ARG, "vrp_write_summary_vcf: variant type / anchor base unavailable");
        }
    }
    if (!out.finish()) return fail(VRP_ERR_OPEN, std::string("write error on ") + path);
    return VRP_OK;
}
