// summary.cpp -- host side of the step right behind the path: per-contig phasing of the superclusters
// (phaseblockData::phase, src/phase.cpp:271-355) and the PRECISION-RECALL SUMMARY arithmetic
// (write_precision_recall, src/print.cpp:444-566).  The counting itself runs on the device (vpr_pr_counts, pr_api.hip).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/vcfdist_pr.h"

extern "C" void vpr_var_class(const uint8_t *var_type, const int32_t *ref_len, const int32_t *alt_len, int64_t n,
                              int32_t sv_threshold, uint8_t *out) {
    for (int64_t i = 0; i < n; i++) {
        const int t = var_type[i];
        const bool small = (t == VPR_TYPE_INS && alt_len[i] < sv_threshold) || (t == VPR_TYPE_DEL && ref_len[i] < sv_threshold);
        out[i] = uint8_t(t == VPR_TYPE_SUB ? 0 : (small ? 1 : 2));
    }
}

extern "C" int vpr_phase(const int32_t *sc_phase, const int32_t *phase_set, int32_t n, int32_t *pb_phase,
                         int32_t *switches, int32_t *n_switches, int32_t *flips, int32_t *n_flips) {
    if (n < 0 || (n && (!sc_phase || !phase_set || !pb_phase))) return VPR_ERR_ARG;
    int32_t ns = 0, nf = 0;
    if (n > 0) {
        // cost[k][p]: best cost of phasing superclusters 0..k-1 with supercluster k-1 in phase p; from[k][p]: 1 if the
        // best way into (k, p) switched phase between k-1 and k.  A switch on the border of two phase sets is free.
        std::vector<int32_t> cost(size_t(n + 1) * 2, 0);
        std::vector<uint8_t> from(size_t(n + 1) * 2, 0);
        for (int32_t k = 0; k < n; k++) {
            int32_t pen[2] = {0, 0};
            if (sc_phase[k] == VPR_PHASE_ORIG) pen[1] = 1;
            else if (sc_phase[k] == VPR_PHASE_SWAP) pen[0] = 1;
            else if (sc_phase[k] != VPR_PHASE_NONE) return VPR_ERR_ARG;   // "Unexpected phase"
            const int32_t sw = (k < n - 1 && phase_set[k] != phase_set[k + 1]) ? 0 : 1;
            for (int p = 0; p < 2; p++) {
                const int32_t keep = cost[size_t(k) * 2 + p] + pen[p];
                const int32_t swit = cost[size_t(k) * 2 + (p ^ 1)] + pen[p ^ 1] + sw;
                if (keep < swit) { cost[size_t(k + 1) * 2 + p] = keep; from[size_t(k + 1) * 2 + p] = 0; }
                else { cost[size_t(k + 1) * 2 + p] = swit; from[size_t(k + 1) * 2 + p] = 1; }
            }
        }
        int p = cost[size_t(n) * 2 + 1] < cost[size_t(n) * 2] ? 1 : 0;
        for (int32_t k = n; k > 0; k--) {
            if (from[size_t(k) * 2 + p]) {
                // (k == n never switches on the optimal trace, so phase_set[k] is in range)
                if (k < n && phase_set[k] == phase_set[k - 1]) { if (switches) switches[ns] = k; ns++; }
                p ^= 1;
            } else if (sc_phase[k - 1] != VPR_PHASE_NONE && sc_phase[k - 1] != p) {
                if (flips) flips[nf] = k - 1;
                nf++;
            }
            pb_phase[k - 1] = p;
        }
        if (switches) std::reverse(switches, switches + ns);
        if (flips) std::reverse(flips, flips + nf);
    }
    if (n_switches) *n_switches = ns;
    if (n_flips) *n_flips = nf;
    return VPR_OK;
}

extern "C" int vpr_pr_summary(const int64_t *counts, int32_t min_qual, int32_t max_qual, vpr_pr_row *rows) {
    if (!counts || !rows || max_qual < min_qual) return VPR_ERR_ARG;
    const int nq = max_qual - min_qual + 1;
    auto at = [&](int callset, int type, int err, int qidx) {
        return counts[((size_t(callset) * VPR_VARTYPES + type) * 3 + err) * nq + qidx];
    };
    for (int type = 0; type < VPR_VARTYPES; type++) {
        float max_f1 = 0;
        int best_qual = 0;
        auto metrics = [&](int qual, vpr_pr_row &r) {
            const int qidx = qual - min_qual;
            const int query_tp = int(at(0, type, VPR_ERRTYPE_TP, qidx)), query_fp = int(at(0, type, VPR_ERRTYPE_FP, qidx));
            const int truth_tp = int(at(1, type, VPR_ERRTYPE_TP, qidx)), truth_fn = int(at(1, type, VPR_ERRTYPE_FN, qidx));
            const int query_tot = query_tp + query_fp, truth_tot = truth_tp + truth_fn;
            const float precision = query_tot == 0 ? 1 : float(query_tp) / query_tot;
            const float recall = truth_tot == 0 ? 1 : float(truth_tp) / truth_tot;
            const float f1 = precision + recall > 0 ? 2 * precision * recall / (precision + recall) : 0;
            r.vartype = type; r.qual = qual;
            r.truth_tp = truth_tp; r.query_tp = query_tp; r.truth_fn = truth_fn; r.query_fp = query_fp;
            r.precision = precision; r.recall = recall; r.f1_score = f1;
            r.f1_qscore = float(std::min(100.0, std::max(0.0, -10 * std::log10(double(1 - f1)))));   // qscore(), edit.cpp:102
            return f1;
        };
        vpr_pr_row tmp;
        for (int qual = min_qual; qual <= max_qual; qual++) {   // first maximum wins (print.cpp:470-473)
            const float f1 = metrics(qual, tmp);
            if (f1 > max_f1) { max_f1 = f1; best_qual = qual; }
        }
        if (best_qual < min_qual || best_qual > max_qual) best_qual = min_qual;   // (all-zero F1 leaves 0 in the reference)
        metrics(min_qual, rows[type * 2]);
        rows[type * 2].best = 0;
        metrics(best_qual, rows[type * 2 + 1]);
        rows[type * 2 + 1].best = 1;
    }
    return VPR_OK;
}
