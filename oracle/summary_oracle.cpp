// summary_oracle.cpp -- CPU restatement of the reference's supercluster phasing and precision/recall counting
// (TimD1/vcfdist v2.6.4), TEST INFRASTRUCTURE ONLY.
//   vso_phase      <- phaseblockData::phase        src/phase.cpp:271-355 (one contig)
//   vso_pr_counts  <- write_precision_recall       src/print.cpp:324-441 (the float counters, one contig)
//   vso_pr_summary <- write_precision_recall       src/print.cpp:444-566 (NONE / BEST rows)
// PARITY PIN: part of the chain that reproduces the published SNP / SV rows of the reference's demo/output.txt
// (tests/test_demo_known_answer.py; see pr_oracle.cpp); the reference ships no unit tests for these functions, so
// beyond that known answer there are hand-worked cases only.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../include/vcfdist_pr.h"

extern "C" {

int vso_phase(const int32_t *sc_phase, const int32_t *phase_sets, int32_t n, int32_t *pb_phase,
              int32_t *switches_out, int32_t *n_switches, int32_t *flips_out, int32_t *n_flips) {
    enum { PHASE_ORIG = 0, PHASE_SWAP = 1, PHASE_NONE = 2, PHASE_PTR_KEEP = 0, PHASE_PTR_SWAP = 1 };
    std::vector<std::vector<int>> mat(2, std::vector<int>(n + 1)), ptrs(2, std::vector<int>(n + 1));
    std::vector<int> switches, flips;
    for (int i = 0; i < n; i++) {
        std::vector<int> costs = {0, 0};
        switch (sc_phase[i]) {
            case PHASE_NONE: costs[PHASE_ORIG] = 0; costs[PHASE_SWAP] = 0; break;
            case PHASE_ORIG: costs[PHASE_ORIG] = 0; costs[PHASE_SWAP] = 1; break;
            case PHASE_SWAP: costs[PHASE_ORIG] = 1; costs[PHASE_SWAP] = 0; break;
            default: return VPR_ERR_ARG;
        }
        int cost_swap = 1;
        if (i < n - 1 && phase_sets[i] != phase_sets[i + 1]) cost_swap = 0;
        for (int phase = 0; phase < 2; phase++) {
            if (mat[phase][i] + costs[phase] < mat[phase ^ 1][i] + costs[phase ^ 1] + cost_swap) {
                mat[phase][i + 1] = mat[phase][i] + costs[phase];
                ptrs[phase][i + 1] = PHASE_PTR_KEEP;
            } else {
                mat[phase][i + 1] = mat[phase ^ 1][i] + costs[phase ^ 1] + cost_swap;
                ptrs[phase][i + 1] = PHASE_PTR_SWAP;
            }
        }
    }
    if (n > 0) {
        int phase = PHASE_ORIG;
        if (mat[PHASE_SWAP][n] < mat[PHASE_ORIG][n]) phase = PHASE_SWAP;
        int i = n;
        while (i > 0) {
            if (ptrs[phase][i] == PHASE_PTR_SWAP) {
                if (i < n && phase_sets[i] == phase_sets[i - 1]) switches.push_back(i);   // (i == n: out of range in the reference, never taken)
                phase ^= 1;
            } else if (ptrs[phase][i] == PHASE_PTR_KEEP) {
                if (sc_phase[i - 1] != PHASE_NONE && sc_phase[i - 1] != phase) flips.push_back(i - 1);
            }
            i--;
            pb_phase[i] = phase;
        }
        std::reverse(flips.begin(), flips.end());
        std::reverse(switches.begin(), switches.end());
    }
    for (size_t k = 0; k < switches.size(); k++) switches_out[k] = switches[k];
    for (size_t k = 0; k < flips.size(); k++) flips_out[k] = flips[k];
    *n_switches = int(switches.size());
    *n_flips = int(flips.size());
    return VPR_OK;
}

// one contig: var_off[slot][n_sc+1], errtype/callq[slot][swap][n_var], cls[slot][n_var] in {0 SNP, 1 INDEL, 2 SV}
int vso_pr_counts(int32_t n_sc, const int64_t *const var_off[4], const uint8_t *const errtype[4][2],
                  const float *const callq[4][2], const uint8_t *const cls[4], const int32_t *sc_phase,
                  const int32_t *pb_phase, int32_t min_qual, int32_t max_qual, int64_t *counts) {
    const int nq = max_qual - min_qual + 1;
    std::vector<std::vector<std::vector<float>>> cnt[2];
    for (int c = 0; c < 2; c++) cnt[c].assign(VPR_VARTYPES, std::vector<std::vector<float>>(3, std::vector<float>(nq, 0.0f)));
    for (int slot = 0; slot < 4; slot++) {
        const int callset = slot >> 1;
        for (int sci = 0; sci < n_sc; sci++) {
            bool swap;
            switch (sc_phase[sci]) {
                case VPR_PHASE_ORIG: swap = false; break;
                case VPR_PHASE_SWAP: swap = true; break;
                default: swap = pb_phase ? pb_phase[sci] : 0; break;
            }
            for (int64_t i = var_off[slot][sci]; i < var_off[slot][sci + 1]; i++) {
                float q = callq[slot][swap][i];
                int t = cls[slot][i];
                int e = errtype[slot][swap][i];
                if (e == VPR_ERRTYPE_UN || e > 2) continue;
                for (int qual = min_qual; qual <= q; qual++) {
                    if (qual - min_qual >= nq) break;
                    cnt[callset][t][e][qual - min_qual]++;
                    cnt[callset][VPR_VARTYPE_ALL][e][qual - min_qual]++;
                }
                if (callset == 1)
                    for (int qual = q + 1; qual <= max_qual; qual++) {
                        if (qual < min_qual) continue;
                        cnt[callset][t][VPR_ERRTYPE_FN][qual - min_qual]++;
                        cnt[callset][VPR_VARTYPE_ALL][VPR_ERRTYPE_FN][qual - min_qual]++;
                    }
            }
        }
    }
    for (int c = 0; c < 2; c++)
        for (int t = 0; t < VPR_VARTYPES; t++)
            for (int e = 0; e < 3; e++)
                for (int k = 0; k < nq; k++) counts[((size_t(c) * VPR_VARTYPES + t) * 3 + e) * nq + k] = int64_t(cnt[c][t][e][k]);
    return VPR_OK;
}

int vso_pr_summary(const int64_t *counts, int32_t min_qual, int32_t max_qual, vpr_pr_row *rows) {
    const int nq = max_qual - min_qual + 1;
    auto qc = [&](int type, int e, int qidx) { return float(counts[((size_t(0) * VPR_VARTYPES + type) * 3 + e) * nq + qidx]); };
    auto tc = [&](int type, int e, int qidx) { return float(counts[((size_t(1) * VPR_VARTYPES + type) * 3 + e) * nq + qidx]); };
    std::vector<float> max_f1_score(VPR_VARTYPES, 0);
    std::vector<int> max_f1_qual(VPR_VARTYPES, 0);
    for (int type = 0; type < VPR_VARTYPES; type++)
        for (int qual = min_qual; qual <= max_qual; qual++) {
            int qidx = qual - min_qual;
            int query_tp = qc(type, 0, qidx), query_fp = qc(type, 1, qidx), query_tot = query_tp + query_fp;
            int truth_tp = tc(type, 0, qidx), truth_fn = tc(type, 2, qidx), truth_tot = truth_tp + truth_fn;
            float precision = query_tot == 0 ? 1 : float(query_tp) / query_tot;
            float recall = truth_tot == 0 ? 1 : float(truth_tp) / truth_tot;
            float f1_score = precision + recall ? 2 * precision * recall / (precision + recall) : 0;
            if (f1_score > max_f1_score[type]) { max_f1_score[type] = f1_score; max_f1_qual[type] = qual; }
        }
    for (int type = 0; type < VPR_VARTYPES; type++) {
        std::vector<int> quals = {min_qual, max_f1_qual[type]};
        for (int i = 0; i < 2; i++) {
            int qual = quals[i];
            if (qual < min_qual) qual = min_qual;
            int qidx = qual - min_qual;
            int query_tp = qc(type, 0, qidx), query_fp = qc(type, 1, qidx), query_tot = query_tp + query_fp;
            int truth_tp = tc(type, 0, qidx), truth_fn = tc(type, 2, qidx), truth_tot = truth_tp + truth_fn;
            float precision = query_tot == 0 ? 1 : float(query_tp) / query_tot;
            float recall = truth_tot == 0 ? 1 : float(truth_tp) / truth_tot;
            float f1_score = precision + recall > 0 ? 2 * precision * recall / (precision + recall) : 0;
            vpr_pr_row &r = rows[type * 2 + i];
            r.vartype = type; r.best = i; r.qual = qual;
            r.truth_tp = truth_tp; r.query_tp = query_tp; r.truth_fn = truth_fn; r.query_fp = query_fp;
            r.precision = precision; r.recall = recall; r.f1_score = f1_score;
            r.f1_qscore = float(std::min(100.0, std::max(0.0, -10 * std::log10(double(1 - f1_score)))));
        }
    }
    return VPR_OK;
}

}  // extern "C"
