// wfa_oracle.cpp -- CPU restatement of the reference's biWFA dependency clustering (TimD1/vcfdist v2.6.4),
// TEST INFRASTRUCTURE ONLY: imported from tests/ to check the HIP implementation behind vcl_wfa_cluster.
// Data shapes and loop structure follow the reference so that each block can be read against the cited lines.
//
//   gen_str()        <- generate_str        src/dist.cpp:81-136   (min_qual = 0: every variant is applied)
//   swg_align()      <- wf_swg_align        src/dist.cpp:1510-1652 (only the score is used by the caller)
//   swg_max_reach()  <- wf_swg_max_reach    src/dist.cpp:2150-2333
//   vco_wfa_cluster  <- wf_swg_cluster      src/cluster.cpp:954-1263
//
// PARITY PIN: part of the chain that reproduces the published SNP / SV rows of the reference's demo/output.txt
// (tests/test_demo_known_answer.py; see pr_oracle.cpp); the reference ships no unit tests for these functions, so
// beyond that known answer there are hand-worked cases only.
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/vcfdist_cluster.h"

namespace {

enum { MAT_SUB = 0, MAT_INS = 1, MAT_DEL = 2, MATS = 3 };   // defs.h:131-134
enum { TYPE_SUB = 1, TYPE_INS = 2, TYPE_DEL = 3 };          // defs.h:33-35

struct Vars {                       // ctgVariants, variant.h:29-40
    int n = 0;
    std::vector<int> poss, rlens, types;
    std::vector<std::string> refs, alts;
    std::vector<int> clusters;
};

// dist.cpp:81-136
std::string gen_str(const std::string &fasta, const Vars &vars, int beg_idx, int end_idx, int beg_pos, int end_pos) {
    int var_idx = beg_idx;
    std::string str = "";
    while (var_idx < vars.n && vars.poss[var_idx] < beg_pos) var_idx++;
    for (int ref_pos = beg_pos; ref_pos < end_pos;) {
        if (var_idx < end_idx && ref_pos == vars.poss[var_idx]) {
            switch (vars.types[var_idx]) {
                case TYPE_INS: str += vars.alts[var_idx]; break;
                case TYPE_DEL: ref_pos += vars.refs[var_idx].size(); break;
                case TYPE_SUB: str += vars.alts[var_idx]; ref_pos++; break;
            }
            var_idx++;
        } else {
            int ref_end = (var_idx < end_idx) ? std::min(end_pos, vars.poss[var_idx]) : end_pos;
            if (ref_end < ref_pos) return str;   // the reference ERRORs here (overlapping variants)
            str += fasta.substr(ref_pos, ref_end - ref_pos);
            ref_pos = ref_end;
        }
    }
    return str;
}

// dist.cpp:1510-1652; returns the alignment score s
int swg_align(const std::string &query, const std::string &truth, int x, int o, int e) {
    int query_len = query.size(), truth_len = truth.size();
    int mat_len = query_len + truth_len - 1;
    bool done = false;
    std::vector<std::vector<std::vector<int>>> offs(MATS);
    for (int m = 0; m < MATS; m++) offs[m].push_back(std::vector<int>(mat_len, -2));
    int s = 0;
    offs[MAT_SUB][s][query_len - 1] = -1;
    while (true) {
        for (int m = MAT_INS; m < MATS; m++) {
            for (int d = 0; d < mat_len; d++) {
                int off = offs[m][s][d];
                int diag = d + 1 - query_len;
                if (off >= 0 && off < query_len && diag + off >= 0 && diag + off < truth_len &&
                    offs[m][s][d] >= offs[MAT_SUB][s][d])
                    offs[MAT_SUB][s][d] = offs[m][s][d];
            }
        }
        for (int d = 0; d < mat_len; d++) {
            int off = offs[MAT_SUB][s][d];
            int diag = d + 1 - query_len;
            while (off != -2 && diag + off >= -1 && off < query_len - 1 && diag + off < truth_len - 1) {
                if (query[off + 1] == truth[diag + off + 1]) off++;
                else break;
            }
            offs[MAT_SUB][s][d] = off;
            if (off == query_len - 1 && off + diag == truth_len - 1) { done = true; break; }
        }
        if (done) break;
        s++;
        for (int m = 0; m < MATS; m++) offs[m].push_back(std::vector<int>(mat_len, -2));
        for (int d = 0; d < mat_len; d++) {
            int diag = d + 1 - query_len;
            if (s - x >= 0 && offs[MAT_SUB][s - x][d] != -2 && offs[MAT_SUB][s - x][d] + 1 < query_len &&
                diag + offs[MAT_SUB][s - x][d] + 1 < truth_len && offs[MAT_SUB][s - x][d] + 1 >= offs[MAT_SUB][s][d])
                offs[MAT_SUB][s][d] = offs[MAT_SUB][s - x][d] + 1;
            if (s - (o + e) >= 0 && d > 0 && offs[MAT_SUB][s - (o + e)][d - 1] != -2 &&
                diag + offs[MAT_SUB][s - (o + e)][d - 1] < truth_len &&
                offs[MAT_SUB][s - (o + e)][d - 1] >= offs[MAT_DEL][s][d])
                offs[MAT_DEL][s][d] = offs[MAT_SUB][s - (o + e)][d - 1];
            if (s - (o + e) >= 0 && d < mat_len - 1 && offs[MAT_SUB][s - (o + e)][d + 1] != -2 &&
                offs[MAT_SUB][s - (o + e)][d + 1] + 1 < query_len && diag + offs[MAT_SUB][s - (o + e)][d + 1] + 1 < truth_len &&
                diag + offs[MAT_SUB][s - (o + e)][d + 1] + 1 >= 0 &&
                offs[MAT_SUB][s - (o + e)][d + 1] + 1 >= offs[MAT_INS][s][d])
                offs[MAT_INS][s][d] = offs[MAT_SUB][s - (o + e)][d + 1] + 1;
            if (s - e >= 0 && d > 0 && offs[MAT_DEL][s - e][d - 1] != -2 && diag + offs[MAT_DEL][s - e][d - 1] < truth_len &&
                offs[MAT_DEL][s - e][d - 1] >= offs[MAT_DEL][s][d])
                offs[MAT_DEL][s][d] = offs[MAT_DEL][s - e][d - 1];
            if (s - e >= 0 && d < mat_len - 1 && offs[MAT_INS][s - e][d + 1] != -2 && offs[MAT_INS][s - e][d + 1] + 1 < query_len &&
                diag + offs[MAT_INS][s - e][d + 1] + 1 < truth_len && diag + offs[MAT_INS][s - e][d + 1] + 1 >= 0 &&
                offs[MAT_INS][s - e][d + 1] + 1 >= offs[MAT_INS][s][d])
                offs[MAT_INS][s][d] = offs[MAT_INS][s - e][d + 1] + 1;
        }
    }
    return s;
}

// dist.cpp:2150-2333; offs is the caller's buffer, all -2 on entry
int swg_max_reach(const std::string &query, const std::string &truth, std::vector<int> &offs, int main_diag,
                  int main_diag_start, int max_score, int x, int o, int e, bool reverse) {
    int query_len = query.size(), truth_len = truth.size();
    int mat_len = query_len + truth_len - 1;
    int main_diag_off = main_diag_start - main_diag;
    int s = 0, s2 = 0;
    int scores = std::max(x, o + e) + 1;
    int y = mat_len;
    int z = y * scores;
    offs[MAT_SUB * z + s2 * y + query_len - 1] = -1;
    while (true) {
        if (!reverse)
            for (int m = MAT_INS; m < MATS; m++)
                for (int d = 0; d < mat_len; d++) {
                    int off = offs[m * z + s2 * y + d];
                    int diag = d + 1 - query_len;
                    if (off >= 0 && off < query_len && diag + off >= 0 && diag + off < truth_len &&
                        off >= offs[MAT_SUB * z + s2 * y + d])
                        offs[MAT_SUB * z + s2 * y + d] = off;
                }
        for (int d = 0; d < mat_len; d++) {
            int off = offs[MAT_SUB * z + s2 * y + d];
            int diag = d + 1 - query_len;
            while ((diag != main_diag || off + 1 < main_diag_off) && off != -2 && diag + off >= -1 &&
                   off < query_len - 1 && diag + off < truth_len - 1) {
                if (query[off + 1] == truth[diag + off + 1]) off++;
                else break;
            }
            offs[MAT_SUB * z + s2 * y + d] = off;
            if (off + diag == truth_len - 1) return truth_len - 1;
            if (off == query_len - 1 && off + diag >= 0 && off + diag < truth_len - 1) return off + diag;
        }
        if (s == max_score) break;
        s++; s2++;
        if (s2 == scores) s2 = 0;
        for (int m = MAT_INS; m < MATS; m++)
            for (int d = 0; d < mat_len; d++) offs[m * z + s2 * y + d] = -2;
        for (int d = 0; d < mat_len; d++) {
            int diag = d + 1 - query_len;
            int p = s - x, p2 = s2 - x;
            if (p2 < 0) p2 += scores;
            if (p >= 0 && offs[MAT_SUB * z + p2 * y + d] != -2 && offs[MAT_SUB * z + p2 * y + d] + 1 < query_len &&
                diag + offs[MAT_SUB * z + p2 * y + d] + 1 < truth_len &&
                offs[MAT_SUB * z + p2 * y + d] + 1 >= offs[MAT_SUB * z + s2 * y + d])
                offs[MAT_SUB * z + s2 * y + d] = offs[MAT_SUB * z + p2 * y + d] + 1;
            p = reverse ? s - e : s - (o + e);
            p2 = reverse ? s2 - e : s2 - (o + e);
            if (p2 < 0) p2 += scores;
            if (p >= 0 && d > 0 && offs[MAT_SUB * z + p2 * y + d - 1] != -2 && diag + offs[MAT_SUB * z + p2 * y + d - 1] < truth_len &&
                offs[MAT_SUB * z + p2 * y + d - 1] >= offs[MAT_DEL * z + s2 * y + d])
                offs[MAT_DEL * z + s2 * y + d] = offs[MAT_SUB * z + p2 * y + d - 1];
            if (p >= 0 && d < mat_len - 1 && offs[MAT_SUB * z + p2 * y + d + 1] != -2 &&
                offs[MAT_SUB * z + p2 * y + d + 1] + 1 < query_len && diag + offs[MAT_SUB * z + p2 * y + d + 1] + 1 < truth_len &&
                diag + offs[MAT_SUB * z + p2 * y + d + 1] + 1 >= 0 &&
                offs[MAT_SUB * z + p2 * y + d + 1] + 1 >= offs[MAT_INS * z + s2 * y + d])
                offs[MAT_INS * z + s2 * y + d] = offs[MAT_SUB * z + p2 * y + d + 1] + 1;
            p = s - o; p2 = s2 - o;
            if (p2 < 0) p2 += scores;
            if (reverse && p >= 0)
                for (int m = MAT_INS; m < MATS; m++)
                    if (offs[m * z + p2 * y + d] >= 0 && offs[m * z + p2 * y + d] < query_len && diag + offs[m * z + p2 * y + d] >= 0 &&
                        diag + offs[m * z + p2 * y + d] < truth_len && offs[m * z + p2 * y + d] > offs[MAT_SUB * z + s2 * y + d])
                        offs[MAT_SUB * z + s2 * y + d] = offs[m * z + p2 * y + d];
            p = s - e; p2 = s2 - e;
            if (p2 < 0) p2 += scores;
            if (p >= 0 && d > 0 && offs[MAT_DEL * z + p2 * y + d - 1] != -2 && diag + offs[MAT_DEL * z + p2 * y + d - 1] < truth_len &&
                offs[MAT_DEL * z + p2 * y + d - 1] >= offs[MAT_DEL * z + s2 * y + d])
                offs[MAT_DEL * z + s2 * y + d] = offs[MAT_DEL * z + p2 * y + d - 1];
            if (p >= 0 && d < mat_len - 1 && offs[MAT_INS * z + p2 * y + d + 1] != -2 &&
                offs[MAT_INS * z + p2 * y + d + 1] + 1 < query_len && diag + offs[MAT_INS * z + p2 * y + d + 1] + 1 < truth_len &&
                diag + offs[MAT_INS * z + p2 * y + d + 1] + 1 >= 0 &&
                offs[MAT_INS * z + p2 * y + d + 1] + 1 >= offs[MAT_INS * z + s2 * y + d])
                offs[MAT_INS * z + s2 * y + d] = offs[MAT_INS * z + p2 * y + d + 1] + 1;
        }
    }
    int max_reach = 0;
    for (int r = 0; r < scores; r++)
        for (int m = 0; m < MATS; m++)
            for (int d = 0; d < mat_len; d++) {
                int off = offs[m * z + r * y + d];
                int diag = d + 1 - query_len;
                if (off >= 0 && off < query_len && diag + off >= 0 && diag + off < truth_len)
                    max_reach = std::max(max_reach, diag + off);
            }
    return max_reach;
}

}  // namespace

extern "C" {

// cluster.cpp:954-1263 for one (callset, hap) on one contig
int vco_wfa_cluster(const vcl_hap_seq *hs, const uint8_t *ctg_seq, int32_t ctg_len, int32_t sub, int32_t open,
                    int32_t extend, int32_t max_cluster_itrs, int32_t reach_min_gap, vcl_clusters **out,
                    vcl_wfa_stats *stats) {
    const vcl_hap *h = &hs->cols;
    Vars vars;
    vars.n = h->n_var;
    for (int v = 0; v < vars.n; v++) {
        vars.poss.push_back(h->pos[v]);
        vars.rlens.push_back(h->rlen[v]);
        vars.types.push_back(h->type[v]);
        vars.refs.emplace_back(reinterpret_cast<const char *>(hs->pool + hs->ref_off[v]), size_t(h->ref_len[v]));
        vars.alts.emplace_back(reinterpret_cast<const char *>(hs->pool + hs->alt_off[v]), size_t(h->alt_len[v]));
    }
    const std::string fasta(reinterpret_cast<const char *>(ctg_seq), size_t(ctg_len));
    vcl_wfa_stats st;
    memset(&st, 0, sizeof(st));
    std::vector<int> left_reach, right_reach, prev_clusters;
    if (vars.n) {
        std::vector<int> offs_buffer(1024, -2);
        prev_clusters.resize(vars.n + 1);
        for (int i = 0; i < vars.n + 1; i++) prev_clusters[i] = i;
        std::vector<bool> prev_active(vars.n + 1, true);
        right_reach.assign(vars.n + 1, 0);
        left_reach.assign(vars.n + 1, 0);
        std::vector<int> next_clusters, tmp_clusters;
        std::vector<bool> next_active, tmp_active;
        int iter = 0;
        while (std::find(prev_active.begin(), prev_active.end(), true) != prev_active.end()) {
            iter++;
            if (iter > max_cluster_itrs) break;
            st.iterations = iter;
            vars.clusters = prev_clusters;
            left_reach[prev_clusters.size() - 1] = INT_MAX;
            right_reach[prev_clusters.size() - 1] = INT_MAX;
            for (size_t clust = 0; clust < prev_clusters.size(); clust++) {
                bool left_compute = prev_active[clust], right_compute = prev_active[clust];
                if (clust == prev_clusters.size() - 1) { left_compute = false; right_compute = false; }
                int score = 0;
                if (left_compute || right_compute) {
                    int beg_idx = vars.clusters[clust], end_idx = vars.clusters[clust + 1];
                    int beg = std::max(0, vars.poss[beg_idx] - 1);
                    int end = std::min(ctg_len, vars.poss[end_idx - 1] + vars.rlens[end_idx - 1] + 1);
                    std::string query = gen_str(fasta, vars, beg_idx, end_idx, beg, end);
                    std::string ref = fasta.substr(beg, end - beg);
                    score = swg_align(query, ref, sub, open, extend);
                    st.align_calls++;
                }
                if (left_compute) {
                    std::string query, ref;
                    int beg_pos = vars.poss[vars.clusters[clust]] - 1;
                    int end_pos = vars.poss[vars.clusters[clust + 1] - 1] + vars.rlens[vars.clusters[clust + 1] - 1] + 1;
                    int main_diag_start = end_pos - vars.poss[vars.clusters[clust]];
                    int main_diag = 0;
                    for (int vi = vars.clusters[clust]; vi < vars.clusters[clust + 1]; vi++)
                        main_diag += int(vars.refs[vi].size()) - int(vars.alts[vi].size());
                    int ref_len = end_pos - beg_pos;
                    int reach = ref_len - 1;
                    while (reach == ref_len - 1) {
                        ref_len *= 2;
                        beg_pos = std::max(0, end_pos - ref_len - std::abs(main_diag) - score / extend - 3);
                        query = gen_str(fasta, vars, vars.clusters[clust], vars.clusters[clust + 1], beg_pos, end_pos);
                        ref = fasta.substr(std::max(0, end_pos - ref_len), ref_len);
                        std::reverse(query.begin(), query.end());
                        std::reverse(ref.begin(), ref.end());
                        size_t offs_size = MATS * (std::max(open + extend, sub) + 1) * (query.size() + ref.size() - 1);
                        if (offs_size > offs_buffer.size()) offs_buffer.resize(offs_size, -2);
                        reach = swg_max_reach(query, ref, offs_buffer, main_diag, main_diag_start, score, sub, open, extend, true);
                        st.reach_calls++;
                        for (size_t i = 0; i < offs_size; i++) offs_buffer[i] = -2;
                        if (beg_pos == 0) break;
                    }
                    left_reach[clust] = end_pos - reach;
                }
                if (right_compute) {
                    std::string query, ref;
                    int beg_pos = vars.poss[vars.clusters[clust]] - 1;
                    int end_pos = vars.poss[vars.clusters[clust + 1] - 1] + vars.rlens[vars.clusters[clust + 1] - 1] + 1;
                    int main_diag_start = vars.poss[vars.clusters[clust + 1] - 1] + vars.rlens[vars.clusters[clust + 1] - 1] - beg_pos;
                    int main_diag = 0;
                    for (int vi = vars.clusters[clust]; vi < vars.clusters[clust + 1]; vi++)
                        main_diag += int(vars.refs[vi].size()) - int(vars.alts[vi].size());
                    int ref_len = end_pos - beg_pos;
                    int reach = ref_len - 1;
                    while (reach == ref_len - 1) {
                        ref_len *= 2;
                        end_pos = std::min(ctg_len, beg_pos + ref_len + std::abs(main_diag) + score / extend + 3);
                        query = gen_str(fasta, vars, vars.clusters[clust], vars.clusters[clust + 1], beg_pos, end_pos);
                        ref = fasta.substr(beg_pos, std::min(ref_len, end_pos - beg_pos));
                        size_t offs_size = MATS * (std::max(sub, open + extend) + 1) * (query.size() + ref.size() - 1);
                        if (offs_size > offs_buffer.size()) offs_buffer.resize(offs_size, -2);
                        reach = swg_max_reach(query, ref, offs_buffer, main_diag, main_diag_start, score, sub, open, extend, false);
                        st.reach_calls++;
                        for (size_t i = 0; i < offs_size; i++) offs_buffer[i] = -2;
                        if (end_pos == ctg_len) break;
                    }
                    right_reach[clust] = beg_pos + reach + 1;
                }
            }
            // merge dependent clusters rightwards, cluster.cpp:1173-1193
            std::vector<int> tmp_left_reach, tmp_right_reach;
            int clust = 0;
            while (clust < int(prev_clusters.size())) {
                int clust_size = 1;
                int max_right_reach = right_reach[clust];
                int min_left_reach = left_reach[clust];
                while (clust + clust_size < int(prev_clusters.size()) &&
                       max_right_reach + reach_min_gap >= left_reach[clust + clust_size]) {
                    max_right_reach = std::max(max_right_reach, right_reach[clust + clust_size]);
                    min_left_reach = std::min(min_left_reach, left_reach[clust + clust_size]);
                    clust_size++;
                }
                tmp_right_reach.push_back(max_right_reach);
                tmp_left_reach.push_back(min_left_reach);
                tmp_clusters.push_back(prev_clusters[clust]);
                tmp_active.push_back(clust_size > 1);
                clust += clust_size;
            }
            left_reach.clear();
            right_reach.clear();
            // merge dependent clusters leftwards, cluster.cpp:1208-1232
            clust = tmp_clusters.size() - 1;
            while (clust >= 0) {
                int min_left_reach = tmp_left_reach[clust];
                int max_right_reach = tmp_right_reach[clust];
                bool active = tmp_active[clust];
                while (clust > 0 && min_left_reach <= tmp_right_reach[clust - 1] + reach_min_gap) {
                    min_left_reach = std::min(min_left_reach, tmp_left_reach[clust - 1]);
                    max_right_reach = std::max(max_right_reach, tmp_right_reach[clust - 1]);
                    active = true;
                    clust--;
                }
                left_reach.push_back(min_left_reach);
                right_reach.push_back(max_right_reach);
                next_clusters.push_back(tmp_clusters[clust]);
                next_active.push_back(active);
                clust--;
            }
            std::reverse(next_clusters.begin(), next_clusters.end());
            std::reverse(next_active.begin(), next_active.end());
            std::reverse(left_reach.begin(), left_reach.end());
            std::reverse(right_reach.begin(), right_reach.end());
            prev_clusters = next_clusters;
            prev_active = next_active;
            tmp_clusters.clear(); tmp_active.clear(); next_clusters.clear(); next_active.clear();
        }
    }
    vcl_clusters *c = static_cast<vcl_clusters *>(calloc(1, sizeof(vcl_clusters)));
    const size_t m = prev_clusters.size();
    c->n = m ? int(m) - 1 : 0;
    c->var_beg = static_cast<int32_t *>(malloc((m + 1) * 4));
    c->left_reach = static_cast<int32_t *>(malloc((m + 1) * 4));
    c->right_reach = static_cast<int32_t *>(malloc((m + 1) * 4));
    for (size_t k = 0; k < m; k++) {
        c->var_beg[k] = prev_clusters[k];
        c->left_reach[k] = left_reach[k];
        c->right_reach[k] = right_reach[k];
    }
    *out = c;
    if (stats) *stats = st;
    return VCL_OK;
}

}  // extern "C"
