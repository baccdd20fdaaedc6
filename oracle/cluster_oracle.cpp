// cluster_oracle.cpp -- CPU restatement of the reference's distance clustering and superclustering
// (TimD1/vcfdist v2.6.4), TEST INFRASTRUCTURE ONLY: imported from tests/ to check vcfdist_amd/csrc/cluster.cpp.
// It keeps the reference's data shapes (per-hap std::vector tables that are edited in place, a vector of
// breakpoint 4-tuples) and loop structure so that each block can be read against the cited lines.
//
//   vco_simple_cluster  <- simple_cluster                      src/cluster.cpp:826-945
//   vco_supercluster    <- superclusterData::supercluster      src/cluster.cpp:404-553
//       range()         <- get_supercluster_range              src/cluster.cpp:563-594
//       split_large()   <- split_large_supercluster            src/cluster.cpp:601-648
//       split_one()     <- split_cluster                       src/cluster.cpp:655-706
//       next_var()      <- get_next_variant_info               src/cluster.cpp:709-733
//       split_where()   <- get_supercluster_split_location     src/cluster.cpp:738-808
//
// PARITY PIN: part of the chain that reproduces the published SNP / SV rows of the reference's demo/output.txt
// (tests/test_demo_known_answer.py; see pr_oracle.cpp); the reference ships no unit tests for these functions, so
// beyond that known answer there are hand-worked cases only.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/vcfdist_cluster.h"

namespace {

struct HapVars {                       // ctgVariants, variant.h:29-40 (only what these functions read)
    int n = 0;
    std::vector<int> poss, rlens, types, ref_sz, alt_sz;
    std::vector<int> clusters, left_reaches, right_reaches;
};

struct VarInfo { int hap_idx, start_pos, end_pos; };   // var_info, cluster.h

void load(const vcl_hap *h, HapVars &v) {
    v.n = h->n_var;
    v.poss.assign(h->pos, h->pos + v.n);
    v.rlens.assign(h->rlen, h->rlen + v.n);
    if (h->type) v.types.assign(h->type, h->type + v.n);
    if (h->ref_len) v.ref_sz.assign(h->ref_len, h->ref_len + v.n);
    if (h->alt_len) v.alt_sz.assign(h->alt_len, h->alt_len + v.n);
}

vcl_clusters *dump(const HapVars &v) {
    vcl_clusters *c = static_cast<vcl_clusters *>(calloc(1, sizeof(vcl_clusters)));
    const size_t m = v.clusters.size();
    c->n = m ? int(m) - 1 : 0;
    c->var_beg = static_cast<int32_t *>(malloc((m + 1) * 4));
    c->left_reach = static_cast<int32_t *>(malloc((m + 1) * 4));
    c->right_reach = static_cast<int32_t *>(malloc((m + 1) * 4));
    for (size_t k = 0; k < m; k++) {
        c->var_beg[k] = v.clusters[k];
        c->left_reach[k] = v.left_reaches[k];
        c->right_reach[k] = v.right_reaches[k];
    }
    return c;
}

// cluster.cpp:563-594
std::vector<int> range(const HapVars vars[4], const std::vector<int> &start, const std::vector<int> &stop) {
    int beg_pos = INT_MAX, end_pos = -1;
    for (int i = 0; i < 4; i++) {
        if (stop[i] - start[i]) {
            beg_pos = std::min(beg_pos, vars[i].poss[vars[i].clusters[start[i]]] - 1);
            end_pos = std::max(end_pos, vars[i].poss[vars[i].clusters[stop[i]] - 1] +
                                            vars[i].rlens[vars[i].clusters[stop[i]] - 1] + 1);
        }
    }
    return {beg_pos, end_pos};
}

// cluster.cpp:709-733
VarInfo next_var(const HapVars vars[4], const std::vector<int> &curr, const std::vector<int> &stop) {
    int next_idx = -1, next_start = INT_MAX, next_end = INT_MAX;
    for (int i = 0; i < 4; i++) {
        if (curr[i] < stop[i]) {
            const int start_pos = vars[i].poss[curr[i]];
            const int end_pos = vars[i].poss[curr[i]] + vars[i].rlens[curr[i]];
            if (start_pos < next_start) { next_start = start_pos; next_end = end_pos; next_idx = i; }
        }
    }
    return VarInfo{next_idx, next_start, next_end};
}

// cluster.cpp:738-808
std::vector<int> split_where(const HapVars vars[4], const std::vector<int> &cstart, const std::vector<int> &cstop) {
    const std::vector<int> orig = range(vars, cstart, cstop);
    const int orig_beg = orig[0], orig_end = orig[1], orig_size = orig_end - orig_beg;
    std::vector<int> var_start(4, 0), var_stop(4, 0);
    for (int i = 0; i < 4; i++)
        if (vars[i].clusters.size()) {
            var_start[i] = vars[i].clusters[cstart[i]];
            var_stop[i] = vars[i].clusters[cstop[i]];
        }
    std::vector<int> split_indices = var_start;
    double best_split_score = 0;
    std::vector<int> best = {};
    int total_vars = 0;
    for (int i = 0; i < 4; i++) total_vars += var_stop[i] - var_start[i];
    if (total_vars < 2) return best;
    VarInfo curr = next_var(vars, split_indices, var_stop);
    split_indices[curr.hap_idx]++;
    VarInfo nxt = next_var(vars, split_indices, var_stop);
    while (nxt.hap_idx >= 0) {
        int gap = std::max(0, nxt.start_pos - curr.end_pos);
        double size_reduction_factor = std::max(double((curr.end_pos + gap / 2) - orig_beg) / orig_size,
                                                double(orig_end - (curr.end_pos + gap / 2)) / orig_size);
        double splits_to_halve_size = -1 / log2(size_reduction_factor);
        double split_score = gap / splits_to_halve_size;
        if (split_score > best_split_score) { best_split_score = split_score; best = split_indices; }
        curr = nxt;
        split_indices[curr.hap_idx]++;
        nxt = next_var(vars, split_indices, var_stop);
    }
    return best;
}

// cluster.cpp:655-706
std::vector<int> split_one(HapVars vars[4], const std::vector<int> &var_split, std::vector<std::vector<int>> &breakpoints,
                           int breakpoint_idx) {
    std::vector<int> cluster_curr(4, 0);
    for (int i = 0; i < 4; i++) {
        if (!vars[i].clusters.size()) continue;
        int var_idx = var_split[i];
        auto itr = std::lower_bound(vars[i].clusters.begin(), vars[i].clusters.end(), var_idx);
        int clust_idx = std::distance(vars[i].clusters.begin(), itr);
        cluster_curr[i] = clust_idx;
        if (*itr == var_idx) {
            // already a cluster break here
        } else {
            int right_reach = vars[i].right_reaches[clust_idx - 1];
            int var_pos = vars[i].poss[var_idx];
            vars[i].right_reaches[clust_idx - 1] = var_pos;
            vars[i].left_reaches.insert(vars[i].left_reaches.begin() + clust_idx, var_pos);
            vars[i].right_reaches.insert(vars[i].right_reaches.begin() + clust_idx, right_reach);
            vars[i].clusters.insert(vars[i].clusters.begin() + clust_idx, var_idx);
            for (int j = breakpoint_idx + 1; j < int(breakpoints.size()); j++) breakpoints[j][i]++;
        }
    }
    return cluster_curr;
}

// cluster.cpp:601-648
std::vector<std::vector<int>> split_large(HapVars vars[4], const std::vector<int> &cstart, std::vector<int> &cstop,
                                          int max_size, int &n_unsplittable) {
    std::vector<std::vector<int>> breakpoints = {cstart, cstop};
    bool large_exists = true;
    while (large_exists) {
        large_exists = false;
        std::vector<std::vector<int>> next_breakpoints;
        for (int i = 0; i < int(breakpoints.size()) - 1; i++) {
            std::vector<int> poss = range(vars, breakpoints[i], breakpoints[i + 1]);
            if (poss[1] - poss[0] > max_size) {
                large_exists = true;
                next_breakpoints.push_back(breakpoints[i]);
                std::vector<int> best = split_where(vars, breakpoints[i], breakpoints[i + 1]);
                if (int(best.size()) == 4) {
                    std::vector<int> idx = split_one(vars, best, breakpoints, i);
                    next_breakpoints.push_back(idx);
                } else {
                    n_unsplittable++;
                    large_exists = false;
                }
            } else {
                next_breakpoints.push_back(breakpoints[i]);
            }
        }
        cstop = breakpoints[breakpoints.size() - 1];
        next_breakpoints.push_back(cstop);
        breakpoints = next_breakpoints;
    }
    return breakpoints;
}

}  // namespace

extern "C" {

// cluster.cpp:826-945
int vco_simple_cluster(const vcl_hap *hap, int size_mode, int32_t cluster_min_gap, int32_t reach_min_gap,
                       vcl_clusters **out) {
    HapVars vars;
    load(hap, vars);
    if (vars.n) {
        std::vector<int> prev_clusters(vars.n + 1);
        for (int i = 0; i < vars.n + 1; i++) prev_clusters[i] = i;
        std::vector<int> right_reach(vars.n + 1), left_reach(vars.n + 1);
        std::vector<int> next_clusters, tmp_clusters;
        left_reach[prev_clusters.size() - 1] = INT_MAX;
        right_reach[prev_clusters.size() - 1] = INT_MAX;
        for (int var = 0; var < vars.n; var++) {
            int var_size = 0;
            if (size_mode) {
                switch (vars.types[var]) {
                    case 1: var_size = 1; break;                       // TYPE_SUB
                    case 2: var_size = vars.alt_sz[var]; break;        // TYPE_INS
                    case 3: var_size = vars.ref_sz[var]; break;        // TYPE_DEL
                    default: return VCL_ERR_TYPE;
                }
            }
            left_reach[var] = vars.poss[var] - std::max(cluster_min_gap, var_size);
            right_reach[var] = vars.poss[var] + vars.rlens[var] + std::max(cluster_min_gap, var_size);
        }
        // merge dependent clusters rightwards
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
            clust += clust_size;
        }
        left_reach.clear();
        right_reach.clear();
        // merge dependent clusters leftwards
        clust = tmp_clusters.size() - 1;
        while (clust >= 0) {
            int min_left_reach = tmp_left_reach[clust];
            int max_right_reach = tmp_right_reach[clust];
            while (clust > 0 && min_left_reach <= tmp_right_reach[clust - 1] + reach_min_gap) {
                min_left_reach = std::min(min_left_reach, tmp_left_reach[clust - 1]);
                max_right_reach = std::max(max_right_reach, tmp_right_reach[clust - 1]);
                clust--;
            }
            left_reach.push_back(min_left_reach);
            right_reach.push_back(max_right_reach);
            next_clusters.push_back(tmp_clusters[clust]);
            clust--;
        }
        std::reverse(next_clusters.begin(), next_clusters.end());
        std::reverse(left_reach.begin(), left_reach.end());
        std::reverse(right_reach.begin(), right_reach.end());
        vars.clusters = next_clusters;
        vars.left_reaches = left_reach;
        vars.right_reaches = right_reach;
        if (vars.clusters[vars.clusters.size() - 1] != vars.n) return VCL_ERR_ARG;
    }
    *out = dump(vars);
    return VCL_OK;
}

// cluster.cpp:404-553 (one contig)
int vco_supercluster(const vcl_hap haps[4], const vcl_clusters *const clusters[4], int32_t max_supercluster_size,
                     vcl_superclusters **out) {
    HapVars vars[4];
    int nvars = 0;
    for (int i = 0; i < 4; i++) {
        load(&haps[i], vars[i]);
        nvars += vars[i].n;
        if (clusters[i]->n > 0) {
            vars[i].clusters.assign(clusters[i]->var_beg, clusters[i]->var_beg + clusters[i]->n + 1);
            vars[i].left_reaches.assign(clusters[i]->left_reach, clusters[i]->left_reach + clusters[i]->n + 1);
            vars[i].right_reaches.assign(clusters[i]->right_reach, clusters[i]->right_reach + clusters[i]->n + 1);
        }
    }
    std::vector<int> sc[4], begs, ends;
    int n_oversize = 0, n_unsplittable = 0;
    auto add_supercluster = [&](const std::vector<int> &brks, int beg, int end) {   // cluster.cpp:18-29
        for (int i = 0; i < 4; i++) sc[i].push_back(brks[i]);
        begs.push_back(beg);
        ends.push_back(end);
    };
    std::vector<int> brks = {0, 0, 0, 0};
    if (nvars) {
        while (true) {
            std::vector<int> next_brks = brks;
            std::vector<int> lefts(4, INT_MAX);
            for (int i = 0; i < 4; i++)
                if (brks[i] < int(vars[i].clusters.size()) - 1) lefts[i] = vars[i].left_reaches[next_brks[i]];
            int idx = std::distance(lefts.begin(), std::min_element(lefts.begin(), lefts.end()));
            if (lefts[idx] == INT_MAX) break;
            int curr_right = vars[idx].right_reaches[next_brks[idx]];
            next_brks[idx]++;
            lefts[idx] = next_brks[idx] < int(vars[idx].clusters.size()) - 1 ? vars[idx].left_reaches[next_brks[idx]] : INT_MAX;
            bool just_active = true;
            while (just_active) {
                just_active = false;
                for (int i = 0; i < 4; i++) {
                    while (lefts[i] <= curr_right) {
                        curr_right = std::max(curr_right, vars[i].right_reaches[next_brks[i]]);
                        next_brks[i]++;
                        lefts[i] = next_brks[i] < int(vars[i].clusters.size()) - 1 ? vars[i].left_reaches[next_brks[i]] : INT_MAX;
                        just_active = true;
                    }
                }
            }
            std::vector<int> poss = range(vars, brks, next_brks);
            if (poss[1] - poss[0] > max_supercluster_size) {
                n_oversize++;
                std::vector<std::vector<int>> all_brks = split_large(vars, brks, next_brks, max_supercluster_size, n_unsplittable);
                for (int b = 0; b < int(all_brks.size()) - 1; b++) {
                    brks = all_brks[b];
                    next_brks = all_brks[b + 1];
                    poss = range(vars, brks, next_brks);
                    add_supercluster(brks, poss[0], poss[1]);
                }
            } else {
                add_supercluster(brks, poss[0], poss[1]);
            }
            brks = next_brks;
        }
    }
    vcl_superclusters *s = static_cast<vcl_superclusters *>(calloc(1, sizeof(vcl_superclusters)));
    s->n = int(begs.size());
    s->n_oversize = n_oversize;
    s->n_unsplittable = n_unsplittable;
    for (int i = 0; i < 4; i++) {
        sc[i].push_back(brks[i]);   // sentinel, cluster.cpp:541-544
        s->brk[i] = static_cast<int32_t *>(malloc(sc[i].size() * 4));
        for (size_t k = 0; k < sc[i].size(); k++) s->brk[i][k] = sc[i][k];
        s->clusters[i] = dump(vars[i]);
    }
    s->beg = static_cast<int32_t *>(malloc((begs.size() + 1) * 4));
    s->end = static_cast<int32_t *>(malloc((ends.size() + 1) * 4));
    for (size_t k = 0; k < begs.size(); k++) { s->beg[k] = begs[k]; s->end[k] = ends[k]; }
    *out = s;
    return VCL_OK;
}

}  // extern "C"
