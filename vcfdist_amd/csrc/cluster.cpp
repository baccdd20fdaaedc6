// cluster.cpp -- host side of include/vcfdist_cluster.h: distance-based dependency clustering and
// superclustering, the step that defines the work units of the precision/recall path.
//
// Superclustering is an interval-union sweep: every cluster of the four haplotypes is an interval
// [left_reach, right_reach]; a supercluster is a maximal chain of intervals in which each next interval starts
// at or before the running maximum of the right reaches.  The four per-hap cluster lists are consumed through a
// four-way merge on the next cluster's left reach, which is what the reference's loop does
// (superclusterData::supercluster, src/cluster.cpp:404-553), so the membership and the emitted cluster indices are
// identical.  Oversized superclusters are cut at the inter-variant gap with the best
// gap / log-size-reduction score (cluster.cpp:738-808) and the affected clusters are split in place
// (cluster.cpp:655-706).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/vcfdist_cluster.h"

namespace {

const int32_t INF = VCL_SENTINEL;

// growable copy of one hap's cluster table
struct Table {
    std::vector<int32_t> var_beg, left, right;   // n + 1 entries each (sentinel last), or empty
    int32_t n() const { return var_beg.empty() ? 0 : int32_t(var_beg.size()) - 1; }
};

vcl_clusters *export_table(const Table &t) {
    vcl_clusters *c = static_cast<vcl_clusters *>(calloc(1, sizeof(vcl_clusters)));
    if (!c) return nullptr;
    c->n = t.n();
    const size_t m = t.var_beg.size();
    c->var_beg = static_cast<int32_t *>(malloc(std::max<size_t>(m, 1) * 4));
    c->left_reach = static_cast<int32_t *>(malloc(std::max<size_t>(m, 1) * 4));
    c->right_reach = static_cast<int32_t *>(malloc(std::max<size_t>(m, 1) * 4));
    if (m) {
        memcpy(c->var_beg, t.var_beg.data(), m * 4);
        memcpy(c->left_reach, t.left.data(), m * 4);
        memcpy(c->right_reach, t.right.data(), m * 4);
    }
    return c;
}

bool hap_ok(const vcl_hap *h) {
    if (!h || h->n_var < 0) return false;
    if (h->n_var && (!h->pos || !h->rlen)) return false;
    for (int32_t v = 1; v < h->n_var; v++)
        if (h->pos[v] < h->pos[v - 1]) return false;
    return true;
}

// positions spanned by clusters [a_i, b_i) of every hap: one base left of the first variant, one right of the
// last variant's end (get_supercluster_range, cluster.cpp:563-594)
struct Span { int32_t beg, end; };
Span span_of(const vcl_hap haps[4], const Table tab[4], const int32_t a[4], const int32_t b[4]) {
    Span s{INF, -1};
    for (int i = 0; i < 4; i++) {
        if (b[i] == a[i]) continue;
        const int32_t first = tab[i].var_beg[a[i]], last = tab[i].var_beg[b[i]] - 1;
        s.beg = std::min(s.beg, haps[i].pos[first] - 1);
        s.end = std::max(s.end, haps[i].pos[last] + haps[i].rlen[last] + 1);
    }
    return s;
}

struct Cut { int32_t v[4]; bool found; };

// best place to cut the variants [vb_i, ve_i): walk all variants in position order (ties: lowest hap slot) and
// score the gap behind each one, cluster.cpp:738-808
Cut best_cut(const vcl_hap haps[4], const Table tab[4], const int32_t a[4], const int32_t b[4]) {
    Cut best{{0, 0, 0, 0}, false};
    const Span sp = span_of(haps, tab, a, b);
    const int32_t size = sp.end - sp.beg;
    int32_t cur[4] = {0, 0, 0, 0}, ve[4] = {0, 0, 0, 0};
    int64_t total = 0;
    for (int i = 0; i < 4; i++)
        if (!tab[i].var_beg.empty()) {
            cur[i] = tab[i].var_beg[a[i]];
            ve[i] = tab[i].var_beg[b[i]];
            total += ve[i] - cur[i];
        }
    if (total < 2) return best;
    auto next = [&](int &hap, int32_t &start, int32_t &stop) {
        hap = -1; start = INF; stop = INF;
        for (int i = 0; i < 4; i++)
            if (cur[i] < ve[i] && haps[i].pos[cur[i]] < start) {
                start = haps[i].pos[cur[i]];
                stop = start + haps[i].rlen[cur[i]];
                hap = i;
            }
    };
    int hc, hn;
    int32_t cs, ce, ns, ne;
    next(hc, cs, ce);
    cur[hc]++;
    next(hn, ns, ne);
    double best_score = 0;
    while (hn >= 0) {
        const int32_t gap = std::max(0, ns - ce);
        const int32_t mid = ce + gap / 2;
        const double frac = std::max(double(mid - sp.beg) / size, double(sp.end - mid) / size);
        const double halvings = -1 / log2(frac);
        const double score = gap / halvings;
        if (score > best_score) {
            best_score = score;
            best.found = true;
            for (int i = 0; i < 4; i++) best.v[i] = cur[i];
        }
        hc = hn; cs = ns; ce = ne;
        cur[hc]++;
        next(hn, ns, ne);
    }
    (void)cs;
    return best;
}

}  // namespace

extern "C" {

int vcl_simple_cluster(const vcl_hap *hap, int size_mode, int32_t cluster_min_gap, int32_t reach_min_gap,
                       vcl_clusters **out) {
    if (!out) return VCL_ERR_ARG;
    *out = nullptr;
    if (!hap_ok(hap)) return VCL_ERR_ARG;
    Table t;
    const int32_t n = hap->n_var;
    if (n > 0) {   // a hap without variants keeps empty tables (cluster.cpp:838)
        // per-variant reaches (+ sentinel), cluster.cpp:853-886
        std::vector<int32_t> L(size_t(n) + 1), R(size_t(n) + 1);
        for (int32_t v = 0; v < n; v++) {
            int32_t sz = 0;
            if (size_mode) {
                if (!hap->type || !hap->ref_len || !hap->alt_len) return VCL_ERR_ARG;
                switch (hap->type[v]) {
                    case 1: sz = 1; break;
                    case 2: sz = hap->alt_len[v]; break;
                    case 3: sz = hap->ref_len[v]; break;
                    default: return VCL_ERR_TYPE;
                }
            }
            const int32_t reach = std::max(cluster_min_gap, sz);
            L[v] = hap->pos[v] - reach;
            R[v] = hap->pos[v] + hap->rlen[v] + reach;
        }
        L[n] = R[n] = INF;
        // pass 1, left to right: a group absorbs its right neighbour while the farthest right reach seen so far
        // (+ reach_min_gap) touches the neighbour's own left reach (cluster.cpp:889-906)
        struct Group { int32_t first, lo, hi; };
        std::vector<Group> g1;
        for (int32_t v = 0; v <= n;) {
            Group g{v, L[v], R[v]};
            int32_t w = v + 1;
            while (w <= n && int64_t(g.hi) + reach_min_gap >= L[w]) {
                g.hi = std::max(g.hi, R[w]);
                g.lo = std::min(g.lo, L[w]);
                w++;
            }
            g1.push_back(g);
            v = w;
        }
        // pass 2, right to left over the groups: merge with the left neighbour while the group's smallest left
        // reach touches the neighbour's right reach (cluster.cpp:911-929)
        std::vector<Group> g2;
        for (int64_t k = int64_t(g1.size()) - 1; k >= 0;) {
            Group g = g1[size_t(k)];
            while (k > 0 && g.lo <= int64_t(g1[size_t(k - 1)].hi) + reach_min_gap) {
                k--;
                g.lo = std::min(g.lo, g1[size_t(k)].lo);
                g.hi = std::max(g.hi, g1[size_t(k)].hi);
                g.first = g1[size_t(k)].first;
            }
            g2.push_back(g);
            k--;
        }
        for (auto it = g2.rbegin(); it != g2.rend(); ++it) {
            t.var_beg.push_back(it->first);
            t.left.push_back(it->lo);
            t.right.push_back(it->hi);
        }
        if (t.var_beg.back() != n) return VCL_ERR_ARG;   // "Mismatch between original and clustered variant count"
    }
    *out = export_table(t);
    return *out ? VCL_OK : VCL_ERR_ARG;
}

void vcl_clusters_free(vcl_clusters *c) {
    if (!c) return;
    free(c->var_beg);
    free(c->left_reach);
    free(c->right_reach);
    free(c);
}

int vcl_supercluster(const vcl_hap haps[4], const vcl_clusters *const clusters[4], int32_t max_supercluster_size,
                     vcl_superclusters **out) {
    if (!out) return VCL_ERR_ARG;
    *out = nullptr;
    if (!haps || !clusters) return VCL_ERR_ARG;
    Table tab[4];
    int64_t nvars = 0;
    for (int i = 0; i < 4; i++) {
        if (!hap_ok(&haps[i]) || !clusters[i] || clusters[i]->n < 0) return VCL_ERR_ARG;
        nvars += haps[i].n_var;
        if (clusters[i]->n > 0) {
            const size_t m = size_t(clusters[i]->n) + 1;
            tab[i].var_beg.assign(clusters[i]->var_beg, clusters[i]->var_beg + m);
            tab[i].left.assign(clusters[i]->left_reach, clusters[i]->left_reach + m);
            tab[i].right.assign(clusters[i]->right_reach, clusters[i]->right_reach + m);
            if (tab[i].var_beg.front() != 0 || tab[i].var_beg.back() != haps[i].n_var) return VCL_ERR_ARG;
            for (size_t k = 1; k < m; k++)
                if (tab[i].var_beg[k] <= tab[i].var_beg[k - 1]) return VCL_ERR_ARG;
        } else if (haps[i].n_var != 0) {
            return VCL_ERR_ARG;
        }
    }
    std::vector<int32_t> brk[4], begs, ends;
    int32_t n_oversize = 0, n_unsplittable = 0;
    auto emit = [&](const int32_t a[4], const Span &sp) {
        for (int i = 0; i < 4; i++) brk[i].push_back(a[i]);
        begs.push_back(sp.beg);
        ends.push_back(sp.end);
    };
    int32_t at[4] = {0, 0, 0, 0};   // first cluster of the supercluster being built, per hap
    if (nvars > 0) {                // contigs without variants get no superclusters at all (cluster.cpp:420-424)
        while (true) {
            int32_t to[4] = {at[0], at[1], at[2], at[3]};
            auto left_of = [&](int i) { return to[i] < tab[i].n() ? tab[i].left[size_t(to[i])] : INF; };
            // seed: the hap whose next cluster reaches farthest left (ties: lowest slot)
            int seed = 0;
            for (int i = 1; i < 4; i++)
                if (left_of(i) < left_of(seed)) seed = i;
            if (left_of(seed) == INF) break;
            int32_t reach = tab[seed].right[size_t(to[seed])];
            to[seed]++;
            // absorb every cluster that starts at or before the running right reach
            for (bool grew = true; grew;) {
                grew = false;
                for (int i = 0; i < 4; i++)
                    while (left_of(i) <= reach) {
                        reach = std::max(reach, tab[i].right[size_t(to[i])]);
                        to[i]++;
                        grew = true;
                    }
            }
            const Span sp = span_of(haps, tab, at, to);
            if (sp.end - sp.beg > max_supercluster_size) {
                // cut until every piece fits (split_large_supercluster, cluster.cpp:601-648)
                n_oversize++;
                std::vector<std::vector<int32_t>> cuts = {{at[0], at[1], at[2], at[3]}, {to[0], to[1], to[2], to[3]}};
                for (bool again = true; again;) {
                    again = false;
                    std::vector<std::vector<int32_t>> nextc;
                    for (size_t k = 0; k + 1 < cuts.size(); k++) {
                        const Span ps = span_of(haps, tab, cuts[k].data(), cuts[k + 1].data());
                        nextc.push_back(cuts[k]);
                        if (ps.end - ps.beg <= max_supercluster_size) continue;
                        again = true;
                        const Cut c = best_cut(haps, tab, cuts[k].data(), cuts[k + 1].data());
                        if (!c.found) { n_unsplittable++; again = false; continue; }
                        // turn the variant cut into cluster indices, splitting clusters where needed (cluster.cpp:655-706)
                        std::vector<int32_t> ci(4, 0);
                        for (int i = 0; i < 4; i++) {
                            if (tab[i].var_beg.empty()) continue;
                            const auto it = std::lower_bound(tab[i].var_beg.begin(), tab[i].var_beg.end(), c.v[i]);
                            const int32_t idx = int32_t(it - tab[i].var_beg.begin());
                            ci[i] = idx;
                            if (*it == c.v[i]) continue;   // already a cluster boundary
                            const int32_t p = haps[i].pos[c.v[i]];
                            const int32_t old_right = tab[i].right[size_t(idx) - 1];
                            tab[i].right[size_t(idx) - 1] = p;
                            tab[i].left.insert(tab[i].left.begin() + idx, p);
                            tab[i].right.insert(tab[i].right.begin() + idx, old_right);
                            tab[i].var_beg.insert(tab[i].var_beg.begin() + idx, c.v[i]);
                            for (size_t j = k + 1; j < cuts.size(); j++) cuts[j][size_t(i)]++;
                        }
                        nextc.push_back(ci);
                    }
                    nextc.push_back(cuts.back());
                    cuts.swap(nextc);
                }
                for (size_t k = 0; k + 1 < cuts.size(); k++)
                    emit(cuts[k].data(), span_of(haps, tab, cuts[k].data(), cuts[k + 1].data()));
                for (int i = 0; i < 4; i++) at[i] = cuts.back()[size_t(i)];
            } else {
                emit(at, sp);
                for (int i = 0; i < 4; i++) at[i] = to[i];
            }
        }
    }
    vcl_superclusters *s = static_cast<vcl_superclusters *>(calloc(1, sizeof(vcl_superclusters)));
    if (!s) return VCL_ERR_ARG;
    s->n = int32_t(begs.size());
    s->n_oversize = n_oversize;
    s->n_unsplittable = n_unsplittable;
    for (int i = 0; i < 4; i++) {
        brk[i].push_back(at[i]);   // the reference's sentinel entry (cluster.cpp:541-544)
        s->brk[i] = static_cast<int32_t *>(malloc(brk[i].size() * 4));
        memcpy(s->brk[i], brk[i].data(), brk[i].size() * 4);
        s->clusters[i] = export_table(tab[i]);
    }
    s->beg = static_cast<int32_t *>(malloc(std::max<size_t>(begs.size(), 1) * 4));
    s->end = static_cast<int32_t *>(malloc(std::max<size_t>(ends.size(), 1) * 4));
    if (!begs.empty()) {
        memcpy(s->beg, begs.data(), begs.size() * 4);
        memcpy(s->end, ends.data(), ends.size() * 4);
    }
    *out = s;
    return VCL_OK;
}

void vcl_superclusters_free(vcl_superclusters *s) {
    if (!s) return;
    for (int i = 0; i < 4; i++) {
        free(s->brk[i]);
        vcl_clusters_free(s->clusters[i]);
    }
    free(s->beg);
    free(s->end);
    free(s);
}

int64_t vcl_supercluster_cells(const vcl_hap haps[4], const vcl_superclusters *s, int32_t k) {
    if (!haps || !s || k < 0 || k >= s->n) return -1;
    int64_t len[2] = {0, 0};   // longest hap of each callset after applying its variants (cluster.cpp:54-91)
    for (int i = 0; i < 4; i++) {
        const vcl_clusters *c = s->clusters[i];
        if (!c || c->n == 0) continue;
        int64_t l = int64_t(s->end[k]) - s->beg[k];
        for (int32_t v = c->var_beg[s->brk[i][k]]; v < c->var_beg[s->brk[i][k + 1]]; v++)
            l += int64_t(haps[i].alt_len ? haps[i].alt_len[v] : 0) - int64_t(haps[i].ref_len ? haps[i].ref_len[v] : 0);
        len[i >> 1] = std::max(len[i >> 1], l);
    }
    return len[0] * len[1];
}

// the same for every supercluster in one call: out[s->n]
int vcl_supercluster_cells_all(const vcl_hap haps[4], const vcl_superclusters *s, int64_t *out) {
    if (!haps || !s || !out) return VCL_ERR_ARG;
    for (int32_t k = 0; k < s->n; k++) out[k] = vcl_supercluster_cells(haps, s, k);
    return VCL_OK;
}

}  // extern "C"
