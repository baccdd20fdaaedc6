// synth.cpp -- deterministic synthetic supercluster workloads (host only).
//
// Implements the generator specified in SURVEY.md 8(d) for BASELINE.json configs
// 2/4 (WGS-like log-normal spans) and 5 (log-uniform 32..16384 stress): per
// supercluster an i.i.d. ACGT (or tandem-repeat) reference span, a diploid
// "query" genotype of SNP / indel sites, and a "truth" genotype derived from it
// by keeping / dropping / perturbing each site.  The PRNG is an explicit
// splitmix64 stream per supercluster, so a workload is reproducible from
// (seed, parameters) on any host and can be generated rank-locally.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/vcfdist_pr.h"

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
    int below(int n) { return int(next() % uint64_t(n)); }
    double normal() {  // Box-Muller, one value
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
    int poisson(double lam) {  // Knuth; lam <= ~700
        const double L = std::exp(-lam);
        int k = 0;
        double p = 1.0;
        do { k++; p *= uni(); } while (p > L);
        return k - 1;
    }
    int geometric(double mean) {  // >= 1
        const double p = 1.0 / mean;
        int k = 1;
        while (uni() > p && k < 1000000) k++;
        return k;
    }
};

const char BASES[4] = {'A', 'C', 'G', 'T'};

struct Site {
    int pos;            // relative to span start
    uint8_t type;
    std::string ref, alt;
    int hapmask;        // bit0 = hap1, bit1 = hap2
    float qual;
    int rlen() const { return type == VPR_TYPE_INS ? 0 : int(ref.size()); }
};

struct HapVars {
    std::vector<int64_t> var_off{0};
    std::vector<int32_t> pos;
    std::vector<uint8_t> type;
    std::vector<float> qual;
    std::vector<int64_t> ref_off, alt_off;
    std::vector<int32_t> ref_len, alt_len;
    std::vector<uint8_t> pool;
    void add(int32_t p, const Site &s) {
        pos.push_back(p);
        type.push_back(s.type);
        qual.push_back(s.qual);
        ref_off.push_back(pool.size());
        ref_len.push_back(int32_t(s.ref.size()));
        pool.insert(pool.end(), s.ref.begin(), s.ref.end());
        alt_off.push_back(pool.size());
        alt_len.push_back(int32_t(s.alt.size()));
        pool.insert(pool.end(), s.alt.begin(), s.alt.end());
    }
};

}  // namespace

struct vpr_synth {
    vpr_synth_params p;
    std::vector<int64_t> ctg_off;
    std::vector<uint8_t> ctg_seq;
    std::vector<int32_t> sc_ctg, sc_beg, sc_end;
    HapVars hv[VPR_HAPS];
    vpr_variants view;
};

extern "C" {

void vpr_synth_default_params(vpr_synth_params *p) {
    memset(p, 0, sizeof(*p));
    p->seed = 0x5eed;
    p->n_sc = 1000;
    p->len_mode = 0;
    p->len_a = 32;
    p->len_b = 16384;
    p->len_min = 8;
    p->len_max = 16384;
    p->p_repeat = 0.2;
    p->var_per_base = 1.0 / 200.0;
    p->p_snp = 0.8;
    p->indel_mean = 3.0;
    p->p_hom = 0.7;
    p->p_keep = 0.9;
    p->p_drop = 0.05;
    p->max_qual = 60;
    p->p_sv = 0.0;
    p->sv_min = 50;
    p->sv_max = 10000;
}

int vpr_synth_create(const vpr_synth_params *pp, vpr_synth **out) {
    if (!pp || !out || pp->n_sc < 0) return VPR_ERR_ARG;
    vpr_synth *S = new (std::nothrow) vpr_synth();
    if (!S) return VPR_ERR_NOMEM;
    S->p = *pp;
    const vpr_synth_params &P = S->p;
    // one contig holds all spans back to back; contig coordinates are int32 in the
    // reference's data model, so start a new contig before 2^30 bases
    S->ctg_off.push_back(0);
    int32_t cur_ctg = 0;
    int64_t ctg_start = 0;

    for (int sc = 0; sc < P.n_sc; sc++) {
        Rng rng(P.seed * 0x9e3779b97f4a7c15ULL + uint64_t(sc) * 0xd1342543de82ef95ULL + 1);
        // joint small-variant + SV workloads (BASELINE configs[3]): a stream of its own decides whether this supercluster
        // carries one SV-sized indel, so that the small-variant mix of a seed is the same with and without SVs
        Rng rsv(P.seed * 0xc2b2ae3d27d4eb4fULL + uint64_t(sc) * 0x9e3779b97f4a7c15ULL + 7);
        int sv_len = 0;
        bool sv_ins = false;
        if (P.p_sv > 0 && rsv.uni() < P.p_sv) {
            const double lo = std::log(double(std::max(P.sv_min, 1))), hi = std::log(double(std::max(P.sv_max, P.sv_min)));
            sv_len = std::max(1, int(std::exp(lo + rsv.uni() * (hi - lo)) + 0.5));
            sv_ins = rsv.below(2) != 0;
        }
        // --- span length
        double Lf;
        if (P.len_mode == 0) Lf = std::exp(std::log(P.len_a) + rng.uni() * (std::log(P.len_b) - std::log(P.len_a)));
        else if (P.len_mode == 1) Lf = P.len_a * std::exp(P.len_b * rng.normal());
        else Lf = P.len_a;
        int L = int(Lf + 0.5);
        L = std::max(std::max(P.len_min, 4), std::min(L, P.len_max));
        if (sv_len && !sv_ins) {        // a deletion lies inside the span: the small-variant span + the deleted bases
            L = std::max(8, std::min(L + sv_len, P.len_max));
            sv_len = std::min(sv_len, L - 4);
        } else if (sv_len) {
            L = std::max(L, 8);
        }

        if (int64_t(S->ctg_seq.size()) - ctg_start + L > (int64_t(1) << 30)) {
            S->ctg_off.push_back(S->ctg_seq.size());
            ctg_start = S->ctg_seq.size();
            cur_ctg++;
        }
        const int64_t span0 = int64_t(S->ctg_seq.size()) - ctg_start;

        // --- reference span
        std::string ref(L, 'A');
        int unit = 0;
        if (rng.uni() < P.p_repeat) {
            unit = 1 + rng.below(6);
            char u[6];
            for (int k = 0; k < unit; k++) u[k] = BASES[rng.below(4)];
            for (int k = 0; k < L; k++) ref[k] = u[k % unit];
            // sprinkle a few point differences so the repeat is imperfect
            const int nmut = L / 50;
            for (int k = 0; k < nmut; k++) ref[rng.below(L)] = BASES[rng.below(4)];
        } else {
            for (int k = 0; k < L; k++) ref[k] = BASES[rng.below(4)];
        }

        // --- query sites
        const double lam = std::max(1.0, L * P.var_per_base);
        int nsite = rng.poisson(std::min(lam, 600.0));
        std::vector<Site> sites;
        for (int k = 0; k < nsite; k++) {
            Site s;
            if (L < 4) break;
            s.pos = 1 + rng.below(L - 2);
            s.qual = float(1 + rng.below(std::max(1, P.max_qual)));
            const double hz = rng.uni();
            s.hapmask = (hz < P.p_hom) ? 3 : (rng.below(2) ? 1 : 2);
            if (rng.uni() < P.p_snp) {
                s.type = VPR_TYPE_SUB;
                s.ref = std::string(1, ref[s.pos]);
                char a;
                do { a = BASES[rng.below(4)]; } while (a == ref[s.pos]);
                s.alt = std::string(1, a);
            } else {
                int len = std::min(rng.geometric(P.indel_mean), std::max(1, L / 4));
                if (unit && rng.uni() < 0.7) len = std::max(unit, (len / unit) * unit);
                if (rng.below(2)) {
                    s.type = VPR_TYPE_INS;
                    s.alt.resize(len);
                    for (int j = 0; j < len; j++)
                        s.alt[j] = unit ? ref[(s.pos + j) % L] : BASES[rng.below(4)];
                } else {
                    s.type = VPR_TYPE_DEL;
                    if (s.pos + len > L - 1) len = L - 1 - s.pos;
                    if (len < 1) continue;
                    s.ref = ref.substr(s.pos, len);
                }
            }
            sites.push_back(s);
        }
        if (sv_len > 0) {               // the SV site: small sites whose footprint touches it give way
            Site v;
            v.qual = float(1 + rsv.below(std::max(1, P.max_qual)));
            v.hapmask = (rsv.uni() < P.p_hom) ? 3 : (rsv.below(2) ? 1 : 2);
            if (sv_ins) {
                v.type = VPR_TYPE_INS;
                v.pos = 1 + rsv.below(L - 2);
                v.alt.resize(sv_len);
                for (int j = 0; j < sv_len; j++) v.alt[j] = unit ? ref[(v.pos + j) % L] : BASES[rsv.below(4)];
            } else {
                v.type = VPR_TYPE_DEL;
                v.pos = 1 + rsv.below(L - 2 - sv_len);
                v.ref = ref.substr(v.pos, sv_len);
            }
            std::vector<Site> keep;
            for (const Site &s : sites)
                if (!(s.pos + s.rlen() + 1 > v.pos && s.pos < v.pos + v.rlen() + 1)) keep.push_back(s);
            keep.push_back(v);
            sites.swap(keep);
        }
        std::stable_sort(sites.begin(), sites.end(), [](const Site &a, const Site &b) { return a.pos < b.pos; });
        // drop overlapping sites (footprint = [pos, pos+rlen], one clear base after)
        std::vector<Site> qsites;
        int next_free = 1;
        for (const Site &s : sites) {
            if (s.pos < next_free) continue;
            qsites.push_back(s);
            next_free = s.pos + s.rlen() + 1;
        }

        // --- truth sites: keep / drop / perturb
        std::vector<Site> tsites;
        for (size_t k = 0; k < qsites.size(); k++) {
            const double u = rng.uni();
            Site t = qsites[k];
            if (u < P.p_keep) {
                // kept as is
            } else if (u < P.p_keep + P.p_drop) {
                continue;
            } else {
                const int next_pos = (k + 1 < qsites.size()) ? qsites[k + 1].pos : L - 1;
                const int prev_end = tsites.empty() ? 1 : tsites.back().pos + tsites.back().rlen() + 1;
                if (t.type == VPR_TYPE_SUB) {
                    char a;
                    do { a = BASES[rng.below(4)]; } while (a == ref[t.pos]);
                    t.alt = std::string(1, a);
                } else if (unit && rng.below(2)) {  // shift by one repeat unit (equivalent placement)
                    const int np = t.pos + (rng.below(2) ? unit : -unit);
                    if (np >= prev_end && np + t.rlen() < next_pos && np >= 1) {
                        t.pos = np;
                        if (t.type == VPR_TYPE_DEL) t.ref = ref.substr(t.pos, t.ref.size());
                    }
                } else if (t.type == VPR_TYPE_INS) {
                    if (rng.below(2) && t.alt.size() > 1) t.alt.pop_back();
                    else t.alt.push_back(BASES[rng.below(4)]);
                } else {  // DEL: change the length by one
                    int len = int(t.ref.size()) + (rng.below(2) ? 1 : -1);
                    if (len >= 1 && t.pos + len < next_pos) t.ref = ref.substr(t.pos, len);
                }
                if (rng.below(4) == 0) t.hapmask = (t.hapmask == 3) ? (1 + rng.below(2)) : 3;  // genotype error
            }
            tsites.push_back(t);
        }

        // --- emit
        S->sc_ctg.push_back(cur_ctg);
        S->sc_beg.push_back(int32_t(span0));
        S->sc_end.push_back(int32_t(span0 + L - 1));
        S->ctg_seq.insert(S->ctg_seq.end(), ref.begin(), ref.end());
        for (int h = 0; h < 2; h++) {
            for (const Site &s : qsites)
                if (s.hapmask & (1 << h)) S->hv[h].add(int32_t(span0 + s.pos), s);
            S->hv[h].var_off.push_back(S->hv[h].pos.size());
            for (const Site &s : tsites)
                if (s.hapmask & (1 << h)) S->hv[2 + h].add(int32_t(span0 + s.pos), s);
            S->hv[2 + h].var_off.push_back(S->hv[2 + h].pos.size());
        }
    }
    S->ctg_off.push_back(S->ctg_seq.size());

    vpr_variants &v = S->view;
    memset(&v, 0, sizeof(v));
    v.n_sc = P.n_sc;
    v.n_ctg = int32_t(S->ctg_off.size()) - 1;
    v.ctg_off = S->ctg_off.data();
    v.ctg_seq = S->ctg_seq.data();
    v.sc_ctg = S->sc_ctg.data();
    v.sc_beg = S->sc_beg.data();
    v.sc_end = S->sc_end.data();
    for (int h = 0; h < VPR_HAPS; h++) {
        v.var_off[h] = S->hv[h].var_off.data();
        v.var_pos[h] = S->hv[h].pos.data();
        v.var_type[h] = S->hv[h].type.data();
        v.var_qual[h] = S->hv[h].qual.data();
        v.var_ref_off[h] = S->hv[h].ref_off.data();
        v.var_alt_off[h] = S->hv[h].alt_off.data();
        v.var_ref_len[h] = S->hv[h].ref_len.data();
        v.var_alt_len[h] = S->hv[h].alt_len.data();
        v.allele_pool[h] = S->hv[h].pool.data();
    }
    *out = S;
    return VPR_OK;
}

const vpr_variants *vpr_synth_variants(const vpr_synth *s) { return s ? &s->view : nullptr; }
void vpr_synth_destroy(vpr_synth *s) { delete s; }

}  // extern "C"
