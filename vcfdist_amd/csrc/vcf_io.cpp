// vcf_io.cpp -- input formats behind include/vcfdist_io.h: VCF records -> per-contig / per-hap variant columns,
// BED regions, FASTA contigs.  Text or gzip / bgzip through zlib (gzgets reads both); no htslib.
//
// The record logic follows variantData::variantData (src/variant.cpp:556-875) step by step -- FILTER, QUAL, GQ, GT,
// PS, then per haplotype: allele selection, unphased / spanning-deletion skips, prefix / suffix trimming and typing,
// BED test on the original span, size filter, overlap filter, CPX -> INS + DEL -- but works on the text fields of a
// line instead of htslib's unpacked record.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/vcfdist_io.h"

namespace {

thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

enum { T_REF = 0, T_SUB = 1, T_INS = 2, T_DEL = 3, T_CPX = 4 };

// whole lines of a (possibly compressed) text file
struct LineReader {
    gzFile f = nullptr;
    std::vector<char> buf;
    bool open(const char *path) { f = gzopen(path, "rb"); if (f) gzbuffer(f, 1 << 20); buf.resize(1 << 16); return f != nullptr; }
    bool next(std::string &line) {
        line.clear();
        while (true) {
            if (!gzgets(f, buf.data(), int(buf.size()))) return !line.empty();
            line += buf.data();
            if (!line.empty() && line.back() == '\n') { line.pop_back(); if (!line.empty() && line.back() == '\r') line.pop_back(); return true; }
        }
    }
    ~LineReader() { if (f) gzclose(f); }
};

std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out;
    size_t a = 0;
    while (true) {
        const size_t b = s.find(sep, a);
        if (b == std::string::npos) { out.push_back(s.substr(a)); break; }
        out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return out;
}

char *dup_str(const std::string &s) {
    char *p = static_cast<char *>(malloc(s.size() + 1));
    memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

struct HapBuild {
    std::vector<int32_t> pos, rlen, phase_set, ref_len, alt_len;
    std::vector<uint8_t> type, orig_gt;
    std::vector<float> var_qual, gt_qual;
    std::vector<int64_t> ref_off, alt_off;
    std::string pool;
    void add(int p, int rl, int t, const std::string &ref, const std::string &alt, int gt, float gq, float vq, int ps) {
        pos.push_back(p); rlen.push_back(rl); type.push_back(uint8_t(t)); orig_gt.push_back(uint8_t(gt));
        gt_qual.push_back(gq); var_qual.push_back(vq); phase_set.push_back(ps);
        ref_off.push_back(int64_t(pool.size())); ref_len.push_back(int32_t(ref.size())); pool += ref;
        alt_off.push_back(int64_t(pool.size())); alt_len.push_back(int32_t(alt.size())); pool += alt;
    }
};

template <typename T>
T *dup_vec(const std::vector<T> &v) {
    T *p = static_cast<T *>(malloc(std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

}  // namespace

struct vio_bed {
    std::map<std::string, std::pair<std::vector<int32_t>, std::vector<int32_t>>> regions;   // contig -> (starts, stops)
};

extern "C" {

const char *vio_last_error(void) { return g_err.c_str(); }

int vio_read_bed(const char *path, vio_bed **out) {
    if (!path || !out) return VIO_ERR_ARG;
    *out = nullptr;
    LineReader R;
    if (!R.open(path)) return fail(VIO_ERR_OPEN, "Failed to open BED file '%s'", path);
    vio_bed *b = new vio_bed();
    std::string line;
    while (R.next(line)) {
        if (line.empty()) continue;
        const std::vector<std::string> f = split(line, '\t');
        if (f.size() < 3) { delete b; return fail(VIO_ERR_FORMAT, "BED line with fewer than 3 fields: '%s'", line.c_str()); }
        auto &r = b->regions[f[0]];
        r.first.push_back(atoi(f[1].c_str()));
        r.second.push_back(atoi(f[2].c_str()));
    }
    // bedData::check (bed.cpp:36-71, called from globals.cpp:118): the reference stops on empty, unsorted or overlapping
    // regions (vio_bed_contains bisects starts and stops, so it relies on exactly that) and warns about abutting ones
    for (const auto &kv : b->regions) {
        const std::string &ctg = kv.first;
        const std::vector<int32_t> &st = kv.second.first, &sp = kv.second.second;
        for (size_t j = 0; j < st.size(); j++) {
            if (sp[j] < st[j]) { const int rc = fail(VIO_ERR_FORMAT, "BED region %s:%d-%d stop precedes start.", ctg.c_str(), st[j], sp[j]); delete b; return rc; }
            if (sp[j] == st[j]) { const int rc = fail(VIO_ERR_FORMAT, "BED region %s:%d-%d length zero.", ctg.c_str(), st[j], sp[j]); delete b; return rc; }
            if (j) {
                if (sp[j] < st[j - 1]) {
                    const int rc = fail(VIO_ERR_FORMAT, "BED is unsorted; region %s:%d-%d precedes %s:%d-%d.", ctg.c_str(), st[j - 1], sp[j - 1],
                                        ctg.c_str(), st[j], sp[j]);
                    delete b; return rc;
                }
                if (st[j] < sp[j - 1]) {
                    const int rc = fail(VIO_ERR_FORMAT, "BED overlap detected: regions %s:%d-%d and %s:%d-%d.", ctg.c_str(), st[j - 1], sp[j - 1],
                                        ctg.c_str(), st[j], sp[j]);
                    delete b; return rc;
                }
                if (sp[j - 1] == st[j])
                    fprintf(stderr, "[WARN ] BED regions %s:%d-%d and %s:%d-%d should be merged.\n", ctg.c_str(), st[j - 1], sp[j - 1],
                            ctg.c_str(), st[j], sp[j]);
            }
        }
    }
    *out = b;
    return VIO_OK;
}

void vio_bed_free(vio_bed *b) { delete b; }

int vio_bed_contains(const vio_bed *bed, const char *contig, int32_t start, int32_t stop, int32_t type) {
    if (!bed) return VIO_BED_INSIDE;
    const auto it = bed->regions.find(contig);
    if (it == bed->regions.end()) return VIO_BED_OFFCTG;
    const std::vector<int32_t> &st = it->second.first, &sp = it->second.second;
    if (stop <= st.front()) return VIO_BED_OUTSIDE;
    if (start >= sp.back()) return VIO_BED_OUTSIDE;
    const int a = int(std::upper_bound(st.begin(), st.end(), start) - st.begin()) - 1;
    const int b = int(std::lower_bound(sp.begin(), sp.end(), stop) - sp.begin());
    if (a < 0 || b >= int(sp.size())) return VIO_BED_BORDER;
    if (b == a) return (type == T_INS && start == sp[size_t(b)] - 1) ? VIO_BED_BORDER : VIO_BED_INSIDE;   // no INS at a region's last base
    if (b == a + 1) return (start >= sp[size_t(a)] && stop <= st[size_t(b)]) ? VIO_BED_OUTSIDE : VIO_BED_BORDER;
    return VIO_BED_BORDER;
}

int vio_read_vcf(const char *path, const vio_bed *bed, const vio_params *prm, const char *const *filters,
                 int32_t n_filters, vio_callset **out) {
    if (!path || !prm || !out) return VIO_ERR_ARG;
    *out = nullptr;
    LineReader R;
    if (!R.open(path)) return fail(VIO_ERR_OPEN, "Failed to open VCF '%s'", path);
    std::map<std::string, int64_t> hdr_len;
    std::vector<std::string> ctgs;
    std::vector<int32_t> ploidy;
    std::vector<std::vector<HapBuild>> build;      // [ctg][hap]
    std::string sample, line, prev_ctg;
    bool have_ps = false, have_gt = false, saw_header = false;
    vio_callset S;
    memset(&S, 0, sizeof(S));
    int prev_end[2] = {0, 0}, prev_type[2] = {T_SUB, T_SUB};
    while (R.next(line)) {
        if (line.empty()) continue;
        if (line[0] == '#') {
            if (line.compare(0, 9, "##contig=") == 0) {
                const size_t i = line.find("ID="), l = line.find("length=");
                if (i != std::string::npos) {
                    const size_t e = line.find_first_of(",>", i);
                    hdr_len[line.substr(i + 3, e - i - 3)] = l == std::string::npos ? 0 : atoll(line.c_str() + l + 7);
                }
            } else if (line.compare(0, 9, "##FORMAT=") == 0) {
                if (line.find("ID=PS,") != std::string::npos || line.find("ID=PS>") != std::string::npos) have_ps = true;
                if (line.find("ID=GT,") != std::string::npos || line.find("ID=GT>") != std::string::npos) have_gt = true;
            } else if (line.compare(0, 6, "#CHROM") == 0) {
                const std::vector<std::string> f = split(line, '\t');
                if (f.size() != 10)
                    return fail(VIO_ERR_FORMAT, "Expected 1 sample but found %d in VCF '%s'", int(f.size()) - 9, path);
                sample = f[9];
                saw_header = true;
            }
            continue;
        }
        if (!saw_header) return fail(VIO_ERR_FORMAT, "VCF '%s' has records before the #CHROM line", path);
        const std::vector<std::string> f = split(line, '\t');
        if (f.size() < 10) return fail(VIO_ERR_FORMAT, "VCF '%s': record with %d fields", path, int(f.size()));
        const std::string &ctg = f[0];
        if (ctg != prev_ctg) {
            if (std::find(ctgs.begin(), ctgs.end(), ctg) != ctgs.end())
                return fail(VIO_ERR_FORMAT, "Unsorted VCF '%s', contig '%s' already parsed", path, ctg.c_str());
            ctgs.push_back(ctg);
            ploidy.push_back(0);
            build.emplace_back(2);
            prev_ctg = ctg;
            prev_end[0] = prev_end[1] = -2 * prm->cluster_min_gap;
            prev_type[0] = prev_type[1] = T_SUB;
        }
        const size_t ci = ctgs.size() - 1;
        S.n_records++;
        // FILTER: passes when nothing was selected, when the field is missing, or when it carries a selected id
        bool pass = n_filters == 0 || f[6] == ".";
        if (!pass)
            for (const std::string &fl : split(f[6], ';'))
                for (int k = 0; k < n_filters; k++)
                    if (fl == filters[k]) pass = true;
        if (!pass) { S.n_failed_filter++; continue; }
        float vq = f[5] == "." ? 0.0f : float(atof(f[5].c_str()));
        if (std::isnan(vq)) vq = 0;
        if (vq < float(prm->min_qual)) { S.n_low_qual++; continue; }
        // FORMAT fields of the single sample
        const std::vector<std::string> keys = split(f[8], ':'), vals = split(f[9], ':');
        auto field = [&](const char *k) -> const std::string * {
            for (size_t i = 0; i < keys.size() && i < vals.size(); i++)
                if (keys[i] == k) return &vals[i];
            return nullptr;
        };
        float gq = 0;
        if (const std::string *g = field("GQ")) gq = (*g == ".") ? 0.0f : float(int(atof(g->c_str())));
        // GT
        int ngt = -1;
        int al[2] = {-1, -1};
        bool phased = false;
        if (const std::string *g = field("GT")) {
            ngt = 0;
            size_t a = 0;
            while (true) {
                const size_t b = g->find_first_of("|/", a);
                const std::string tok = g->substr(a, b == std::string::npos ? std::string::npos : b - a);
                if (ngt < 2) al[ngt] = (tok == "." || tok.empty()) ? -1 : atoi(tok.c_str());
                ngt++;
                if (b == std::string::npos) break;
                if ((*g)[b] == '|') phased = true;      // htslib sets the phased bit on the allele behind a '|'
                a = b + 1;
            }
            if (ngt > 2)
                return fail(VIO_ERR_FORMAT, "Expected monoploid/diploid VCF, found variant with ploidy %d at %s:%s", ngt, ctg.c_str(), f[1].c_str());
        }
        // a record without GT: monoploid when the header does not declare GT either (the reference warns once and goes on,
        // variant.cpp:631-636), an error when it does (variant.cpp:637-640)
        if (ngt == -1 && have_gt)
            return fail(VIO_ERR_FORMAT, "Failed to read GT at %s:%s in VCF '%s'", ctg.c_str(), f[1].c_str(), path);
        if (ploidy[ci] != 0) {
            if (std::abs(ngt) != ploidy[ci] && ctg.back() != 'X') S.n_wrong_ploidy++;
        } else ploidy[ci] = std::abs(ngt);
        const bool same = ngt == 2 && al[0] >= 0 && al[1] >= 0 && al[0] == al[1];
        // PS
        int phase_set = 0;
        if (have_ps) {
            const std::string *p = field("PS");
            if (!p || *p == ".") { if (ngt > 1 && al[0] != al[1]) S.n_ps_missing++; }
            else phase_set = atoi(p->c_str());
        }
        const std::vector<std::string> alts = split(f[4], ',');
        const int rpos = atoi(f[1].c_str()) - 1;
        for (int hap = 0; hap < std::abs(ngt); hap++) {
            const int simple_gt = same ? 5 : (hap ? 3 : 4);   // GT_ALT1_ALT1 : GT_REF_ALT1 : GT_ALT1_REF
            std::string ref = f[3];
            const int alt_idx = ngt < 0 ? 1 : al[hap];
            if (alt_idx < 0) { S.n_unknown_allele++; continue; }
            if (alt_idx == 0) continue;
            if (alt_idx > int(alts.size())) return fail(VIO_ERR_FORMAT, "allele index %d out of range at %s:%s", alt_idx, ctg.c_str(), f[1].c_str());
            std::string alt = alts[size_t(alt_idx) - 1];
            if (ngt == 2 && !same && !phased) { S.n_unphased++; continue; }
            if (alt == "*") { S.n_spanning_del++; continue; }
            // Allele normalisation (same outcome as variant.cpp:769-806).  A length-changing record loses its longest
            // common prefix, then the longest common suffix of what is left of the SHORTER allele; it is a plain INS / DEL
            // when that uses the shorter allele up, otherwise a complex record (kept trimmed, split below).  An
            // equal-length record is a SUB when only its first base differs, complex (untrimmed) otherwise.
            const int reflen = int(ref.size()), altlen = int(alt.size());
            int pos = rpos, type;
            if (reflen != altlen) {
                const int shorter = std::min(reflen, altlen);
                int pre = 0, suf = 0;
                while (pre < shorter && ref[size_t(pre)] == alt[size_t(pre)]) pre++;
                while (suf < shorter - pre && ref[size_t(reflen - 1 - suf)] == alt[size_t(altlen - 1 - suf)]) suf++;
                type = pre + suf < shorter ? T_CPX : (altlen > reflen ? T_INS : T_DEL);
                ref = ref.substr(size_t(pre), size_t(reflen - pre - suf));
                alt = alt.substr(size_t(pre), size_t(altlen - pre - suf));
                pos += pre;
            } else if (reflen == 1) {
                if (ref[0] == alt[0]) { S.n_ref_call++; continue; }
                type = T_SUB;
            } else if (ref.compare(1, std::string::npos, alt, 1, std::string::npos) == 0) {
                type = T_SUB;
                ref.resize(1); alt.resize(1);
            } else type = T_CPX;
            const int rlen = type == T_INS ? 0 : (type == T_SUB ? 1 : int(ref.size()));
            const int loc = vio_bed_contains(bed, ctg.c_str(), rpos, rpos + reflen, type);
            if (loc == VIO_BED_OUTSIDE) { S.n_bed_outside++; continue; }
            if (loc == VIO_BED_OFFCTG) { S.n_bed_offctg++; continue; }
            if (loc == VIO_BED_BORDER) { S.n_bed_border++; continue; }
            if (int(ref.size()) > prm->max_size || int(alt.size()) > prm->max_size) { S.n_large++; continue; }
            if (prev_end[hap] > pos || (prev_end[hap] == pos && prev_type[hap] == T_INS && type == T_INS)) { S.n_overlap++; continue; }
            for (char &c : ref) c = char(toupper(c));
            for (char &c : alt) c = char(toupper(c));
            const float q = std::min(vq, float(prm->max_qual));
            HapBuild &H = build[ci][size_t(hap)];
            if (type == T_CPX) {
                H.add(pos, 0, T_INS, "", alt, simple_gt, gq, q, phase_set);
                H.add(pos, rlen, T_DEL, ref, "", simple_gt, gq, q, phase_set);
                S.n_complex++;
            } else {
                H.add(pos, rlen, type, ref, alt, simple_gt, gq, q, phase_set);
            }
            prev_end[hap] = pos + rlen;
            prev_type[hap] = type;
        }
    }
    vio_callset *C = static_cast<vio_callset *>(malloc(sizeof(vio_callset)));
    *C = S;
    C->n_ctg = int32_t(ctgs.size());
    C->ctg_name = static_cast<char **>(malloc(std::max<size_t>(ctgs.size(), 1) * sizeof(char *)));
    C->ctg_len = static_cast<int64_t *>(malloc(std::max<size_t>(ctgs.size(), 1) * 8));
    C->ploidy = dup_vec(ploidy);
    C->vars = static_cast<vio_hap_vars *>(calloc(std::max<size_t>(ctgs.size(), 1) * 2, sizeof(vio_hap_vars)));
    C->sample = dup_str(sample);
    for (size_t c = 0; c < ctgs.size(); c++) {
        C->ctg_name[c] = dup_str(ctgs[c]);
        C->ctg_len[c] = hdr_len.count(ctgs[c]) ? hdr_len[ctgs[c]] : 0;
        for (int hp = 0; hp < 2; hp++) {
            const HapBuild &H = build[c][size_t(hp)];
            vio_hap_vars &V = C->vars[c * 2 + size_t(hp)];
            V.n = int32_t(H.pos.size());
            V.pos = dup_vec(H.pos); V.rlen = dup_vec(H.rlen); V.type = dup_vec(H.type); V.orig_gt = dup_vec(H.orig_gt);
            V.var_qual = dup_vec(H.var_qual); V.gt_qual = dup_vec(H.gt_qual); V.phase_set = dup_vec(H.phase_set);
            V.ref_len = dup_vec(H.ref_len); V.alt_len = dup_vec(H.alt_len); V.ref_off = dup_vec(H.ref_off); V.alt_off = dup_vec(H.alt_off);
            V.pool_len = int64_t(H.pool.size());
            V.pool = static_cast<uint8_t *>(malloc(H.pool.size() + 1));
            memcpy(V.pool, H.pool.data(), H.pool.size());
        }
    }
    *out = C;
    return VIO_OK;
}

void vio_callset_free(vio_callset *c) {
    if (!c) return;
    for (int32_t k = 0; k < c->n_ctg; k++) {
        free(c->ctg_name[k]);
        for (int hp = 0; hp < 2; hp++) {
            vio_hap_vars &V = c->vars[k * 2 + hp];
            free(V.pos); free(V.rlen); free(V.type); free(V.orig_gt); free(V.var_qual); free(V.gt_qual); free(V.phase_set);
            free(V.ref_len); free(V.alt_len); free(V.ref_off); free(V.alt_off); free(V.pool);
        }
    }
    free(c->ctg_name); free(c->ctg_len); free(c->ploidy); free(c->vars); free(c->sample);
    free(c);
}

int vio_read_fasta(const char *path, vio_fasta **out) {
    if (!path || !out) return VIO_ERR_ARG;
    *out = nullptr;
    LineReader R;
    if (!R.open(path)) return fail(VIO_ERR_OPEN, "Failed to open FASTA '%s'", path);
    std::vector<std::string> names;
    std::vector<int64_t> offs;
    std::string seq, line;
    while (R.next(line)) {
        if (line.empty()) continue;
        if (line[0] == '>') {
            const size_t e = line.find_first_of(" \t", 1);
            names.push_back(line.substr(1, e == std::string::npos ? std::string::npos : e - 1));
            offs.push_back(int64_t(seq.size()));
        } else if (!names.empty()) {
            for (char c : line) if (!isspace(static_cast<unsigned char>(c))) seq.push_back(char(toupper(c)));
        }
    }
    offs.push_back(int64_t(seq.size()));
    vio_fasta *F = static_cast<vio_fasta *>(calloc(1, sizeof(vio_fasta)));
    F->n_ctg = int32_t(names.size());
    F->ctg_name = static_cast<char **>(malloc(std::max<size_t>(names.size(), 1) * sizeof(char *)));
    for (size_t k = 0; k < names.size(); k++) F->ctg_name[k] = dup_str(names[k]);
    F->ctg_off = dup_vec(offs);
    F->seq = static_cast<uint8_t *>(malloc(seq.size() + 1));
    memcpy(F->seq, seq.data(), seq.size());
    *out = F;
    return VIO_OK;
}

void vio_fasta_free(vio_fasta *f) {
    if (!f) return;
    for (int32_t k = 0; k < f->n_ctg; k++) free(f->ctg_name[k]);
    free(f->ctg_name); free(f->ctg_off); free(f->seq);
    free(f);
}

}  // extern "C"
