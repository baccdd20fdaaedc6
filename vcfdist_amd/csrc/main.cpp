// main.cpp -- vcfdist_gpu: the reference's command line for the precision/recall evaluation with the host orchestration in C++
// (main.cpp / globals.cpp of vcfdist v2.6.4: parse the arguments, read the VCFs / BED / FASTA, per contig cluster -> supercluster
// -> precision_recall_wrapper -> phase, then the counters, the PRECISION-RECALL SUMMARY and the output files) over the C ABIs of
// this repository: include/vcfdist_io.h (readers), vcfdist_cluster.h (clustering on the GPU / host, superclustering),
// vcfdist_pr.h (the alignment path on the GPU: generate_ptrs_strs on the device, vpr_execute, phasing, counters),
// vcfdist_report.h (writers).  One process, one GPU; the sharded runs (one process per GPU, RCCL) are `python -m vcfdist_amd`
// under torch.distributed.run, which is the same sequence of calls.  No CPU fallback: without a HIP device vpr_create fails.
//
//   vcfdist_gpu <query.vcf[.gz]> <truth.vcf[.gz]> <ref.fasta[.gz]> [-b regions.bed] [-p prefix] [-n] [-c biwfa | gap N | size N]
//               [-l max variant size] [-s max supercluster size] [-mn / -mx qual] [-f filters] [-i iterations] [-x -o -e penalties]
//               [-ct credit threshold] [-pt phasing threshold] [-sv threshold] [--reach-min-gap N] [--strict] [--device N]
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/vcfdist_cluster.h"
#include "../../include/vcfdist_io.h"
#include "../../include/vcfdist_pr.h"
#include "../../include/vcfdist_report.h"

namespace {

struct Args {
    std::string query, truth, fasta, bed, filter, prefix = "./", cluster = "biwfa";
    int max_size = 5000, min_qual = 0, max_qual = 60, cluster_gap = 50, max_iterations = 4, max_supercluster_size = 10000;
    int sub = 5, open = 6, extend = 2, sv_threshold = 50, reach_min_gap = 10, device = 0;
    double credit_threshold = 0.7, phase_threshold = 0.6;
    bool no_output_files = false, strict = false;
};

[[noreturn]] void die(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    exit(1);
}
void warn(const std::string &m) { fprintf(stderr, "[WARN  vcfdist] %s\n", m.c_str()); }

Args parse(int argc, char **argv) {
    Args a;
    std::vector<std::string> pos;
    auto need = [&](int &i) -> const char * { if (i + 1 >= argc) die("ERROR: option '%s' needs a value", argv[i]); return argv[++i]; };
    for (int i = 1; i < argc; i++) {
        const std::string o = argv[i];
        if (o == "-b" || o == "--bed") a.bed = need(i);
        else if (o == "-f" || o == "--filter") a.filter = need(i);
        else if (o == "-l" || o == "--largest-variant") a.max_size = atoi(need(i));
        else if (o == "-mn" || o == "--min-qual") a.min_qual = atoi(need(i));
        else if (o == "-mx" || o == "--max-qual") a.max_qual = atoi(need(i));
        else if (o == "-c" || o == "--cluster") {
            a.cluster = need(i);
            if ((a.cluster == "gap" || a.cluster == "size") && i + 1 < argc && argv[i + 1][0] != '-') a.cluster_gap = atoi(argv[++i]);
            if (a.cluster != "biwfa" && a.cluster != "gap" && a.cluster != "size") die("ERROR: unknown clustering method '%s'", a.cluster.c_str());
        }
        else if (o == "-i" || o == "--max-iterations") a.max_iterations = atoi(need(i));
        else if (o == "-s" || o == "--max-supercluster-size") a.max_supercluster_size = atoi(need(i));
        else if (o == "-x" || o == "--mismatch-penalty") a.sub = atoi(need(i));
        else if (o == "-o" || o == "--gap-open-penalty") a.open = atoi(need(i));
        else if (o == "-e" || o == "--gap-extend-penalty") a.extend = atoi(need(i));
        else if (o == "-ct" || o == "--credit-threshold") a.credit_threshold = atof(need(i));
        else if (o == "-pt" || o == "--phasing-threshold") a.phase_threshold = atof(need(i));
        else if (o == "-sv" || o == "--sv-threshold") a.sv_threshold = atoi(need(i));
        else if (o == "--reach-min-gap") a.reach_min_gap = atoi(need(i));
        else if (o == "-p" || o == "--prefix") a.prefix = need(i);
        else if (o == "-n" || o == "--no-output-files") a.no_output_files = true;
        else if (o == "--strict") a.strict = true;
        else if (o == "--device") a.device = atoi(need(i));
        else if (!o.empty() && o[0] == '-' && o.size() > 1) die("ERROR: unknown option '%s'", o.c_str());
        else pos.push_back(o);
    }
    if (pos.size() != 3) die("usage: vcfdist_gpu <query.vcf> <truth.vcf> <ref.fasta> [options]");
    a.query = pos[0]; a.truth = pos[1]; a.fasta = pos[2];
    if (a.max_size + 2 > a.max_supercluster_size)          // globals.cpp:478-481
        die("ERROR: Max supercluster size (-s) must be at least two larger than max variant size (-l).");
    return a;
}

// contigs of a BED file in the order it first names them (bedData::contigs, bed.cpp:22-26)
std::vector<std::string> bed_contigs(const std::string &path) {
    std::vector<std::string> out;
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string c;
        if (ss >> c && std::find(out.begin(), out.end(), c) == out.end()) out.push_back(c);
    }
    return out;
}

int find(const std::vector<std::string> &v, const std::string &s) {
    for (size_t i = 0; i < v.size(); i++) if (v[i] == s) return int(i);
    return -1;
}

// check_contigs (bed.cpp:135-284): the contigs to evaluate, in the order the reference's superclusterData walks them
std::vector<std::string> check_contigs(const std::vector<std::string> &q_in, const std::vector<std::string> &t_in,
                                       const std::vector<std::string> &fa, const std::vector<std::string> *bed) {
    std::vector<std::string> qc = q_in, tc = t_in;
    if (bed) {
        auto in_bed = [&](const std::string &c) { return find(*bed, c) >= 0; };
        qc.erase(std::remove_if(qc.begin(), qc.end(), [&](const std::string &c) { return !in_bed(c); }), qc.end());
        tc.erase(std::remove_if(tc.begin(), tc.end(), [&](const std::string &c) { return !in_bed(c); }), tc.end());
        for (const auto &c : qc) if (find(tc, c) < 0) warn("Contig '" + c + "' found in query VCF but not truth VCF.");
        for (const auto &c : tc) if (find(qc, c) < 0) warn("Contig '" + c + "' found in truth VCF but not query VCF.");
        for (const auto &c : *bed) {
            if (find(fa, c) < 0) die("ERROR: Contig '%s' found in BED but not reference FASTA.", c.c_str());
            if (find(qc, c) < 0) qc.push_back(c);
        }
        return qc;
    }
    for (const auto &c : tc) {
        if (find(fa, c) < 0) die("ERROR: Contig '%s' found in truth VCF but not reference FASTA. Please provide BED file.", c.c_str());
        if (find(qc, c) < 0) {
            warn("Contig '" + c + "' found in truth VCF but not query VCF. All truth variants on '" + c + "' will be false negatives.");
            qc.push_back(c);
        }
    }
    for (const auto &c : qc)
        if (find(tc, c) < 0) {
            warn("Contig '" + c + "' found in query VCF but not truth VCF. All query variants on '" + c + "' will be false positives.");
            if (find(fa, c) < 0) die("ERROR: contig '%s' not in reference FASTA", c.c_str());
        }
    return qc;
}

// superclusterData::transfer_phase_sets (cluster.cpp:186-330): one phase set per supercluster from the variants' PS -- the
// reference walks the superclusters, inside one the four haps (query 1, 2, truth 1, 2) and their variants, keeps one running
// maximum of PS per callset, and a variant whose PS exceeds its callset's maximum makes its PS the current phase set; a
// supercluster gets the phase set current at its end
std::vector<int32_t> transfer_phase_sets(const vio_hap_vars *slot[4], const std::vector<int64_t> var_off[4], int n_sc) {
    int first_pos = -1, phase_set = 0;
    for (int i = 0; i < 4; i++)
        for (int v = 0; v < slot[i]->n; v++)
            if (slot[i]->phase_set[v]) {
                if (first_pos < 0 || slot[i]->pos[v] < first_pos) { first_pos = slot[i]->pos[v]; phase_set = slot[i]->phase_set[v]; }
                break;
            }
    std::vector<int32_t> out(size_t(n_sc), 0);
    int cur[2] = {0, 0};
    for (int k = 0; k < n_sc; k++) {
        for (int i = 0; i < 4; i++)
            for (int64_t v = var_off[i][size_t(k)]; v < var_off[i][size_t(k) + 1]; v++) {
                const int ps = slot[i]->phase_set[v];
                if (ps > cur[i >> 1]) { cur[i >> 1] = ps; phase_set = ps; }
            }
        out[size_t(k)] = phase_set;
    }
    return out;
}

struct ContigOut {       // what the writers need of one contig, kept alive until they have run
    std::string name;
    int64_t length = 0;
    int ploidy = 0;
    const vio_hap_vars *slot[4];
    vcl_superclusters *sc = nullptr;
    vpr_results res;
    void *res_block = nullptr;
    std::vector<int32_t> phase_sets, pb, sw, fl, phase_block;
};

const vio_hap_vars EMPTY_HAP = {0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};

}  // namespace

int main(int argc, char **argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);        // (before HIP starts: include/vcfdist_pr.h, vpr_select_device)
    const Args A = parse(argc, argv);
    std::string cmd = "vcfdist";
    for (int i = 1; i < argc; i++) { cmd += " "; cmd += argv[i]; }
    (void)vpr_select_device(A.device);

    vio_bed *bed = nullptr;
    std::vector<std::string> bedc;
    if (!A.bed.empty()) {
        if (vio_read_bed(A.bed.c_str(), &bed)) die("ERROR: %s", vio_last_error());
        bedc = bed_contigs(A.bed);
    }
    std::vector<std::string> filters;
    { std::istringstream ss(A.filter); std::string f; while (std::getline(ss, f, ',')) if (!f.empty()) filters.push_back(f); }
    std::vector<const char *> fptr;
    for (const auto &f : filters) fptr.push_back(f.c_str());
    const vio_params P = {A.min_qual, A.max_qual, A.max_size, A.cluster_gap};
    vio_callset *q = nullptr, *t = nullptr;
    vio_fasta *fa = nullptr;
    if (vio_read_vcf(A.query.c_str(), bed, &P, fptr.data(), int32_t(fptr.size()), &q)) die("ERROR: %s", vio_last_error());
    if (vio_read_vcf(A.truth.c_str(), bed, &P, fptr.data(), int32_t(fptr.size()), &t)) die("ERROR: %s", vio_last_error());
    if (vio_read_fasta(A.fasta.c_str(), &fa)) die("ERROR: %s", vio_last_error());
    std::vector<std::string> qn, tn, fn;
    for (int k = 0; k < q->n_ctg; k++) qn.push_back(q->ctg_name[k]);
    for (int k = 0; k < t->n_ctg; k++) tn.push_back(t->ctg_name[k]);
    for (int k = 0; k < fa->n_ctg; k++) fn.push_back(fa->ctg_name[k]);
    const std::vector<std::string> contigs = check_contigs(qn, tn, fn, bed ? &bedc : nullptr);

    const int nq = A.max_qual - A.min_qual + 1;
    std::vector<int64_t> total(size_t(2) * VPR_VARTYPES * 3 * size_t(nq), 0);
    vpr_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.device = A.device; cfg.max_qual = float(A.max_qual); cfg.credit_threshold = A.credit_threshold; cfg.phase_threshold = A.phase_threshold;
    cfg.band_mode = 1;
    vpr_handle *h = nullptr;
    if (vpr_create(&cfg, &h)) { fprintf(stderr, "vpr_create: %s\n", vpr_last_error(nullptr)); return 2; }

    std::vector<ContigOut *> outs;
    for (const std::string &ctg : contigs) {
        const int fi = find(fn, ctg);
        if (fi < 0) die("ERROR: contig '%s' not in reference FASTA", ctg.c_str());
        const uint8_t *seq = fa->seq + fa->ctg_off[fi];
        const int64_t seq_len = fa->ctg_off[fi + 1] - fa->ctg_off[fi];
        const int qi = find(qn, ctg), ti = find(tn, ctg);
        ContigOut *C = new ContigOut();
        C->name = ctg;
        C->slot[0] = qi >= 0 ? &q->vars[2 * qi] : &EMPTY_HAP; C->slot[1] = qi >= 0 ? &q->vars[2 * qi + 1] : &EMPTY_HAP;
        C->slot[2] = ti >= 0 ? &t->vars[2 * ti] : &EMPTY_HAP; C->slot[3] = ti >= 0 ? &t->vars[2 * ti + 1] : &EMPTY_HAP;
        // superclusterData ctor, cluster.cpp:134-157: the query's header wins
        if (qi >= 0) { C->length = q->ctg_len[qi]; C->ploidy = q->ploidy[qi]; }
        else if (ti >= 0) { C->length = t->ctg_len[ti]; C->ploidy = t->ploidy[ti]; }
        else { C->length = seq_len; C->ploidy = 0; }

        // ---- clustering (cluster.cpp:954-1263 on the GPU, or the distance rules) and superclustering (cluster.cpp:404-808)
        vcl_hap haps[4];
        vcl_clusters *cl[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < 4; i++) {
            const vio_hap_vars *s = C->slot[i];
            haps[i] = vcl_hap{s->n, s->pos, s->rlen, s->type, s->ref_len, s->alt_len};
            int rc;
            if (A.cluster == "biwfa") {
                const vcl_hap_seq hs = {haps[i], s->ref_off, s->alt_off, s->pool};
                rc = vcl_wfa_cluster(&hs, seq, int32_t(seq_len), A.sub, A.open, A.extend, A.max_iterations, A.reach_min_gap, A.device, &cl[i], nullptr);
            } else {
                rc = vcl_simple_cluster(&haps[i], A.cluster == "size" ? 1 : 0, A.cluster_gap, A.reach_min_gap, &cl[i]);
            }
            if (rc) die("ERROR: contig '%s': clustering failed (%d)", ctg.c_str(), rc);
        }
        const vcl_clusters *ccl[4] = {cl[0], cl[1], cl[2], cl[3]};
        if (vcl_supercluster(haps, ccl, A.max_supercluster_size, &C->sc)) die("ERROR: contig '%s': superclustering failed", ctg.c_str());
        const vcl_superclusters *sc = C->sc;
        const int n_sc = sc->n;
        std::vector<int64_t> var_off[4];
        for (int i = 0; i < 4; i++) {
            var_off[i].assign(size_t(n_sc) + 1, 0);
            const vcl_clusters *c = sc->clusters[i];
            if (c->n > 0) for (int k = 0; k <= n_sc; k++) var_off[i][size_t(k)] = c->var_beg[sc->brk[i][k]];
        }
        memset(&C->res, 0, sizeof(C->res));
        int n_clusters = 0;
        int64_t n_hapvars = 0;
        for (int i = 0; i < 4; i++) { n_clusters += cl[i]->n; n_hapvars += C->slot[i]->n; }
        if (n_sc > 0) {
            // ---- the path: variant tables + contig over the link, generate_ptrs_strs and everything behind it on the device
            const int64_t ctg_off[2] = {0, seq_len};
            std::vector<int32_t> sc_ctg(size_t(n_sc), 0);
            vpr_variants V;
            memset(&V, 0, sizeof(V));
            V.n_sc = n_sc; V.n_ctg = 1; V.ctg_off = ctg_off; V.ctg_seq = seq; V.sc_ctg = sc_ctg.data(); V.sc_beg = sc->beg; V.sc_end = sc->end;
            static const uint8_t no_pool[1] = {0};
            for (int i = 0; i < 4; i++) {
                const vio_hap_vars *s = C->slot[i];
                V.var_off[i] = var_off[i].data(); V.var_pos[i] = s->pos; V.var_type[i] = s->type; V.var_qual[i] = s->var_qual;
                V.var_ref_off[i] = s->ref_off; V.var_ref_len[i] = s->ref_len; V.var_alt_off[i] = s->alt_off; V.var_alt_len[i] = s->alt_len;
                V.allele_pool[i] = s->pool ? s->pool : no_pool;
            }
            if (vpr_upload_variants(h, &V) || vpr_execute(h)) die("ERROR: contig '%s': %s", ctg.c_str(), vpr_last_error(h));
            if (vpr_results_alloc(h, &C->res, &C->res_block) || vpr_download(h, &C->res)) die("ERROR: contig '%s': %s", ctg.c_str(), vpr_last_error(h));
            // a supercluster with an alignment the GPU path did not evaluate takes no side in the contig's phasing
            const uint32_t err_bits = VPR_ST_ERR_LIMIT | VPR_ST_ERR_NO_PTR | VPR_ST_ERR_UNFINISHED;
            for (int a = 0; a < 4 * n_sc; a++) if (C->res.aln_status[a] & err_bits) C->res.sc_phase[a >> 2] = VPR_PHASE_NONE;
            C->phase_sets = transfer_phase_sets(C->slot, var_off, n_sc);
            C->pb.assign(size_t(n_sc), 0); C->sw.assign(size_t(n_sc), 0); C->fl.assign(size_t(n_sc), 0);
            int32_t ns = 0, nf = 0;
            if (vpr_phase(C->res.sc_phase, C->phase_sets.data(), n_sc, C->pb.data(), C->sw.data(), &ns, C->fl.data(), &nf))
                die("ERROR: contig '%s': unexpected phase", ctg.c_str());
            C->sw.resize(size_t(ns)); C->fl.resize(size_t(nf));
            std::vector<uint8_t> cls[4];
            const uint8_t *clsp[4];
            for (int i = 0; i < 4; i++) {
                const vio_hap_vars *s = C->slot[i];
                cls[i].assign(size_t(std::max(s->n, 1)), 0);
                if (s->n) vpr_var_class(s->type, s->ref_len, s->alt_len, s->n, A.sv_threshold, cls[i].data());
                clsp[i] = cls[i].data();
            }
            std::vector<int64_t> counts(total.size(), 0);
            if (vpr_pr_counts(h, clsp, C->pb.data(), A.min_qual, A.max_qual, counts.data())) die("ERROR: contig '%s': %s", ctg.c_str(), vpr_last_error(h));
            for (size_t k = 0; k < total.size(); k++) total[k] += counts[k];
            // the reference's WARN lines (dist.cpp:1203-1223) and -- loudly -- what this implementation did not evaluate
            static const struct { uint32_t bit; const char *text; } W[] = {
                {VPR_ST_WARN_REF_ED, "Nonzero reference edit distance with no truth variants at ctg %s supercluster %d"},
                {VPR_ST_WARN_QUERY_ED, "Query edit distance changed with no query variants at ctg %s supercluster %d"},
                {VPR_ST_WARN_EXCEEDS, "Query edit distance exceeds reference edit distance at ctg %s supercluster %d"},
                {VPR_ST_WARN_ZERO_ED, "Zero edit distance with truth variants at ctg %s supercluster %d"}};
            for (const auto &w : W)
                for (int a = 0; a < 4 * n_sc; a++)
                    if (C->res.aln_status[a] & w.bit) { fputs("[WARN  vcfdist] ", stderr); fprintf(stderr, w.text, ctg.c_str(), a / 4); fputc('\n', stderr); }
            int n_bad = 0;
            for (int k = 0; k < n_sc; k++) {
                bool bad = false;
                for (int i = 0; i < 4; i++) bad = bad || (C->res.aln_status[4 * k + i] & err_bits);
                n_bad += bad;
            }
            if (n_bad) {
                fprintf(stderr, "[WARN  vcfdist_amd] contig '%s': %d supercluster(s) with alignments beyond an implementation limit of the GPU path: "
                                "NOT EVALUATED -- their variants are left out of every count and table\n", ctg.c_str(), n_bad);
                if (A.strict) die("ERROR: contig '%s': %d supercluster(s) not evaluated (--strict)", ctg.c_str(), n_bad);
            }
        } else {
            // (no superclusters: the writers still want the per-variant columns -- there are no variants either)
            static uint8_t zero8[1]; static int32_t zero32[1]; static float zerof[1];
            for (int i = 0; i < 4; i++)
                for (int w = 0; w < 2; w++) {
                    C->res.errtype[i][w] = zero8; C->res.sync_group[i][w] = zero32; C->res.credit[i][w] = zerof;
                    C->res.ref_ed[i][w] = zero32; C->res.query_ed[i][w] = zero32; C->res.callq[i][w] = zerof;
                }
            C->res.sc_phase = zero32; C->res.orig_phase_dist = zero32; C->res.swap_phase_dist = zero32;
        }
        fprintf(stderr, "[vcfdist_amd] %s: %lld hap-variants, %d clusters, %d superclusters, %zu switch / %zu flip errors\n", ctg.c_str(),
                (long long)n_hapvars, n_clusters, n_sc, C->sw.size(), C->fl.size());
        C->phase_block.assign(size_t(n_sc) + 1, 0);
        const int32_t n_pb = vrp_phase_blocks(C->phase_sets.data(), n_sc, C->phase_block.data());
        C->phase_block.resize(size_t(std::max(n_pb, 0)) + 1);
        for (int i = 0; i < 4; i++) vcl_clusters_free(cl[i]);
        outs.push_back(C);
    }

    vpr_pr_row rows[2 * VPR_VARTYPES];
    if (vpr_pr_summary(total.data(), A.min_qual, A.max_qual, rows)) die("ERROR: vpr_pr_summary failed");
    if (!A.no_output_files) {
        if (vrp_write_precision_recall(A.prefix.c_str(), total.data(), A.min_qual, A.max_qual)) die("ERROR: %s", vrp_last_error());
        {   // parameters.txt, write_params (print.cpp:30-56)
            FILE *f = fopen((A.prefix + "parameters.txt").c_str(), "w");
            if (!f) die("ERROR: cannot write %sparameters.txt", A.prefix.c_str());
            fprintf(f, "program = '%s'\nversion = '%s'\nout_prefix = '%s'\ncommand = '%s'\nreference_fasta = '%s'\n"
                       "query_vcf = '%s'\ntruth_vcf = '%s'\nbed_file = '%s'\nwrite_outputs = %s\nfilters = '%s'\n"
                       "min_var_qual = %d\nmax_var_qual = %d\nmax_var_size = %d\nsv_threshold = %d\n"
                       "phase_threshold = %f\ncredit_threshold = %f\nrealign_truth = %s\nrealign_query = %s\n"
                       "realign_only = %s\ncluster_method = '%s'\ncluster_min_gap = %d\n"
                       "reach_min_gap = %d\nmax_cluster_itrs = %d\nmax_threads = %d\nmax_ram = %f\n"
                       "sub = %d\nopen = %d\nextend = %d\neval_sub = %d\neval_open = %d\neval_extend = %d\ndistance = %s",
                    "vcfdist_amd", vpr_version(), A.prefix.c_str(), cmd.c_str(), A.fasta.c_str(), A.query.c_str(), A.truth.c_str(), A.bed.c_str(), "true",
                    A.filter.c_str(), A.min_qual, A.max_qual, A.max_size, A.sv_threshold, A.phase_threshold, A.credit_threshold, "false", "false",
                    "false", A.cluster.c_str(), A.cluster_gap, A.reach_min_gap, A.max_iterations, 64, 64.0, A.sub, A.open, A.extend, 3, 2, 1, "false");
            fclose(f);
        }
        std::vector<vrp_contig> ctgs(outs.size());
        for (size_t k = 0; k < outs.size(); k++) {
            ContigOut *C = outs[k];
            vrp_contig &c = ctgs[k];
            memset(&c, 0, sizeof(c));
            const int fi = find(fn, C->name);
            c.name = C->name.c_str(); c.length = int32_t(C->length); c.ploidy = C->ploidy;
            c.seq = fa->seq + fa->ctg_off[fi]; c.seq_len = fa->ctg_off[fi + 1] - fa->ctg_off[fi];
            static const uint8_t no_pool[1] = {0};
            for (int i = 0; i < 4; i++) {
                const vio_hap_vars *s = C->slot[i];
                vrp_hap &hp = c.hap[i];
                hp.n_var = s->n; hp.pos = s->pos; hp.type = s->type; hp.loc = nullptr; hp.var_qual = s->var_qual; hp.phase_set = s->phase_set;
                hp.ref_len = s->ref_len; hp.alt_len = s->alt_len; hp.ref_off = s->ref_off; hp.alt_off = s->alt_off; hp.pool = s->pool ? s->pool : no_pool;
                hp.n_cluster = C->sc->clusters[i]->n;
                hp.cluster_beg = C->sc->clusters[i]->n ? C->sc->clusters[i]->var_beg : nullptr;
                for (int w = 0; w < 2; w++) {
                    hp.errtype[w] = C->res.errtype[i][w]; hp.credit[w] = C->res.credit[i][w]; hp.sync_group[w] = C->res.sync_group[i][w];
                    hp.ref_ed[w] = C->res.ref_ed[i][w]; hp.query_ed[w] = C->res.query_ed[i][w];
                }
                c.sc_brk[i] = C->sc->brk[i];
            }
            c.n_sc = C->sc->n; c.sc_beg = C->sc->beg; c.sc_end = C->sc->end;
            c.sc_phase = C->res.sc_phase; c.pb_phase = C->pb.data(); c.orig_phase_dist = C->res.orig_phase_dist; c.swap_phase_dist = C->res.swap_phase_dist;
            c.sc_phase_set = C->phase_sets.data(); c.n_pb = int32_t(C->phase_block.size()) - 1; c.phase_block = C->phase_block.data();
            c.n_switches = int32_t(C->sw.size()); c.n_flips = int32_t(C->fl.size()); c.switches = C->sw.data(); c.flips = C->fl.data();
        }
        const int32_t n = int32_t(ctgs.size());
        auto path = [&](const char *name) { return A.prefix + name; };
        if (vrp_write_phase_blocks(path("phase-blocks.tsv").c_str(), ctgs.data(), n) || vrp_write_switchflips(path("switchflips.tsv").c_str(), ctgs.data(), n) ||
            vrp_write_phasing_summary(path("phasing-summary.tsv").c_str(), ctgs.data(), n) || vrp_write_superclusters(path("superclusters.tsv").c_str(), ctgs.data(), n) ||
            vrp_write_variants(path("query.tsv").c_str(), ctgs.data(), n, 0) || vrp_write_variants(path("truth.tsv").c_str(), ctgs.data(), n, 1) ||
            vrp_write_summary_vcf(path("summary.vcf").c_str(), ctgs.data(), n, cmd.c_str(), nullptr, float(A.credit_threshold)))
            die("ERROR: %s", vrp_last_error());
    }
    printf("PRECISION-RECALL SUMMARY\n\n");
    printf("TYPE\tTHRESHOLD\tTRUTH_TP\tQUERY_TP\tTRUTH_FN\tQUERY_FP\tPREC\t\tRECALL\t\tF1_SCORE\tF1_QSCORE\n");
    static const char *NAMES[] = {"SNP", "INDEL", "SV", "ALL"};
    for (const vpr_pr_row &r : rows) {
        printf("%s\t%s Q >= %-2d\t%-16d%-16d%-16d%-16d%f\t%f\t%f\t%f\n", NAMES[r.vartype], r.best ? "BEST" : "NONE", r.qual, r.truth_tp, r.query_tp,
               r.truth_fn, r.query_fp, r.precision, r.recall, r.f1_score, r.f1_qscore);
        if (r.best) printf("\n");
    }
    for (ContigOut *C : outs) {
        if (C->res_block) vpr_host_free(C->res_block);
        vcl_superclusters_free(C->sc);
        delete C;
    }
    vpr_destroy(h);
    vio_callset_free(q); vio_callset_free(t); vio_fasta_free(fa);
    if (bed) vio_bed_free(bed);
    return 0;
}
