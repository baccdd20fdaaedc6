// caller.cpp -- a compiled C++ caller of the C ABI (include/vcfdist_pr.h), the shape of what INTEGRATION.md sketches for
// the reference's precision_recall_wrapper (src/dist.cpp:1731-1904): variants + reference in, the fields the reference
// writes in place out.  The input is the reference-produced toy vector of SURVEY.md Appendix A.1 (tests/golden/toy_a1.json):
// one supercluster chr:3-15 of ACGTACGTTTTTGGCA, query = SUB 4 A>G, INS 8 TT, DEL 12 GG on both haplotypes, truth = the
// SUB on both.  Prints one line per alignment and per variant; exits 0 when the known answer (s = 0 for all four
// alignments, QUERY end plane) comes back.
//     g++ -std=c++17 -I include examples/caller.cpp -L vcfdist_amd/lib -lvcfdist_pr -Wl,-rpath,$PWD/vcfdist_amd/lib -o caller
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "vcfdist_pr.h"

int main() {
    const char *ref = "ACGTACGTTTTTGGCA";
    const int64_t ctg_off[2] = {0, 16};
    const int32_t sc_ctg[1] = {0}, sc_beg[1] = {3}, sc_end[1] = {15};
    // query haplotypes (slots 0, 1): three variants each; truth haplotypes (slots 2, 3): one
    const int64_t off_q[2] = {0, 3}, off_t[2] = {0, 1};
    const int32_t pos_q[3] = {4, 8, 12}, pos_t[1] = {4};
    const uint8_t type_q[3] = {VPR_TYPE_SUB, VPR_TYPE_INS, VPR_TYPE_DEL}, type_t[1] = {VPR_TYPE_SUB};
    const float qual_q[3] = {30, 30, 30}, qual_t[1] = {30};
    const uint8_t pool_q[] = {'A', 'G', 'T', 'T', 'G', 'G'};     // SUB: ref A alt G | INS: alt TT | DEL: ref GG
    const int64_t roff_q[3] = {0, 2, 4}, aoff_q[3] = {1, 2, 6};
    const int32_t rlen_q[3] = {1, 0, 2}, alen_q[3] = {1, 2, 0};
    const uint8_t pool_t[] = {'A', 'G'};
    const int64_t roff_t[1] = {0}, aoff_t[1] = {1};
    const int32_t rlen_t[1] = {1}, alen_t[1] = {1};

    vpr_variants in;
    memset(&in, 0, sizeof(in));
    in.n_sc = 1; in.n_ctg = 1;
    in.ctg_off = ctg_off; in.ctg_seq = reinterpret_cast<const uint8_t *>(ref);
    in.sc_ctg = sc_ctg; in.sc_beg = sc_beg; in.sc_end = sc_end;
    for (int h = 0; h < VPR_HAPS; h++) {
        const bool q = h < 2;
        in.var_off[h] = q ? off_q : off_t;
        in.var_pos[h] = q ? pos_q : pos_t;
        in.var_type[h] = q ? type_q : type_t;
        in.var_qual[h] = q ? qual_q : qual_t;
        in.var_ref_off[h] = q ? roff_q : roff_t; in.var_ref_len[h] = q ? rlen_q : rlen_t;
        in.var_alt_off[h] = q ? aoff_q : aoff_t; in.var_alt_len[h] = q ? alen_q : alen_t;
        in.allele_pool[h] = q ? pool_q : pool_t;
    }

    vpr_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.device = 0; cfg.max_qual = 60.0f; cfg.credit_threshold = 0.7; cfg.phase_threshold = 0.6; cfg.band_mode = 1;
    vpr_handle *h = nullptr;
    if (vpr_create(&cfg, &h) != VPR_OK) { fprintf(stderr, "vpr_create: %s\n", vpr_last_error(nullptr)); return 2; }   // no GPU: hard error
    if (vpr_upload_variants(h, &in) != VPR_OK || vpr_execute(h) != VPR_OK) { fprintf(stderr, "%s\n", vpr_last_error(h)); return 2; }

    const int nv[VPR_HAPS] = {3, 3, 1, 1};
    std::vector<int32_t> aln_dist(4), sc_phase(1), orig(1), swp(1);
    std::vector<uint8_t> end_plane(4), beg_plane(4);
    std::vector<uint32_t> status(4);
    std::vector<uint8_t> errtype[VPR_HAPS][2];
    std::vector<int32_t> sync_group[VPR_HAPS][2], ref_ed[VPR_HAPS][2], query_ed[VPR_HAPS][2];
    std::vector<float> credit[VPR_HAPS][2], callq[VPR_HAPS][2];
    vpr_results out;
    memset(&out, 0, sizeof(out));
    out.aln_dist = aln_dist.data(); out.aln_end_plane = end_plane.data(); out.aln_beg_plane = beg_plane.data();
    out.aln_status = status.data(); out.sc_phase = sc_phase.data(); out.orig_phase_dist = orig.data(); out.swap_phase_dist = swp.data();
    for (int s = 0; s < VPR_HAPS; s++)
        for (int w = 0; w < 2; w++) {
            errtype[s][w].resize(nv[s]); sync_group[s][w].resize(nv[s]); ref_ed[s][w].resize(nv[s]); query_ed[s][w].resize(nv[s]);
            credit[s][w].resize(nv[s]); callq[s][w].resize(nv[s]);
            out.errtype[s][w] = errtype[s][w].data(); out.sync_group[s][w] = sync_group[s][w].data();
            out.ref_ed[s][w] = ref_ed[s][w].data(); out.query_ed[s][w] = query_ed[s][w].data();
            out.credit[s][w] = credit[s][w].data(); out.callq[s][w] = callq[s][w].data();
        }
    if (vpr_download(h, &out) != VPR_OK) { fprintf(stderr, "%s\n", vpr_last_error(h)); return 2; }
    vpr_destroy(h);

    bool ok = true;
    for (int i = 0; i < 4; i++) {
        printf("alignment %d: s %d end plane %s status %u\n", i, aln_dist[i], end_plane[i] == VPR_PLANE_QUERY ? "QUERY" : "REF", status[i]);
        ok = ok && aln_dist[i] == 0 && end_plane[i] == VPR_PLANE_QUERY;
    }
    printf("phase %d (orig %d, swap %d)\n", sc_phase[0], orig[0], swp[0]);
    const char *ET[6] = {"TP", "FP", "FN", "?", "?", "UN"};
    for (int s = 0; s < VPR_HAPS; s++)
        for (int v = 0; v < nv[s]; v++)
            printf("hap slot %d variant %d: %s / %s credit %.2f / %.2f\n", s, v, ET[errtype[s][0][v] > 5 ? 3 : errtype[s][0][v]],
                   ET[errtype[s][1][v] > 5 ? 3 : errtype[s][1][v]], credit[s][0][v], credit[s][1][v]);
    return ok ? 0 : 1;
}
