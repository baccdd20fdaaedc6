/* vcfdist_report.h -- C ABI of the output files: the TSV tables and the summary VCF that
 * the reference writes after the precision/recall path, byte for byte in the reference's formats.
 *
 * Reference interfaces replaced (TimD1/vcfdist v2.6.4):
 *   vrp_phase_blocks             phaseblockData::phaseblockData (phase-set runs)   src/phase.cpp:229-262
 *   vrp_write_precision_recall   write_precision_recall (the two TSV files)        src/print.cpp:441-566
 *   vrp_write_phase_blocks       write_results, phase-blocks.tsv                   src/print.cpp:585-609
 *   vrp_write_superclusters      write_results, superclusters.tsv                  src/print.cpp:611-671
 *   vrp_write_variants           write_results, query.tsv / truth.tsv              src/print.cpp:673-876
 *   vrp_write_switchflips        phaseblockData::write_switchflips                 src/phase.cpp:406-509
 *   vrp_write_phasing_summary    phaseblockData::write_phasing_summary             src/phase.cpp:515-528
 *   vrp_ng50                     phaseblockData::calculate_ng50                    src/phase.cpp:534-626
 *   vrp_write_summary_vcf        phaseblockData::write_summary_vcf                 src/phase.cpp:8-222
 *                                ctgVariants::print_var_info / _empty / _sample    src/variant.cpp:229-286
 * Host code, like the reference's; the inputs are the columns of include/vcfdist_io.h, the tables of
 * include/vcfdist_cluster.h and the result arrays of include/vcfdist_pr.h (vpr_results, vpr_phase, vpr_pr_counts).
 */
#ifndef VCFDIST_REPORT_H_
#define VCFDIST_REPORT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRP_OK        0
#define VRP_ERR_ARG  -1     /* null pointer / inconsistent tables (the reference's "Out of bounds ..." ERRORs) */
#define VRP_ERR_OPEN -2     /* output file cannot be created, or a write to it failed (see vrp_last_error) */

/* one (callset, hap) of one contig: ctgVariants (src/variant.h:17-77) as columns */
typedef struct vrp_hap {
    int32_t n_var;
    const int32_t *pos;          /* 0-based */
    const uint8_t *type;         /* VPR_TYPE_SUB / INS / DEL */
    const uint8_t *loc;          /* VIO_BED_* of each variant; NULL: all INSIDE (every variant the reader keeps is) */
    const float   *var_qual;
    const int32_t *phase_set;    /* per-variant PS */
    const int32_t *ref_len, *alt_len;
    const int64_t *ref_off, *alt_off;
    const uint8_t *pool;
    int32_t n_cluster;           /* clusters without the sentinel */
    const int32_t *cluster_beg;  /* [n_cluster + 1] first variant of each cluster; NULL when n_var == 0 */
    /* results of the path, [swap] = the ORIG / SWAP evaluation (vpr_results.*[slot][swap]) */
    const uint8_t *errtype[2];
    const float   *credit[2];
    const int32_t *sync_group[2];
    const int32_t *ref_ed[2];
    const int32_t *query_ed[2];
} vrp_hap;

/* one contig: ctgSuperclusters + ctgPhaseblocks */
typedef struct vrp_contig {
    const char *name;
    int32_t length;              /* ##contig length of the VCF header */
    int32_t ploidy;
    const uint8_t *seq;          /* reference sequence of the contig (anchor bases of the summary VCF) */
    int64_t seq_len;
    vrp_hap hap[4];              /* slot order of include/vcfdist_pr.h: Q1, Q2, T1, T2 */
    int32_t n_sc;
    const int32_t *sc_beg, *sc_end;
    const int32_t *sc_brk[4];    /* [n_sc + 1] vcl_superclusters.brk */
    const int32_t *sc_phase;     /* VPR_PHASE_* */
    const int32_t *pb_phase;     /* vpr_phase */
    const int32_t *orig_phase_dist, *swap_phase_dist;
    const int32_t *sc_phase_set; /* per-supercluster phase set (transfer_phase_sets) */
    int32_t n_pb;
    const int32_t *phase_block;  /* [n_pb + 1] first supercluster of each phase block (vrp_phase_blocks) */
    int32_t n_switches, n_flips;
    const int32_t *switches, *flips;   /* vpr_phase */
} vrp_contig;

/* phase_block[] must hold n_sc + 1 entries; returns the number of phase blocks n_pb (>= 0; entries 0..n_pb are set,
   the last one is n_sc), or VRP_ERR_ARG. */
int32_t vrp_phase_blocks(const int32_t *sc_phase_set, int32_t n_sc, int32_t *phase_block);

/* <prefix>precision-recall.tsv and <prefix>precision-recall-summary.tsv from the counters of vpr_pr_counts
   (summed over contigs by the caller). */
int vrp_write_precision_recall(const char *prefix, const int64_t *counts, int32_t min_qual, int32_t max_qual);

int vrp_write_phase_blocks(const char *path, const vrp_contig *ctgs, int32_t n_ctg);
/* switchflips.tsv: where every switch / flip error may have happened; phasing-summary.tsv: block and error totals with
   the NG50 of the correctly phased stretches (genome size = sum of vrp_contig.length), broken at phase-block starts
   only / also at switch errors / also around flipped superclusters (vrp_ng50; -1: inconsistent tables) */
int vrp_write_switchflips(const char *path, const vrp_contig *ctgs, int32_t n_ctg);
int vrp_write_phasing_summary(const char *path, const vrp_contig *ctgs, int32_t n_ctg);
int32_t vrp_ng50(const vrp_contig *ctgs, int32_t n_ctg, int32_t break_on_switch, int32_t break_on_flip);
int vrp_write_superclusters(const char *path, const vrp_contig *ctgs, int32_t n_ctg);
/* callset 0: query.tsv, 1: truth.tsv */
int vrp_write_variants(const char *path, const vrp_contig *ctgs, int32_t n_ctg, int32_t callset);
/* cmd: the "##CL=" line; file_date: "YYYYMMDD" or NULL for today (local time); credit_threshold: g.credit_threshold */
int vrp_write_summary_vcf(const char *path, const vrp_contig *ctgs, int32_t n_ctg, const char *cmd,
                          const char *file_date, float credit_threshold);

const char *vrp_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
