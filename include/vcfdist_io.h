/* vcfdist_io.h -- C ABI of the input formats: VCF records -> per-contig, per-haplotype
 * variant columns, BED regions, FASTA contigs.  Plain-text or gzip/bgzip files through zlib; htslib is not used.
 *
 * Reference interfaces replaced (TimD1/vcfdist v2.6.4):
 *   vio_read_vcf     variantData::variantData(vcf_fn, reference, callset)   src/variant.cpp:397-1004
 *                    (record filtering, GT / GQ / PS handling, allele trimming, SUB / INS / DEL typing, CPX ->
 *                    INS + DEL, BED test on the original representation, size and overlap filters) and
 *                    ctgVariants::add_var                                     src/variant.cpp:29-53
 *   vio_read_bed / vio_bed_contains   bedData::bedData, ::contains          src/bed.cpp:4-36, 73-121
 *   vio_read_fasta   fastaData::fastaData (kseq)                            src/fasta.h:13-29
 * Host code, like the reference's.
 */
#ifndef VCFDIST_IO_H_
#define VCFDIST_IO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIO_OK        0
#define VIO_ERR_ARG  -1
#define VIO_ERR_OPEN -2     /* file cannot be opened / read */
#define VIO_ERR_FORMAT -3   /* malformed or unsupported input: the reference's ERROR() cases (unsorted contigs, more than
                               one sample, polyploid GT, malformed line); see vio_last_error */

/* bedData::contains results, src/defs.h */
#define VIO_BED_OUTSIDE 0   /* src/defs.h:50-53 */
#define VIO_BED_INSIDE  1
#define VIO_BED_BORDER  2
#define VIO_BED_OFFCTG  3

typedef struct vio_bed vio_bed;
int vio_read_bed(const char *path, vio_bed **out);
void vio_bed_free(vio_bed *b);
/* start / stop: 0-based half-open span of the record's original REF allele; type: VPR_TYPE_* of the parsed variant.
   bed == NULL: always INSIDE (g.bed_exists false). */
int vio_bed_contains(const vio_bed *bed, const char *contig, int32_t start, int32_t stop, int32_t type);

typedef struct vio_params {
    int32_t min_qual;         /* g.min_qual (0): records with QUAL below it are dropped */
    int32_t max_qual;         /* g.max_qual (60): var_qual = min(QUAL, max_qual) */
    int32_t max_size;         /* g.max_size (5000): longer alleles are dropped */
    int32_t cluster_min_gap;  /* g.cluster_min_gap (50): only seeds the overlap filter (prev_end = -2 * gap) */
} vio_params;

/* ctgVariants of one (contig, hap) */
typedef struct vio_hap_vars {
    int32_t n;
    int32_t *pos;             /* 0-based */
    int32_t *rlen;
    uint8_t *type;            /* VPR_TYPE_SUB / INS / DEL */
    uint8_t *orig_gt;         /* simplified GT (src/defs.h:55-64): 3 = 0|1, 4 = 1|0, 5 = 1|1 */
    float   *var_qual;        /* min(QUAL, max_qual) */
    float   *gt_qual;         /* GQ */
    int32_t *phase_set;       /* PS (0: none) */
    int32_t *ref_len, *alt_len;
    int64_t *ref_off, *alt_off;   /* into pool */
    uint8_t *pool;
    int64_t pool_len;
} vio_hap_vars;

typedef struct vio_callset {
    int32_t n_ctg;            /* contigs that have records, in file order */
    char **ctg_name;
    int64_t *ctg_len;         /* from the ##contig header lines (0 if absent) */
    int32_t *ploidy;
    vio_hap_vars *vars;       /* [n_ctg][2] */
    char *sample;
    /* the reference's summary counters */
    int64_t n_records, n_failed_filter, n_low_qual, n_unphased, n_overlap, n_bed_outside, n_bed_border, n_bed_offctg,
            n_large, n_complex, n_ref_call, n_spanning_del, n_unknown_allele, n_wrong_ploidy, n_ps_missing;
} vio_callset;

/* filters: FILTER ids a record must carry one of (g.filters; n_filters = 0: everything passes). */
int vio_read_vcf(const char *path, const vio_bed *bed, const vio_params *params, const char *const *filters,
                 int32_t n_filters, vio_callset **out);
void vio_callset_free(vio_callset *c);

typedef struct vio_fasta {
    int32_t n_ctg;
    char **ctg_name;          /* up to the first whitespace of the header line */
    int64_t *ctg_off;         /* [n_ctg + 1] into seq */
    uint8_t *seq;             /* upper-cased */
} vio_fasta;
int vio_read_fasta(const char *path, vio_fasta **out);
void vio_fasta_free(vio_fasta *f);

const char *vio_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
