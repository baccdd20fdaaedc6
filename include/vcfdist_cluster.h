/* vcfdist_cluster.h -- C ABI of the step immediately in front of the precision/recall path
 * (SURVEY.md 8(f) rank 1): dependency clustering by distance and superclustering of the four
 * haplotypes' clusters, i.e. what defines the work units of include/vcfdist_pr.h.
 *
 * Reference interfaces replaced (TimD1/vcfdist v2.6.4):
 *   vcl_simple_cluster    simple_cluster(variantData*, callset)              src/cluster.cpp:826-945
 *                         ("-c gap N" / "-c size N"; biWFA clustering, cluster.cpp:954, is rank 2)
 *   vcl_supercluster      superclusterData::supercluster(bool)               src/cluster.cpp:404-553
 *                         get_supercluster_range                             src/cluster.cpp:563-594
 *                         split_large_supercluster / split_cluster /
 *                         get_next_variant_info / get_supercluster_split_location   src/cluster.cpp:601-808
 *   vcl_supercluster_cells  the size estimate of sort_superclusters          src/cluster.cpp:42-122
 *   vcl_wfa_cluster       wf_swg_cluster(variantData*, ctg, hap, sub, open, extend)  src/cluster.cpp:954-1263
 *                         with wf_swg_align (src/dist.cpp:1510), wf_swg_max_reach (src/dist.cpp:2150) and
 *                         generate_str (src/dist.cpp:81): the default "-c biwfa" clustering.
 *                         The alignments run on the GPU (HIP, no CPU fallback); the merge passes are host code.
 *
 * Host code (the reference's is host code too, O(#variants)); one contig per call; hap slot
 * i = 2*callset + hap with QUERY = 0, TRUTH = 1, the order of include/vcfdist_pr.h.
 */
#ifndef VCFDIST_CLUSTER_H_
#define VCFDIST_CLUSTER_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCL_OK        0
#define VCL_ERR_ARG  -1     /* null pointer, unsorted positions, inconsistent cluster table */
#define VCL_ERR_TYPE -2     /* size mode: a variant type other than SUB/INS/DEL ("Variant type ... unexpected",
                               cluster.cpp:873, where the size of a variant is computed; gap mode never looks at it) */
#define VCL_ERR_DEVICE -3   /* no HIP device / HIP error (vcl_wfa_cluster has no CPU fallback) */

#define VCL_SENTINEL 0x7fffffff   /* std::numeric_limits<int>::max(): reach of the sentinel cluster */

/* variant columns of one (callset, hap) on one contig: ctgVariants::{poss,rlens,types,refs,alts}, variant.h:29-40 */
typedef struct vcl_hap {
    int32_t n_var;
    const int32_t *pos;       /* 0-based, non-decreasing */
    const int32_t *rlen;      /* reference length (0 for INS) */
    const uint8_t *type;      /* VPR_TYPE_SUB / INS / DEL (src/defs.h:58-63) */
    const int32_t *ref_len;   /* allele lengths without anchor base */
    const int32_t *alt_len;
} vcl_hap;

/* ctgVariants::{clusters,left_reaches,right_reaches}: n clusters + one sentinel entry */
typedef struct vcl_clusters {
    int32_t n;                /* clusters, sentinel not counted (0 when the hap has no variants: arrays are then empty) */
    int32_t *var_beg;         /* [n+1] first variant of each cluster; var_beg[n] = n_var */
    int32_t *left_reach;      /* [n+1]; entry n = VCL_SENTINEL */
    int32_t *right_reach;     /* [n+1]; entry n = VCL_SENTINEL */
} vcl_clusters;

/* size_mode 0: "gap" (reach = cluster_min_gap), 1: "size" (reach = max(cluster_min_gap, variant size)). */
int vcl_simple_cluster(const vcl_hap *hap, int size_mode, int32_t cluster_min_gap, int32_t reach_min_gap,
                       vcl_clusters **out);
void vcl_clusters_free(vcl_clusters *c);

/* ctgSuperclusters::{superclusters,begs,ends,n}, cluster.h:13-62 */
typedef struct vcl_superclusters {
    int32_t n;                /* superclusters (the reference's trailing sentinel entry is brk[.][n]) */
    int32_t *brk[4];          /* [n+1] cluster index where supercluster k starts on hap slot i */
    int32_t *beg, *end;       /* [n] begs / ends: one base left of the first variant, one base right of the last
                                 variant end; the region [beg, end] is inclusive (dist.cpp:163) and is what
                                 vpr_variants.sc_beg / sc_end take */
    int32_t n_oversize;       /* superclusters that exceeded max_supercluster_size and were split (the WARN of cluster.cpp:503) */
    int32_t n_unsplittable;   /* "no valid splits" events (cluster.cpp:632) */
    vcl_clusters *clusters[4];/* the cluster tables after splitting (copies of the inputs when nothing was split) */
} vcl_superclusters;

int vcl_supercluster(const vcl_hap haps[4], const vcl_clusters *const clusters[4], int32_t max_supercluster_size,
                     vcl_superclusters **out);
void vcl_superclusters_free(vcl_superclusters *s);

/* ---- biWFA dependency clustering ---------------------------------------------------------------------- */
/* a hap's variants with their allele strings (ctgVariants::{refs,alts}) */
typedef struct vcl_hap_seq {
    vcl_hap cols;
    const int64_t *ref_off;   /* [n_var] start of the REF allele in pool (length cols.ref_len) */
    const int64_t *alt_off;   /* [n_var] start of the ALT allele in pool (length cols.alt_len) */
    const uint8_t *pool;
} vcl_hap_seq;

typedef struct vcl_wfa_stats {
    int32_t iterations;       /* merge iterations run (<= max_cluster_itrs) */
    int64_t align_calls;      /* wf_swg_align calls (one per active cluster and iteration) */
    int64_t reach_calls;      /* wf_swg_max_reach calls (>= 2 per active cluster: iterative doubling) */
    double  ms_device;        /* kernel time (HIP events), 0 for the oracle */
} vcl_wfa_stats;

/* Clusters of one (callset, hap) on one contig.  sub/open/extend: g.sub, g.open, g.extend (defaults 5, 6, 2);
   max_cluster_itrs: g.max_cluster_itrs (4); reach_min_gap: g.reach_min_gap (10).  device: HIP device ordinal. */
int vcl_wfa_cluster(const vcl_hap_seq *hap, const uint8_t *ctg_seq, int32_t ctg_len, int32_t sub, int32_t open,
                    int32_t extend, int32_t max_cluster_itrs, int32_t reach_min_gap, int32_t device,
                    vcl_clusters **out, vcl_wfa_stats *stats);

/* max_query_len * max_truth_len of supercluster k (the factor in front of the 20 B/cell of cluster.cpp:99) */
int64_t vcl_supercluster_cells(const vcl_hap haps[4], const vcl_superclusters *s, int32_t k);
int vcl_supercluster_cells_all(const vcl_hap haps[4], const vcl_superclusters *s, int64_t *out /* [s->n] */);

#ifdef __cplusplus
}
#endif
#endif
