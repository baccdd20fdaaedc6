/* vcfdist_pr.h -- C ABI of the MI355X precision/recall alignment path.
 *
 * This is the drop-in boundary for the body of the reference's
 *     void precision_recall_wrapper(superclusterData*, sc_groups, thread_step,
 *                                   start, stop, thread4)
 *         (reference: src/dist.h:248-250, src/dist.cpp:1731-1904)
 * i.e. for one *batch* of superclusters it performs what the reference does per
 * supercluster inside that loop:
 *     generate_ptrs_strs x4      src/dist.cpp:145-242    (vpr_generate / device K0)
 *     calc_prec_recall_aln       src/dist.cpp:251-443    (device K1)
 *     store_phase                src/dist.cpp:449-475    (device K5, k_finalize)
 *     calc_prec_recall_path      src/dist.cpp:486-834    (device K2, + the swap_pred
 *                                                         tie replay, pr_tie.hip)
 *     get_prec_recall_path_sync  src/dist.cpp:842-999    (device K3)
 *     calc_prec_recall           src/dist.cpp:1005-1401  (device K3 + K5)
 *     wf_ed                      src/dist.cpp:1406-1506  (device K4)
 *
 * Plain pointers and sizes only; the caller owns every buffer it passes in, the
 * library owns device memory, streams and events.  Everything is computed on the
 * device, including the reference's float expressions (credit, callq, the phase
 * threshold: IEEE single-precision division and compares in the reference's order,
 * k_finalize); vpr_download only copies.  vpr_store_phase is the same expression
 * as a host function for callers that hold four distances.
 *
 * Hap slots (the "4 haps" of a supercluster) are numbered
 *     0 = QUERY hap1, 1 = QUERY hap2, 2 = TRUTH hap1, 3 = TRUTH hap2
 * and the 4 alignments follow src/defs.h:101-104:
 *     0 = Q1T1, 1 = Q1T2, 2 = Q2T1, 3 = Q2T2   (query hap = i>>1, truth hap = i&1)
 * "swap" slot of alignment i is (i==1 || i==2)      (src/dist.cpp:1037).
 */
#ifndef VCFDIST_PR_H_
#define VCFDIST_PR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPR_HAPS 4
#define VPR_ALNS 4

/* pointer flags, src/defs.h:126-129 */
#define VPR_PTR_VARIANT 1
#define VPR_PTR_VAR_BEG 2
#define VPR_PTR_VAR_END 4
#define VPR_PTR_INS_LOC 8

/* variant types, src/defs.h:32-38 */
#define VPR_TYPE_SUB 1
#define VPR_TYPE_INS 2
#define VPR_TYPE_DEL 3

/* error types, src/defs.h:66-72 */
#define VPR_ERRTYPE_TP 0
#define VPR_ERRTYPE_FP 1
#define VPR_ERRTYPE_FN 2
#define VPR_ERRTYPE_UN 5

/* phases, src/defs.h:136-138 */
#define VPR_PHASE_ORIG 0
#define VPR_PHASE_SWAP 1
#define VPR_PHASE_NONE 2

/* planes of the two-plane alignment graph (src/defs.h:96-97) */
#define VPR_PLANE_QUERY 0
#define VPR_PLANE_REF   1

/* per-alignment status bits (vpr_results.aln_status) */
#define VPR_ST_OK            0u
#define VPR_ST_SWAP_TIE      1u   /* a cell on the optimal DAG had >1 optimal swap predecessor: the reference keeps the
                                     last writer (src/dist.cpp:347,376), an order defined by its FIFO and
                                     unordered_set iteration; the library replayed that order for this alignment
                                     (informational: the results are the reference's) */
#define VPR_ST_WARN_REF_ED   2u   /* "Nonzero reference edit distance with no truth variants" dist.cpp:1203 */
#define VPR_ST_WARN_QUERY_ED 4u   /* "Query edit distance changed with no query variants"     dist.cpp:1207 */
#define VPR_ST_WARN_EXCEEDS  8u   /* "Query edit distance exceeds reference edit distance"    dist.cpp:1211 */
#define VPR_ST_WARN_ZERO_ED  16u  /* "Zero edit distance with truth variants" (ref_ed forced to 1) dist.cpp:1219 */
#define VPR_ST_ERR_NO_PTR    32u  /* "No valid pointer" during the walk               dist.cpp:937 */
#define VPR_ST_ERR_UNFINISHED 64u /* "Alignment not finished"                         dist.cpp:440 */
#define VPR_ST_ERR_LIMIT     128u /* the alignment exceeds an implementation limit of this library (the reference has none of them); its
                                     variants stay unevaluated (VPR_ERRTYPE_UN), the rest of the batch is evaluated as usual:
                                     - more than eight allowed swap sources map onto one position of its supercluster (nine or more
                                       directly adjacent separate indel records on one haplotype; dist.cpp:335-350 has no bound);
                                     - it reached the dense level wider than one workgroup holds (Lq + Lr above ~40 000) and cannot
                                       be cut into column strips either: an insertion OR a deletion of more than 4 088 bases inside
                                       it (the planner cuts where both planes map 1:1);
                                     - its walk is longer than the path entries reserved for it */

/* return codes */
#define VPR_OK            0
#define VPR_ERR_ARG      -1
#define VPR_ERR_DEVICE   -2   /* no HIP device / HIP runtime error: there is NO CPU fallback */
#define VPR_ERR_NOMEM    -3
#define VPR_ERR_STATE    -4

/* ------------------------------------------------------------------------- */
/* Level A batch: the outputs of generate_ptrs_strs for every supercluster,    */
/* flattened CSR-style (replaces the std::string / vector<vector<int>> locals  */
/* of src/dist.cpp:1784-1822).                                                 */
/* ------------------------------------------------------------------------- */
typedef struct vpr_batch {
    int32_t n_sc;                       /* superclusters in the batch */

    /* hap strings + hap->ref pointers (query1_ref_ptrs ... truth2_ref_ptrs) */
    const int64_t *hap_off[VPR_HAPS];   /* [n_sc+1] element offsets */
    const uint8_t *hap_seq[VPR_HAPS];   /* bases, 1 B each */
    const int32_t *hap_ptr[VPR_HAPS];   /* [PTRS]  index into the ref string */
    const uint8_t *hap_flag[VPR_HAPS];  /* [FLAGS] VPR_PTR_* */

    /* reference string + ref->query-hap pointers (ref_query1_ptrs, ref_query2_ptrs).
       ref->truth pointers are generated by the reference but never read on this
       path (src/dist.cpp:1805,1815), so they are not part of the boundary. */
    const int64_t *ref_off;             /* [n_sc+1] */
    const uint8_t *ref_seq;
    const int32_t *ref_ptr[2];          /* per query hap */
    const uint8_t *ref_flag[2];

    /* variants of each hap slot inside each supercluster, in position order */
    const int64_t *var_off[VPR_HAPS];   /* [n_sc+1] */
    const int32_t *var_pos[VPR_HAPS];   /* poss[v] - begs[sc]  (src/dist.cpp:1075-1080) */
    const float   *var_qual[VPR_HAPS];  /* var_quals[v], already clamped to max_qual */
} vpr_batch;

/* ------------------------------------------------------------------------- */
/* Level B input: variants + reference; the library generates the Level A      */
/* arrays itself (generate_ptrs_strs, src/dist.cpp:145-242).                   */
/* ------------------------------------------------------------------------- */
typedef struct vpr_variants {
    int32_t n_sc;
    int32_t n_ctg;
    const int64_t *ctg_off;             /* [n_ctg+1] offsets into ctg_seq */
    const uint8_t *ctg_seq;             /* concatenated upper-case contigs */
    const int32_t *sc_ctg;              /* [n_sc] contig of each supercluster */
    const int32_t *sc_beg;              /* [n_sc] begs[sc] (inclusive) */
    const int32_t *sc_end;              /* [n_sc] ends[sc] (inclusive, dist.cpp:163); cut at the contig's last base, see
                                           vpr_batch_from_variants */
    const int64_t *var_off[VPR_HAPS];   /* [n_sc+1] */
    const int32_t *var_pos[VPR_HAPS];   /* absolute 0-based contig position */
    const uint8_t *var_type[VPR_HAPS];  /* VPR_TYPE_* */
    const float   *var_qual[VPR_HAPS];
    const int64_t *var_ref_off[VPR_HAPS]; /* [n_var] start of the REF allele in allele_pool */
    const int32_t *var_ref_len[VPR_HAPS]; /* [n_var] (no anchor base; 0 for INS) */
    const int64_t *var_alt_off[VPR_HAPS]; /* [n_var] start of the ALT allele in allele_pool */
    const int32_t *var_alt_len[VPR_HAPS]; /* [n_var] (0 for DEL) */
    const uint8_t *allele_pool[VPR_HAPS];
} vpr_variants;

typedef struct vpr_config {
    int32_t device;            /* HIP device ordinal */
    float   max_qual;          /* g.max_qual          src/globals.h:27 (60) */
    double  credit_threshold;  /* g.credit_threshold  src/globals.h:49 (0.7) */
    double  phase_threshold;   /* g.phase_threshold   src/globals.h:46 (0.6) */
    int64_t workspace_bytes;   /* device scratch budget for flag matrices; 0 = default */
    int32_t band_mode;         /* 0 = dense sweep; 1 = exact windowed sweep: 16-cell window (four
                                  alignments per wave), widened 64/256/1024/dense on a failed exit test;
                                  2 = as 1 but starting at the 64-cell window; 3 = as 1 without the
                                  zero-distance first round.  1 (the default) first tries the 16-cell
                                  zero-distance sweep, which accepts alignments with s = 0 only */
    int32_t flags;             /* VPR_CFG_* (0 = defaults) */
} vpr_config;
#define VPR_CFG_DENSE_S16 1     /* test aid: the dense backward sweep always uses its int16 score rows */
#define VPR_CFG_TIE_SMALL_LOGS 2 /* test aid: the container-order replays start with 32-entry FIFO logs, so that they overflow
                                   and the second attempt (worst-case logs) has to decide the ties */
#define VPR_CFG_GUARD_ALLOC 4   /* test aid: every device array gets an allocation of its own instead of a slice of a pooled
                                   block, so that an access far behind an array faults instead of reading its neighbour */
#define VPR_CFG_HAP_DEDUP 16     /* an alignment whose haplotypes are identical (strings, pointers, flags, variant positions) to those of
                                   another alignment of its supercluster -- a callset that is homozygous there -- is not computed a
                                   second time: it gets that alignment's results (bit for bit what computing it gives; never where
                                   a swap tie can occur).  Off by default: on whole-genome-like input half of the alignments are
                                   such copies, but they are the short ones -- a tenth of the rows (vpr_timing.n_alignments_computed) */
#define VPR_CFG_KEEP_PATHS 8    /* the zero-distance lane kernel (most alignments of whole-genome input end there) keeps its walk in a
                                   compact form only its own credit kernel reads; with this flag it also writes the 16-byte path
                                   entries vpr_download_path returns (tests, callers who want the alignment path itself) */

/* Results: the fields precision_recall_wrapper writes in place
   (ctgVariants::{errtypes,sync_group,credit,ref_ed,query_ed,callq}, src/variant.h:49-60;
    ctgSuperclusters::{sc_phase,orig_phase_dist,swap_phase_dist}, src/cluster.h:39-42).
   Per-variant arrays are indexed [hap slot][swap][variant index in that hap slot].
   Entries no alignment writes keep the reference's initial values
   (ERRTYPE_UN, 0; src/variant.cpp:45-52). */
typedef struct vpr_results {
    int32_t  *aln_dist;        /* [n_sc*4] s[i]               dist.cpp:426 */
    uint8_t  *aln_end_plane;   /* [n_sc*4] pr_query_ref_end   dist.cpp:436-439 */
    uint8_t  *aln_beg_plane;   /* [n_sc*4] pr_query_ref_beg   dist.cpp:811-814 */
    uint32_t *aln_status;      /* [n_sc*4] VPR_ST_* */
    int32_t  *sc_phase;        /* [n_sc] */
    int32_t  *orig_phase_dist; /* [n_sc] */
    int32_t  *swap_phase_dist; /* [n_sc] */
    uint8_t  *errtype[VPR_HAPS][2];
    int32_t  *sync_group[VPR_HAPS][2];
    float    *credit[VPR_HAPS][2];
    int32_t  *ref_ed[VPR_HAPS][2];
    int32_t  *query_ed[VPR_HAPS][2];
    float    *callq[VPR_HAPS][2];
} vpr_results;

/* kernel timing of the last vpr_execute, measured with HIP events on the
   library's own stream */
typedef struct vpr_timing {
    double ms_total;       /* first launch -> last kernel complete */
    double ms_prep;        /* K0: position attributes / string generation */
    double ms_fwd;         /* K1 */
    double ms_bwd;         /* K2 */
    double ms_walk;        /* K3 */
    double ms_ed;          /* K4 */
    int64_t n_fwd_launches;
    int64_t cells_dense;   /* sum_i (Lq_i + Lr) * Lt_i over the batch */
    int64_t cells_touched; /* cells actually computed by K1 (== dense when band_mode 0) */
    int64_t bytes_algorithmic; /* SURVEY 8(d) formula over the batch */
    int64_t n_band_retries;
    int64_t n_tie_replays; /* alignments whose tied swap predecessors were resolved by replaying the reference's
                              container order (VPR_ST_SWAP_TIE), counting second attempts */
    double  ms_tie;        /* the replay kernel */
    /* host side of the last vpr_execute */
    double  ms_wall;       /* wall-clock time of the call */
    double  ms_wall_phase[6]; /* wall-clock time from the start of the call until: [0] inputs reset, first launches possible,
                              [1] round 0 enqueued, [2] retry ladders and tie rounds drained, [3] final tie pass done,
                              [4] deferred edit distances + finalisation enqueued, [5] last kernel complete */
    double  ms_host_alloc; /* time inside hipMalloc / hipFree / hipHostMalloc */
    double  ms_host_blocked; /* time inside blocking waits other than the final one ([4] -> [5]) */
    int64_t n_alignments_computed; /* alignments the kernels ran (all of them without VPR_CFG_HAP_DEDUP) */
    int64_t n_device_allocs, n_device_frees, n_host_allocs;   /* hipMalloc / hipFree / hipHostMalloc calls of the execute: all 0
                              once the handle's workspaces have settled (normally after the first execute of a batch) */
    int64_t n_lane1_seen, n_lane1_finished, n_lane1_waves_dropped;   /* the distance-1 lane level (pr_d1.hip): alignments the zero-distance
                              lane kernel rejected with a complete wave 0, how many of them were finished there (s = 1), and waves of
                              64 that did not fit its blocks (they stay with the 16-cell kernels) */
} vpr_timing;

/* one kernel launch of the last vpr_execute (HIP events around the launch) */
typedef struct vpr_launch_stat {
    int32_t kind;          /* 1 = K1 forward, 2 = K2 backward, 3 = K3 walk, 4 = K4 section edit distance,
                              5 = K5 finalise + phase + tally, 6 = container-order replay of tied alignments */
    int32_t threads;       /* workgroup size */
    int32_t cells_per_thread; /* C of the thread-chunk configuration (K1/K2) */
    int32_t n_units;       /* alignments (K1-K3) or sections (K4) in the launch */
    int64_t cells;         /* cells the launch sweeps: sum (min(W,Lq)+min(W,Lr))*Lt (window W; dense: Lq+Lr) */
    int64_t bytes_algorithmic; /* K1: 1 B/swept cell store + inputs; K2: 1 B/swept cell load; K3/K4: 0 */
    double  ms;
    int64_t cells_dense;   /* dense-equivalent cells of the same alignments: sum (Lq+Lr)*Lt */
    char    kernel[32];    /* name of the (first) kernel of the launch */
} vpr_launch_stat;

typedef struct vpr_handle vpr_handle;

/* lifecycle */
int  vpr_create(const vpr_config *cfg, vpr_handle **out);
void vpr_destroy(vpr_handle *h);
const char *vpr_last_error(const vpr_handle *h);   /* h may be NULL for create errors */
const char *vpr_version(void);
/* the bucket counts this machine's libstdc++ grows std::unordered_set through against the tie replay's model of them (the
   reference's last-writer ties, dist.cpp:347,376, follow that container's iteration order): entries checked, or -(1 + index of
   the first that differs).  vpr_create runs it once per process and refuses to start on a mismatch. */
int vpr_selfcheck_tie_model(int max_entries);

/* one-shot: upload + execute + download (what a reference-side caller uses) */
int vpr_run(vpr_handle *h, const vpr_batch *batch, vpr_results *res);

/* split phases, so a caller (bench.py) can keep the batch resident in HBM */
int vpr_upload(vpr_handle *h, const vpr_batch *batch);          /* host -> HBM */
int vpr_upload_variants(vpr_handle *h, const vpr_variants *v);  /* host marshalling + host -> HBM */
int vpr_execute(vpr_handle *h);                                 /* K1..K4 on the resident batch */
int vpr_download(vpr_handle *h, vpr_results *res);              /* HBM -> host (the results are final on the device) */
/* Optional: page-locked host memory for result and batch buffers.  vpr_download into such buffers runs
   at the link rate instead of the pageable-copy rate.  NULL when there is no HIP device or the allocation fails. */
/* Result buffers in ONE page-locked block laid out like the device's result columns of the uploaded batch: sets every
   pointer of *res into the block and returns the block (release it with vpr_host_free).  vpr_download into such a
   vpr_results is a single copy (178 MB per million superclusters: 3.4 ms instead of 3.9 ms for 55 copies).  The block
   belongs to the caller and stays valid after the next vpr_upload: for a batch of the same shape (same numbers of superclusters
   and of variants per hap slot -- the columns lie at the same offsets) vpr_download still takes the single copy; for any
   other batch the pointers no longer match the layout and it copies column by column as for any other buffers (which then
   have to be large enough for that batch). */
int vpr_results_alloc(vpr_handle *h, vpr_results *res, void **block);
void *vpr_host_alloc(size_t bytes);
void  vpr_host_free(void *p);
/* Multi-GPU processes: make `device` the calling thread's HIP device before anything that allocates without a handle
   (vpr_host_alloc, vpr_batch_from_variants' staging blocks), so that a rank does not create a context on device 0.
   Also note for C callers: the library keeps up to eight HIP streams busy; export GPU_MAX_HW_QUEUES=8 before the
   process initialises HIP (the Python binding and bench.py do), or pairs of them share a hardware queue. */
int   vpr_select_device(int32_t device);
int vpr_get_timing(const vpr_handle *h, vpr_timing *t);
/* test aid (no device): worker threads the library's planning pool starts beside the calling thread when the process may
   use `cpu_quota` CPUs (cgroup quota / local ranks): 0 for a quota of 1 - 3, never more than the quota */
int32_t vpr_test_pool_workers(int32_t cpu_quota);
int vpr_get_launch_stats(const vpr_handle *h, vpr_launch_stat *out, int32_t cap);  /* returns #launches */
/* TP/FP/FN counts [callset QUERY,TRUTH][TP,FP,FN] of the phasing each supercluster's distances select
   (sc_phase SWAP -> swap slot 1, else slot 0), accumulated on the device by the last vpr_execute: the
   per-rank input of the final tally all-reduce */
int vpr_get_tally(const vpr_handle *h, int64_t out[6]);

/* ------------------------------------------------------------------------- */
/* The step right behind the path: per-contig phasing  */
/* of the superclusters and the precision/recall counts.                          */
/* ------------------------------------------------------------------------- */
#define VPR_VARTYPE_SNP   0   /* src/defs.h:40-44 */
#define VPR_VARTYPE_INDEL 1
#define VPR_VARTYPE_SV    2
#define VPR_VARTYPE_ALL   3
#define VPR_VARTYPES      4

/* phaseblockData::phase() for one contig (src/phase.cpp:271-355): Viterbi over the superclusters' sc_phase with a
   unit cost per phase switch inside a phase set and per supercluster that contradicts the chosen phase.
   phase_set[k]: ctgSuperclusters::phase_sets.  pb_phase[n]: phase of each supercluster; switches / flips: supercluster
   indices (ascending, arrays of n entries).  Host code. */
int vpr_phase(const int32_t *sc_phase, const int32_t *phase_set, int32_t n, int32_t *pb_phase,
              int32_t *switches, int32_t *n_switches, int32_t *flips, int32_t *n_flips);

/* The counters of write_precision_recall (src/print.cpp:324-441) from the results of the last vpr_execute, computed on
   the device: counts[callset][VPR_VARTYPES][TP,FP,FN][max_qual-min_qual+1] = variants counted at every quality
   threshold.  var_class[slot][v] in {SNP, INDEL, SV} (INDEL iff the allele is shorter than g.sv_threshold,
   print.cpp:362-372); pb_phase[n_sc] is consulted for superclusters whose sc_phase is NONE (may be NULL: phase 0).
   The reference accumulates these in `float`; the int64 values are identical below 2^24 per cell. */
/* SNP / INDEL / SV class (VPR_VARTYPE_*) of n variants, src/print.cpp:362-372: a SUB is a SNP, an INS / DEL shorter than
   sv_threshold an INDEL, anything else an SV.  Host code. */
void vpr_var_class(const uint8_t *var_type, const int32_t *ref_len, const int32_t *alt_len, int64_t n, int32_t sv_threshold,
                   uint8_t *out);
/* var_class may be NULL after vpr_upload_var_class (classes stay resident for the batch) */
int vpr_upload_var_class(vpr_handle *h, const uint8_t *const var_class[VPR_HAPS]);
int vpr_pr_counts(vpr_handle *h, const uint8_t *const var_class[VPR_HAPS], const int32_t *pb_phase,
                  int32_t min_qual, int32_t max_qual, int64_t *counts);

/* Multi-GPU, one process per GPU: superclusters are independent, so the ranks own disjoint sets and exchange
   nothing on the data path.  What remains of the reference's serial tail:
     - vpr_allreduce_counts: the counters of write_precision_recall (print.cpp:328-438, a serial loop over every variant in the
       reference) summed over all ranks -- vpr_pr_counts with ONE all-reduce (RCCL over xGMI) of the device histogram in between;
       every rank gets the global counts;
     - vpr_allgather_phase: (sc_phase, orig_phase_dist, swap_phase_dist) of ALL superclusters on every rank, for the per-contig
       phasing (phase.cpp:271-355), which needs every supercluster of a contig: sc_index[k] is the global index of this rank's
       k-th supercluster, the three output arrays hold n_total entries (entries no rank owns are left untouched).
   nccl_comm is the caller's ncclComm_t (rccl.h), created with the librccl of the process; the library resolves RCCL at run time
   (no link-time dependency): vpr_rccl_available() says whether it found one.  Collective: every rank of the communicator
   calls, in the same order. */
int vpr_rccl_available(void);
/* the RCCL copy the library bound: the one the process had already mapped when the first collective (or vpr_rccl_available) was
   called -- PyTorch's in a Python process -- else the one it opened by name; "" = the process's global symbols.  The caller's
   communicator must come from the same copy (vcfdist_amd/rccl.py binds by this path). */
const char *vpr_rccl_library(void);
int vpr_allreduce_counts(vpr_handle *h, void *nccl_comm, const uint8_t *const var_class[VPR_HAPS], const int32_t *pb_phase,
                         int32_t min_qual, int32_t max_qual, int64_t *counts);
int vpr_allgather_phase(vpr_handle *h, void *nccl_comm, int32_t n_ranks, const int32_t *sc_index, int32_t n_total,
                        int32_t *sc_phase, int32_t *orig_phase_dist, int32_t *swap_phase_dist);

/* one line of the PRECISION-RECALL SUMMARY (src/print.cpp:497-566) */
typedef struct vpr_pr_row {
    int32_t vartype;       /* VPR_VARTYPE_* */
    int32_t best;          /* 0: threshold NONE (min_qual), 1: BEST (the quality with the highest F1) */
    int32_t qual;
    int32_t truth_tp, query_tp, truth_fn, query_fp;
    float precision, recall, f1_score, f1_qscore;
} vpr_pr_row;
/* rows[2 * VPR_VARTYPES]: NONE and BEST per variant type, in the reference's print order.  Host code. */
int vpr_pr_summary(const int64_t *counts, int32_t min_qual, int32_t max_qual, vpr_pr_row *rows);

/* debug/parity aid: the walk of one alignment (path/sync/edits of
   get_prec_recall_path_sync).  Returns the number of steps, or <0: VPR_ERR_STATE when the walk is no longer in a
   workspace (workspaces are reused by later chunks and later retry rounds).  Arrays must hold cap entries. */
int64_t vpr_download_path(const vpr_handle *h, int32_t sc, int32_t aln, int64_t cap,
        uint8_t *plane, int32_t *qri, int32_t *ti, uint8_t *sync, uint8_t *edit);

/* ------------------------------------------------------------------------- */
/* Host-side marshalling (no device): Level B -> Level A, the equivalent of    */
/* the four generate_ptrs_strs calls at src/dist.cpp:1784-1822.                */
/* ------------------------------------------------------------------------- */
typedef struct vpr_owned_batch vpr_owned_batch;
/* Returns VPR_ERR_ARG for input generate_ptrs_strs cannot process: a variant type other than SUB/INS/DEL
   (its ERROR, dist.cpp:199), unsorted or overlapping variants on a haplotype, a SUB whose alleles are not one base,
   a variant that reaches behind its contig or its supercluster's region, and a region that starts in front of the
   contig (sc_beg < 0: a variant at position 0; the reference's substr(-1) throws there and it exits, dist.cpp:232-238).
   Contig end: get_supercluster_range (cluster.cpp:563-596) sets end = pos + rlen + 1 of the last variant, which is
   >= the contig length when that variant ends on one of the contig's last two bases.  The reference has no defined
   result there -- its substr (dist.cpp:232) silently returns fewer bases than the pointers it has just appended, and
   calc_prec_recall_path starts from the pointer arrays' size (dist.cpp:539-546), outside its matrices.  The library
   evaluates the region that exists: sc_end is cut at the contig's last base, strings and pointer arrays consistent
   (so such a supercluster has one or no base behind its last variant instead of two); the tables keep the caller's
   sc_end. */
int  vpr_batch_from_variants(const vpr_variants *v, vpr_owned_batch **out);
/* (vpr_upload_variants runs the same sizing and checking pass on the host and has the DEVICE write the arrays, from the
   variant tables and the contig: the equivalent of generate_ptrs_strs inside the timed stage, dist.cpp:1786-1822.)
   Test aid: the resident Level A arrays copied back; dst's arrays must hold what the uploaded batch's offsets say. */
int  vpr_download_level_a(vpr_handle *h, vpr_batch *dst);
const vpr_batch *vpr_owned_batch_view(const vpr_owned_batch *b);
void vpr_owned_batch_free(vpr_owned_batch *b);

/* ------------------------------------------------------------------------- */
/* Synthetic workload generator: deterministic */
/* explicit PRNG (splitmix64), no library distributions.                       */
/* ------------------------------------------------------------------------- */
typedef struct vpr_synth_params {
    uint64_t seed;
    int32_t  n_sc;
    int32_t  len_mode;      /* 0 log-uniform[len_a,len_b]; 1 log-normal(median=len_a, sigma=len_b); 2 fixed len_a */
    double   len_a, len_b;
    int32_t  len_min, len_max;   /* clip */
    double   p_repeat;      /* fraction of superclusters drawn from tandem repeats (unit 1-6 bp) */
    double   var_per_base;  /* Poisson mean = max(1, L*var_per_base) sites per supercluster */
    double   p_snp;         /* else indel, geometric length (mean indel_mean, cap L/4) */
    double   indel_mean;
    double   p_hom;         /* homozygous sites */
    double   p_keep, p_drop;/* truth = query site kept / dropped / (rest) perturbed */
    int32_t  max_qual;      /* quals uniform integer 1..max_qual */
    int32_t  reserved;
    double   p_sv;          /* fraction of superclusters that carry one SV-sized indel beside their small variants (joint
                               SNP + INDEL + SV evaluation, BASELINE configs[3]); 0 = none */
    int32_t  sv_min, sv_max;/* its length: log-uniform [sv_min, sv_max]; a deletion lengthens the span by its length (up to len_max) */
} vpr_synth_params;
typedef struct vpr_synth vpr_synth;
void vpr_synth_default_params(vpr_synth_params *p);
int  vpr_synth_create(const vpr_synth_params *p, vpr_synth **out);
const vpr_variants *vpr_synth_variants(const vpr_synth *s);
void vpr_synth_destroy(vpr_synth *s);

/* host helper (no device): store_phase, src/dist.cpp:449-475 */
int32_t vpr_store_phase(const int32_t s[4], double phase_threshold,
                        int32_t *orig_phase_dist, int32_t *swap_phase_dist);

#ifdef __cplusplus
}
#endif
#endif /* VCFDIST_PR_H_ */
