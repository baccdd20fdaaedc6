/* pr_oracle.h -- C entry points of the CPU oracle (TEST INFRASTRUCTURE ONLY;
 * see the header of pr_oracle.cpp).  Shares only the plain-data structs of
 * include/vcfdist_pr.h with the product. */
#ifndef PR_ORACLE_H_
#define PR_ORACLE_H_
#include <stdint.h>
#include "../include/vcfdist_pr.h"
#ifdef __cplusplus
extern "C" {
#endif

/* optional per-alignment diagnostics; any pointer may be NULL */
typedef struct vpo_extra {
    int64_t *swap_writes;               /* [n_sc*4] writes to swap_pred (dist.cpp:347,376) */
    int64_t *swap_conflict_writes;      /* [n_sc*4] writes that replaced a different predecessor */
    int64_t *swap_used_conflict;        /* [n_sc*4] backward-pass reads of a cell that had >1 writer */
    int64_t *swap_used_conflict_nonmax; /* [n_sc*4] ... where the kept writer is not the max-index one */
    int64_t *path_len;                  /* [n_sc*4] */
    /* dump of one alignment's walk */
    int32_t want_sc, want_aln;
    int64_t path_cap, want_len;
    uint8_t *path_plane; int32_t *path_qri; int32_t *path_ti;
    uint8_t *path_sync; uint8_t *path_edit;   /* path_cap+1 entries */
    /* dump of one alignment's matrices, [plane] -> [q][t] row-major */
    uint8_t *dump_flags[2]; uint8_t *dump_pptr[2]; int16_t *dump_pscore[2];
} vpo_extra;

typedef struct vpo_generated vpo_generated;

int vpo_edit_distance(const uint8_t *a, int32_t la, const uint8_t *b, int32_t lb);
int32_t vpo_store_phase(const int32_t s[4], double phase_threshold, int32_t *orig, int32_t *swp);
int vpo_run(const vpr_batch *b, const vpr_config *cfg, vpr_results *res, vpo_extra *ex);

vpo_generated *vpo_gen_create(const vpr_variants *v);
int vpo_gen_error(const vpo_generated *g);
void vpo_gen_sizes(const vpo_generated *g, int64_t hap_len[VPR_HAPS], int64_t *ref_len);
void vpo_gen_copy(const vpo_generated *g, int64_t *hap_off[VPR_HAPS], uint8_t *hap_seq[VPR_HAPS],
                  int32_t *hap_ptr[VPR_HAPS], uint8_t *hap_flag[VPR_HAPS], int64_t *ref_off,
                  uint8_t *ref_seq, int32_t *ref_ptr[2], uint8_t *ref_flag[2]);
void vpo_gen_free(vpo_generated *g);

#ifdef __cplusplus
}
#endif
#endif
