// pr_device.h -- device-side data layout shared by the kernels and the host API.
#ifndef PR_DEVICE_H_
#define PR_DEVICE_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

// forward-pass flag byte (reference defs.h:110-120 for the low 5 bits)
#define F_INS 1
#define F_DEL 2
#define F_MAT 4
#define F_SUB 8
#define F_SWP 16
#define F_CHOICE_SHIFT 5      // bits 5-6: rank of the chosen swap source within the cell's candidate list (low two bits)
#define F_TIE 128             // bit 7: more than one optimal swap source (order-defined in the reference)
// A position keeps up to eight allowed swap sources (cand_* + cand2_*), so the rank has three bits.  Its third bit sits in
// F_SUB's place: a swap needs equal bases at its target cell (dist.cpp:335-350), and where the bases are equal the
// diagonal move is a match, never a substitution (D(cell) <= D(diagonal predecessor) < D(diagonal predecessor) + 1), so
// F_SUB and F_SWP never meet in one byte.  Readers of the diagonal bits therefore go through f_diag().
#define F_CHOICE_MASK (0x60u | F_SUB)
__host__ __device__ __forceinline__ uint32_t f_choice_bits(int rank) { return (uint32_t(rank & 3) << F_CHOICE_SHIFT) | (uint32_t(rank >> 2) << 3); }
__host__ __device__ __forceinline__ int f_choice_of(uint32_t f) { return int(((f >> F_CHOICE_SHIFT) & 3u) | ((f >> 1) & 4u)); }   // of a byte with F_SWP
__host__ __device__ __forceinline__ uint32_t f_diag(uint32_t f) { return f & (F_MAT | (F_SUB & ~(f >> 1))); }    // F_MAT / F_SUB of a forward flag byte
// a source's rank in its swap target's list, in the constant bytes of the backward sweeps (bk_* >> 24, the dense kc):
// bits 1-2 and bit 4
__host__ __device__ __forceinline__ uint32_t rank_bits(int rank) { return (uint32_t(rank & 3) << 1) | (uint32_t(rank >> 2) << 4); }
__host__ __device__ __forceinline__ int rank_of(uint32_t bits) { return int(((bits >> 1) & 3u) | ((bits >> 2) & 4u)); }
// "z's forward flags say: entered by a swap from the source of this rank" as one masked compare
#define F_SWP_KEY_MASK (F_SWP | F_CHOICE_MASK)
__host__ __device__ __forceinline__ uint32_t f_swp_key(int rank) { return F_SWP | f_choice_bits(rank); }
#define SWAP_SOURCES_MAX 8
// The backward sweeps replace a cell's forward flags with its path_ptr byte (low 5 bits; 0 = not on an optimal path) and
// keep bits 5-7 of the forward flags for cells that are on one: a "used" tied cell is then (byte & F_TIE) != 0 with
// bwd_allow(the cell's pointer flag), which is what the container-order replay looks for (pr_tie.hip); their number is
// added to AlnOut::n_sec of a marked alignment (its walk, which sets n_sec, has not run yet).
#define F_KEEP 0xe0

#define PV 1   // PTR_VARIANT
#define PB 2   // PTR_VAR_BEG
#define PE 4   // PTR_VAR_END
#define PI 8   // PTR_INS_LOC

// AlnOut::band_ok of an alignment whose backward sweep consulted a tied swap cell (F_TIE): the walk / credit kernels of
// the round skip it (they require band_ok == tag) and the host's tie pass (pr_tie.hip) picks it up
#define TIE_MARK(tag) (-(int32_t((tag) & 0x3fff) + 1))
// level tag of a tie round's descriptors and accept test: the level's own tag with this bit, so that the kernels of the
// round that marked the alignment (which may still be running) never pick up its new descriptor
#define TIE_TAG_BIT 0x4000

#define D_INF 0x3f000000
#define S_NEG (-(1 << 28))

// Level A batch resident in HBM (same CSR layout as vpr_batch) + derived position attributes
struct DevBatch {
    int32_t n_sc;
    const int64_t *hap_off[4];
    const uint8_t *hap_seq[4];
    const int32_t *hap_ptr[4];
    const uint8_t *hap_flag[4];
    const int64_t *ref_off;
    const uint8_t *ref_seq;
    const int32_t *ref_ptr[2];
    const uint8_t *ref_flag[2];
    const int64_t *var_off[4];
    const int32_t *var_pos[4];
    // derived by k_prep_*:
    int32_t *sc_hap[4];   // supercluster of every hap position / ref position (k_prep_scof): the other K0 kernels are
    int32_t *sc_ref;      //   one thread per position and need its supercluster's offsets
    int4 *cand_q[2];      // [hap positions of query hap h] allowed swap sources in the REF plane (ascending, -1 pad)
    int4 *cand_r[2];      // [ref positions]               allowed swap sources in QUERY hap h
    int4 *cand2_q[2];     // sources five to eight of a position (read only where cand_*.w >= 0: directly adjacent separate
    int4 *cand2_r[2];     //   indel records on one haplotype, each of which ends at the position)
    uint8_t *has_ins[4];  // [ref positions] an insertion of hap slot s sits at this ref index (dist.cpp:886-894)
    uint8_t *sc_limit;    // [superclusters] nonzero: the supercluster exceeds an implementation limit (more than eight swap sources on
                          //   one position): its alignments are reported with VPR_ST_ERR_LIMIT and no variant results
    // packed per-position constants for the banded kernels (k_prep_pack):
    //   fk_*: .x = first swap source (cand.x) | FK_MULTI if there are more, -1 if none
    //         .y = swap target of this position as a *source* (ptr+1 if fwd_allow, else 0xffffff) | base << 24
    //   bk_*: swap target z (24 bits, 0xffffff none; includes fwd_allow(src), bwd_allow(z), membership in z's
    //         candidate list) | tp(this cell, QUERY plane only) << 24 | rank_bits(rank in z's list) << 24 | tp(z) << 27
    int2 *fk_q[2];        // [hap positions of query hap h]
    int2 *fk_r[2];        // [ref positions]
    int32_t *bk_q[2];
    int32_t *bk_r[2];
    // suffix sums of indel sizes (k_prep_suffix): vs_hap[s][i] = total size of the indels of hap slot s at hap
    // positions >= i; vs_ref[h][r] = total size of query hap h's indels at ref positions >= r.  They bound how
    // far the diagonal can still shift for free, which turns an exit cell's distance into a lower bound of any
    // path through it (pr_band.hip, k_fwd_stripe).
    int32_t *vs_hap[4];
    int32_t *vs_ref[2];
    // Budgets of the exit test of the 64-cell and wider window kernels (k_prep_xb; the derivation is at exit_key, pr_band.hip).
    // A budget is a pair packed into one int: I | Dl << 16 = the inserted / the deleted bases among the query-hap steps a path
    // can still cross (0xffff: that or more).  W(i) = the pair over the steps into hap positions >= i.
    //   Bref(x) = W(r2q[x] + 1), or W(r2q[x] + 2) when x is a deleted base: what a path that is on the REF plane at x can
    //             still cross on the QUERY plane (a deletion is only crossed by a swap from in front of it)
    //   xb_q[h][x] = {W(x + 1), Bref(q2r[x] + 1)}     a QUERY cell's own budget / its swap target's
    //   xb_r[h][x] = {Bref(x), Bref(x + 1)}           a REF cell's own budget / its diagonal successor's
    int2 *xb_q[2];
    int2 *xb_r[2];
    // per supercluster and hap slot: {min, max} of position - pointer over the hap's positions (k_prep_suffix).  A cell
    // whose plane and truth positions map to the same reference base sits on diagonal q - t = (q - q2r[q]) - (t - t2r[t]):
    // the host sizes the stamp grids of the tie replay (pr_tie.hip) from these ranges
    int2 *dspan[4];
    // packed constants of the 16-cell window kernels (pr_q16.hip, k_prep_q16):
    //   fk4_*: {fk.x, fk.y, reference coordinate of the position, free-shift budget behind it}
    //   tk[s]: truth slot 2+s: {t2r[t], base | fwd_allow(flag[t-1]) << 8 | vs_hap[t-1] << 9}
    int4 *fk4_q[2];
    int4 *fk4_r[2];
    int2 *tk[2];
    // truth slot 2+s: how many positions in front of t share t's reference pointer, i.e. t's offset inside an insertion
    // (0 outside; saturates at 65535).  The window origins follow it: where the query hap carries an insertion at the same
    // reference position the optimal path runs along both, not past the query's (k_prep_tj, q16_center)
    uint16_t *tj[2];
    // packed constants of the row-sweep walk (k_walk_q16).  ins4(r) = bit s set iff an insertion of hap slot s
    // sits at ref index r (has_ins[s][r]); an alignment looks at the bits of its query and truth slot.
    //   wk_q[h]: {q2r[x], flag[x] | ins4(q2r[x]) << 8}     wk_r[h]: {r2q_h[x], ins4(x) << 8}
    //   wk_t[s]: truth slot 2+s: {t2r[t], flag[t] | ins4(t2r[t]) << 8}
    int2 *wk_q[2];
    int2 *wk_r[2];
    int2 *wk_t[2];
};
#define FK_MULTI (1 << 30)
#define FK_NONE24 0xffffff

// one (supercluster, alignment) work unit
struct AlnDesc {
    int64_t q_off, t_off, r_off;   // element offsets of the query hap / truth hap / ref strings
    int32_t Lq, Lt, Lr;
    int32_t qs, ts;                // hap slots (qs = i>>1, ts = 2 + (i&1))
    int32_t sc, aln;
    int32_t pitch[2];              // row pitch (bytes) of the [Lt][pitch] flag matrices, QUERY / REF plane
    int64_t mat_off[2];            // byte offsets into the flag workspace
    int64_t path_off;              // entry offset into the path scratch
    int64_t sec_off;               // entry offset into the section table
    int32_t sec_cap;
    int32_t path_cap;
    int32_t band_w;                // 0: dense [Lt][pitch] matrices; >0: banded, row t holds cells [blo[t], blo[t]+band_w)
    int32_t band_pad;              // level tag of the round this descriptor belongs to (= band_w, except 8 for the
                                   // zero-distance 16-cell level); kernels of other rounds skip the alignment
    int64_t blo_off;               // int offset of this alignment's band origins: blo[plane*Lt + t]
    int64_t qv_beg, qv_end, tv_beg, tv_end;   // variant index ranges (batch-global, per hap slot)
};

// per-alignment scalars produced by the kernels
struct AlnOut {
    int32_t dist_q, dist_r;   // D at the two end cells after the forward sweep
    int32_t s;                // min of the two
    int32_t end_plane;        // 0 QUERY / 1 REF      (dist.cpp:436-439)
    int32_t beg_plane;        // (dist.cpp:811-814)
    uint32_t status;          // VPR_ST_*
    int32_t path_len;
    int32_t n_sec;
    int32_t exit_min;         // banded forward sweep: min D over cells with an edge leaving the band
    int32_t band_ok;          // exit_min > s  =>  the band provably contains every cell with D <= s
};

// one sync section that contains variants, dist.cpp:1190-1373 (32 bytes)
struct Section {
    int32_t q_lo, q_hi;       // query variant index range (q_lo, q_hi], i.e. query_var_ptr+1 .. prev_query_var_ptr
    int32_t t_lo, t_hi;       // truth variant index range likewise
    int32_t sync_group;
    int32_t query_ed;
    int32_t ref_ed;           // valid unless deferred
    int32_t flags;            // bit0: deferred to k_ed; bit1: no variants (kept only for the WARN checks)
};
#define SEC_DEFERRED 1
#define SEC_NOVAR 2

// a deferred wf_ed call: both segments longer than the inline limit
struct EdJob {
    int32_t aln, sec;         // alignment id, section index within the alignment
    int32_t ref_beg, ref_len; // ref segment (index into this alignment's ref string)
    int32_t tru_beg, tru_len; // truth segment
};

// per-variant result columns of one (hap slot, swap) -- ctgVariants::{errtypes,sync_group,credit,ref_ed,
// query_ed,callq}[swap], variant.h:49-60
struct VarCols {
    uint8_t *errtype;
    int32_t *sync_group;
    float *credit;
    int32_t *ref_ed;
    int32_t *query_ed;
    float *callq;
};
struct DevResults {
    VarCols v[4][2];          // [hap slot][swap]
    const float *var_qual[4];
    int32_t *aln_dist;        // [n_sc*4]
    uint8_t *aln_end_plane, *aln_beg_plane;
    uint32_t *aln_status;
    int32_t *sc_phase, *orig_phase_dist, *swap_phase_dist;   // [n_sc]
    unsigned long long *tally;   // [2 callsets][3 errtypes]
    float max_qual;
    double credit_threshold, phase_threshold;
};

// packed walk entry: a = qri | plane<<31,  b = ti (28 bits) | sync<<31 | edit<<30 | base-equality bits 28, 29 (credit_walk), plus the reference coordinates of
// the cell (qref = qri on the REF plane, q2r[qri] on the QUERY plane; tref = t2r[ti]) so the backward credit
// walk needs no pointer-array loads
struct PathEnt { uint32_t a, b; int32_t qref, tref; };

// ---------------------------------------------------------------------------
// Descriptors as a function of the batch offsets and a plan's (alignment, workspace offset) pair.  The host planner and
// the device builder (k_build_plan, pr_api.hip) share these, so a round-0 plan crosses the link as 12 bytes per
// alignment (its place in the work list and its workspace offset) instead of a 96-byte descriptor.
// ---------------------------------------------------------------------------
enum { LV_Z = 0, LV_Q16 = 1, LV_C1 = 2, LV_C4 = 3, LV_C16 = 4, LV_DENSE = 5 };
__host__ __device__ inline int lv_window(int lv) { return lv <= LV_Q16 ? 16 : lv == LV_C1 ? 64 : lv == LV_C4 ? 256 : lv == LV_C16 ? 1024 : 0; }
__host__ __device__ inline int lv_tag(int lv) { return lv == LV_Z ? 8 : lv_window(lv); }
__host__ __device__ inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

struct BatchOffsets {       // CSR offsets of a batch (host copies or the resident ones)
    const int64_t *hap_off[4];
    const int64_t *ref_off;
    const int64_t *var_off[4];
};

// the descriptor of alignment a = 4 * supercluster + i without a workspace layout (query hap i >> 1 against truth hap i & 1)
__host__ __device__ inline AlnDesc base_desc(const BatchOffsets &O, int64_t a) {
    AlnDesc d{};
    const int sc = int(a >> 2), i = int(a & 3);
    d.qs = i >> 1; d.ts = 2 + (i & 1);
    d.sc = sc; d.aln = i;
    d.q_off = O.hap_off[d.qs][sc]; d.t_off = O.hap_off[d.ts][sc]; d.r_off = O.ref_off[sc];
    d.Lq = int32_t(O.hap_off[d.qs][sc + 1] - d.q_off);
    d.Lt = int32_t(O.hap_off[d.ts][sc + 1] - d.t_off);
    d.Lr = int32_t(O.ref_off[sc + 1] - d.r_off);
    int64_t nv[4], v0 = 0;
    for (int s = 0; s < 4; s++) { nv[s] = O.var_off[s][sc + 1] - O.var_off[s][sc]; v0 += O.var_off[s][sc]; }
    d.qv_beg = O.var_off[d.qs][sc]; d.qv_end = O.var_off[d.qs][sc + 1];
    d.tv_beg = O.var_off[d.ts][sc]; d.tv_end = O.var_off[d.ts][sc + 1];
    // section table: alignment (q, t) holds (its query's + its truth's variants + 4) entries, a supercluster's four
    // alignments therefore 2 * (its variants) + 16, one after the other
    d.sec_cap = int32_t(nv[d.qs] + nv[d.ts] + 4);
    int64_t sec = 2 * v0 + 16 * int64_t(sc);
    for (int j = 0; j < i; j++) sec += nv[j >> 1] + nv[2 + (j & 1)] + 4;
    d.sec_off = sec;
    d.path_cap = d.Lq + d.Lr + d.Lt + 4;
    return d;
}

// Workspace layout of an alignment at window level dl (not LV_DENSE): flag matrices, window origins, walk scratch.
// Fills the layout fields of d for workspace offset `used` and returns the bytes the alignment occupies.
__host__ __device__ inline int64_t window_layout(AlnDesc &d, int dl, int64_t used, int tag_or) {
    const int W = lv_window(dl);
    int64_t m0, m1, bl;
    if (dl <= LV_Q16) {
        // stripe-transposed records of 128 B per 4 truth rows (both planes) + int2 origins per stripe
        const int64_t nstr = (int64_t(d.Lt) + 3) / 4;
        d.band_w = 16;
        d.pitch[0] = d.pitch[1] = 16;
        m0 = nstr * 128; m1 = 0;
        bl = round_up64(nstr * 8, 128);
    } else {
        d.band_w = W;
        d.pitch[0] = int32_t(round_up64(W < d.Lq ? W : d.Lq, 16));
        d.pitch[1] = int32_t(round_up64(W < d.Lr ? W : d.Lr, 16));
        m0 = round_up64(int64_t(d.pitch[0]) * d.Lt, 128); m1 = round_up64(int64_t(d.pitch[1]) * d.Lt, 128);
        bl = round_up64(int64_t(2) * d.Lt * 4, 128);
    }
    // (every piece a multiple of 128 bytes: the path block of an alignment starts on a 128-byte line, which the credit
    // walk's refills rely on -- the path reads past path_cap inside the block's padding)
    const int64_t pb = round_up64(int64_t(d.path_cap) * int64_t(sizeof(PathEnt)) + 128, 128);   // 16 B per step
    d.band_pad = lv_tag(dl) | tag_or;
    d.mat_off[0] = used;
    d.mat_off[1] = used + m0;
    d.blo_off = (used + m0 + m1) / 4;            // int index into the arena
    d.path_off = (used + m0 + m1 + bl) / int64_t(sizeof(PathEnt));
    return m0 + m1 + bl + pb + 128;
}

// VPR_CFG_HAP_DEDUP: identical haplotypes of a call set make alignments of a supercluster identical (k_hap_alias, pr_api.hip);
// source of alignment i (0..3) of a supercluster with alias bits b, -1: it is computed itself
__host__ __device__ inline int alias_source(int b, int i) {
    const int q = i >> 1, t = i & 1;
    const int qs = (b & 1) ? 0 : q, ts = (b & 2) ? 0 : t;
    const int src = qs * 2 + ts;
    return src == i ? -1 : src;
}

// the lane levels (pr_zl.hip, pr_d1.hip): per wave of 64 alignments: offsets (in 32-bit words) of its interleaved input block and of its log blocks
struct ZlWave {
    int64_t in_off;       // Q block; R block at + 64 * mq, T block at + 64 * (mq + mr)
    int64_t log_off;      // in uint4 units; 80 * mt of them: the cells' flag bytes (8 B per row and lane), one path_ptr word per
                          // row and lane, the walk's steps (8 B per row and lane)
    int32_t mq, mr, mt;   // largest Lq / Lr / Lt of the wave's alignments
    int32_t pad;
};
struct TieJob;      // a replay job of the container-order replay (pr_tie.hip)

#endif
