// pr_band.hip -- exact *banded* single-wave kernels of the precision/recall path.
//
// The reference's wave expansion touches only cells with distance <= s (0.24 % of the dense
// matrix on its own demo).  These kernels sweep, per row of the truth string, only a
// window of W = 64*C cells per plane centred on the reference coordinate of that truth base
// (REF plane: t2r[t]; QUERY plane: r2q[t2r[t]]), one wavefront per alignment, no workgroup barrier,
// distances and per-position constants in small LDS rings.
//
// Exactness does not rest on the window heuristic: while sweeping, every cell that has a graph edge
// (MAT/SUB, INS, DEL or plane swap) leaving the window contributes its distance to `exit_min`.
// Any cell outside the window whose true distance is <= s would have an optimal path whose last
// in-window cell is such an exit cell with distance <= s, so
//        exit_min > s   =>   the window contains every cell with true distance <= s,
// hence s, the end plane and the flag byte of every cell with D <= s equal the dense result.  When
// the test fails the host re-runs the alignment with a 4x wider window and finally with the dense
// kernels (pr_kernels.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcfdist_pr.h"
#include "pr_device.h"
#include "pr_scan.h"

// ---------------------------------------------------------------------------
// K0b: packed per-position constants (needs the candidate lists of k_prep_cand)
// ---------------------------------------------------------------------------
__global__ void k_prep_pack(DevBatch B, int h, int dir, int64_t n_pos) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= n_pos) return;
    // this plane (the positions being packed) and the other plane (swap targets)
    const int64_t *off = dir == 0 ? B.hap_off[h] : B.ref_off;
    const int64_t *ooff = dir == 0 ? B.ref_off : B.hap_off[h];
    const int32_t *ptr = dir == 0 ? B.hap_ptr[h] : B.ref_ptr[h];
    const uint8_t *flg = dir == 0 ? B.hap_flag[h] : B.ref_flag[h];
    const uint8_t *oflg = dir == 0 ? B.ref_flag[h] : B.hap_flag[h];
    const uint8_t *seq = dir == 0 ? B.hap_seq[h] : B.ref_seq;
    const int4 *cand = dir == 0 ? B.cand_q[h] : B.cand_r[h];      // sources of swaps *into* this position
    const int4 *ocand = dir == 0 ? B.cand_r[h] : B.cand_q[h];     // candidate lists of the other plane
    int2 *fk = dir == 0 ? B.fk_q[h] : B.fk_r[h];
    int32_t *bk = dir == 0 ? B.bk_q[h] : B.bk_r[h];

    const int sc = (dir == 0 ? B.sc_hap[h] : B.sc_ref)[g];
    const int32_t x = int32_t(g - off[sc]);
    const int64_t olen = ooff[sc + 1] - ooff[sc];
    const int f = flg[g];
    const int32_t p = ptr[g];
    const bool fa = !(f & PV) || (f & PE);
    const int4 cc = cand[g];
    int2 k;
    k.x = cc.x < 0 ? -1 : (cc.x | (cc.y >= 0 ? FK_MULTI : 0));
    k.y = int32_t((fa ? uint32_t(p + 1) & 0xffffffu : 0xffffffu) | (uint32_t(seq[g]) << 24));
    fk[g] = k;

    // backward constants
    uint32_t z = FK_NONE24, bits = 0;
    auto tp_of = [&](const int32_t *qptr, const uint8_t *qflg, int64_t base, int32_t q) -> uint32_t {
        return (q > 0 && ((qptr[base + q] != qptr[base + q - 1] + 1) || (qflg[base + q] & PB))) ? 1u : 0u;
    };
    if (dir == 0) bits |= tp_of(ptr, flg, off[sc], x);            // tp of this QUERY-plane cell
    const int64_t zz = int64_t(p) + 1;
    if (fa && zz >= 1 && zz < olen) {
        const int zf = oflg[ooff[sc] + zz];
        if (!(zf & PV) || (zf & PB)) {                            // bwd_allow(z), dist.cpp:600-601,638-639
            const int4 oc = ocand[ooff[sc] + zz];
            int rank = -1;
            if (oc.x == x) rank = 0; else if (oc.y == x) rank = 1; else if (oc.z == x) rank = 2; else if (oc.w == x) rank = 3;
            else if (oc.w >= 0) {
                const int4 o2 = (dir == 0 ? B.cand2_r[h] : B.cand2_q[h])[ooff[sc] + zz];
                if (o2.x == x) rank = 4; else if (o2.y == x) rank = 5; else if (o2.z == x) rank = 6; else if (o2.w == x) rank = 7;
            }
            if (rank >= 0) {
                z = uint32_t(zz);
                bits |= rank_bits(rank);
                if (dir == 1) bits |= tp_of(B.hap_ptr[h], B.hap_flag[h], ooff[sc], int32_t(zz)) << 3;  // tp(z), z on QUERY
            }
        }
    }
    bk[g] = int32_t(z | (bits << 24));
}

// K0e: tj (see pr_device.h): t2r is non-decreasing inside a supercluster, so the first position with t's pointer is a
// lower bound search
__global__ void k_prep_tj(DevBatch B, int s, int64_t n_pos) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= n_pos) return;
    const int32_t *ptr = B.hap_ptr[2 + s];
    uint16_t j = 0;
    if ((B.hap_flag[2 + s][g] & PV) && g > 0 && ptr[g - 1] == ptr[g]) {
        int64_t a = B.hap_off[2 + s][B.sc_hap[2 + s][g]], b = g;      // first position in [a, b] whose pointer equals ptr[g]
        const int32_t p = ptr[g];
        while (a < b) { const int64_t m = (a + b) >> 1; if (ptr[m] < p) a = m + 1; else b = m; }
        j = uint16_t(min<int64_t>(g - a, 65535));
    }
    B.tj[s][g] = j;
}

// query-plane coordinate the optimal path is expected at when it is at truth position t: the query position of t's
// reference base, plus t's offset inside a truth insertion as far as the query hap has inserted bases there too
__device__ __forceinline__ int query_center(const int32_t *__restrict__ t2r, const uint16_t *__restrict__ tj, const int32_t *__restrict__ r2q,
                                            int t, int Lr) {
    const int r = min(max(t2r[t], 0), Lr - 1);
    const int q = r2q[r];
    const int j = tj[t];
    if (j == 0) return q;
    const int kq = (r + 1 < Lr) ? r2q[r + 1] - q - 1 : 0;      // bases the query hap inserts behind r
    return q + min(j, max(kq, 0));
}

// suffix sums of |pointer step - 1| (an inserted base contributes 1, a crossed deletion its length):
// which = 0..3: hap slot (hap -> ref pointers), 4..5: ref -> query hap (which - 4)
#define SUFFIX_SC_PER_WAVE 8
__global__ void __launch_bounds__(256) k_prep_suffix(DevBatch B) {
    // blockIdx.y: the array.  One WAVEFRONT per supercluster (eight superclusters after each other): lane l of step j holds
    // position L - 1 - 64 j - l, so the loads are whole lines and the suffix sum is a wave scan plus a carry -- a thread per
    // supercluster walked its own chain of dependent loads, and the launch lasted as long as the longest supercluster of the batch
    const int which = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t *off = which < 4 ? B.hap_off[which] : B.ref_off;
    const int32_t *ptr = which < 4 ? B.hap_ptr[which] : B.ref_ptr[which - 4];
    int32_t *out = which < 4 ? B.vs_hap[which] : B.vs_ref[which - 4];
    const int sc0 = (blockIdx.x * 4 + wave) * SUFFIX_SC_PER_WAVE;
    for (int k = 0; k < SUFFIX_SC_PER_WAVE; k++) {
        const int sc = sc0 + k;
        if (sc >= B.n_sc) return;
        const int64_t b = off[sc];
        const int L = int(off[sc + 1] - b);
        int32_t carry = 0;
        int dmin = 0x7fffffff, dmax = -0x7fffffff;
        for (int top = L; top > 0; top -= 64) {
            const int i = top - 1 - lane;              // this lane's position (descending over the lanes)
            int w = 0, p = 0;
            if (i >= 0) {
                p = ptr[b + i];
                if (i > 0) { const int d = p - ptr[b + i - 1] - 1; w = d < 0 ? -d : d; }
                const int dd = i - p;
                dmin = min(dmin, dd); dmax = max(dmax, dd);
            }
            int sfx = w;                               // inclusive scan over the lanes = suffix sum over the positions
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(sfx, o);
                if (lane >= o) sfx += t;
            }
            if (i >= 0) out[b + i] = carry + sfx;
            carry += __shfl(sfx, 63);
        }
        if (which < 4) {
#pragma unroll
            for (int o = 32; o; o >>= 1) { dmin = min(dmin, __shfl_xor(dmin, o)); dmax = max(dmax, __shfl_xor(dmax, o)); }
            if (lane == 0) B.dspan[which][sc] = make_int2(dmin, dmax);
        }
    }
}

// exact budgets of the exit test (pr_device.h: xb_q / xb_r), needs vs_hap of k_prep_suffix.  dir 0: hap positions of
// query hap h, dir 1: ref positions.  A budget is a PAIR: the inserted bases (steps of q2r that stay on a reference base) and
// the deleted bases (steps that jump) among the query-hap steps a path can still cross -- the first lower the reference
// coordinate a path is ahead by, the second raise it, and the exit test prices the two directions apart (exit_key below).
// Both come from the sums at hand: |steps| summed is vs_hap, the signed sum telescopes to pointer and position differences.
__device__ __forceinline__ int xb_pack(int a_abs, int sg) {
    const int del = (a_abs + sg) >> 1, ins = (a_abs - sg) >> 1;
    return int(uint32_t(min(max(ins, 0), 0xffff)) | (uint32_t(min(max(del, 0), 0xffff)) << 16));      // 0xffff: "that or more" (exit_key)
}
__global__ void k_prep_xb(DevBatch B, int h, int dir, int64_t n_pos) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= n_pos) return;
    const int64_t *off = dir == 0 ? B.hap_off[h] : B.ref_off;
    const int sc = (dir == 0 ? B.sc_hap[h] : B.sc_ref)[g];
    const int64_t qo = B.hap_off[h][sc], ro = B.ref_off[sc];
    const int Lq = int(B.hap_off[h][sc + 1] - qo), Lr = int(B.ref_off[sc + 1] - ro);
    const int32_t *W = B.vs_hap[h] + qo, *q2r = B.hap_ptr[h] + qo, *r2q = B.ref_ptr[h] + ro;
    const uint8_t *rflag = B.ref_flag[h] + ro;
    // the steps INTO hap positions >= i: |step - 1| summed = W[i], (step - 1) summed = (q2r[Lq - 1] - q2r[i - 1]) - (Lq - i)
    auto Wq = [&](int i) -> int {
        if (i >= Lq) return 0;
        i = max(i, 1);              // (position 0 has no step in front of it: W[0] = W[1])
        if (i >= Lq) return 0;
        return xb_pack(W[i], (q2r[Lq - 1] - q2r[i - 1]) - (Lq - i));
    };
    auto Bref = [&](int x) -> int {
        if (x >= Lr) return 0;
        x = max(x, 0);
        const int hq = r2q[x];
        const bool deleted = (rflag[x] & PV) && (hq < 0 || hq >= Lq || q2r[hq] != x);
        return Wq(hq + 1 + (deleted ? 1 : 0));
    };
    const int x = int(g - off[sc]);
    if (dir == 0) B.xb_q[h][g] = make_int2(Wq(x + 1), Bref(q2r[x] + 1));
    else B.xb_r[h][g] = make_int2(Bref(x), Bref(x + 1));
}

// (the DPP wave scans -- dpp_mov, wave_prefix_min2, wave_shr1 -- are in pr_scan.h)

// band origin of row t: REF plane centred on t2r[t], QUERY plane on r2q[t2r[t]]
template <int W>
__device__ __forceinline__ void band_origin(const int32_t *t2r, const int32_t *r2q, int t, int Lt, int Lq, int Lr,
                                            int &loQ, int &loR) {
    loQ = 0; loR = 0;
    if (t < Lt) {
        const int tr = t2r[t];
        loR = max(0, min(tr - W / 2, Lr - min(W, Lr)));
        const int cq = r2q[min(max(tr, 0), Lr - 1)];
        loQ = max(0, min(cq - W / 2, Lq - min(W, Lq)));
    }
}

// s, end plane (prefer QUERY, dist.cpp:436-439) and the window acceptance test.  band_ok holds the window
// width W the alignment was accepted at (0: rejected): the later kernels of a round process an alignment only
// if band_ok and its descriptor both carry their own W, so a retry round that re-plans a rejected alignment
// (new descriptor, wider window) may run concurrently with the rest of the round that rejected it.
__global__ void k_fwd_band_finish(const int32_t *__restrict__ work, int n, AlnOut *__restrict__ outs, int W /* level tag */,
                                  const int32_t *__restrict__ n_dev) {
    if (n_dev) n = min(n, *n_dev);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || work[i] < 0) return;   // (-1: padding of a device-built work list)
    AlnOut &o = outs[work[i]];
    o.s = min(o.dist_q, o.dist_r);
    o.end_plane = (o.dist_q <= o.dist_r) ? VPR_PLANE_QUERY : VPR_PLANE_REF;
    o.band_ok = (o.s < D_INF / 2 && o.exit_min > o.s) ? W : 0;
}

// ---------------------------------------------------------------------------
// K2b: banded backward max-TP sweep, one wavefront per alignment, band-relative flag rows.
// Lanes are mirrored (lane 0 owns the highest cells of the window) so that the suffix composition of
// the max-plus maps carrying the in-row INS chain is a *prefix* scan in lane order and can use the same
// DPP pattern as the forward kernel.
// ---------------------------------------------------------------------------
// Inclusive prefix composition (lane order) of two independent sequences of max-plus maps
//   x -> max(A, x + B)      (B < 0: "link broken", the map is the constant A; A = S_NEG: unreachable),
// compose(cur, prev) = cur after prev.  Inside the scan the pairs are re-encoded so that 0 is the identity of
// both operations and a DPP source that does not exist (bound_ctrl -> reads 0) needs no special case:
//   A' = A + MP_OFF for a reachable score, a value < MP_OFF (0..64) for an unreachable one;
//   B' = B for a live link, -2*MP_OFF for a broken one (64 of them still fit in an int32).
// One step is then   t = dpp(A') + B';  B' += dpp(B');  A' = max(A', t)   = 3 VALU per sequence, written as
// asm because hipcc expands the builtin form to 3-4x as many instructions (and, not seeing the DPP reads,
// cannot place the two wait states a DPP read needs after a VALU write of the same register).
#define MP_OFF (1 << 20)
__device__ __forceinline__ void wave_prefix_mp2(MP &a, MP &b) {
    int Aq = (a.A < 0) ? 0 : a.A + MP_OFF, Bq = (a.B < 0) ? -2 * MP_OFF : a.B;
    int Ar = (b.A < 0) ? 0 : b.A + MP_OFF, Br = (b.B < 0) ? -2 * MP_OFF : b.B;
    int tq, tr;
#define MP_STEP(CTRL)                                                                           \
        "v_add_u32_dpp %4, %0, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
        "v_add_u32_dpp %5, %2, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
        "v_add_u32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
        "v_add_u32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
        "v_max_i32 %0, %0, %4\n\t"                                                              \
        "v_max_i32 %2, %2, %5\n\t"                                                              \
        "s_nop 0\n\t"
    // row_bcast steps: lanes of the rows that are not selected are disabled; their stale t is <= A'
    // (A' already absorbed it in the previous step), so the following v_max is a no-op for them
#define MP_BCAST(CTRL, RM)                                                                      \
        "v_add_u32_dpp %4, %0, %1 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"                   \
        "v_add_u32_dpp %5, %2, %3 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"                   \
        "v_add_u32_dpp %1, %1, %1 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"                   \
        "v_add_u32_dpp %3, %3, %3 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"                   \
        "v_max_i32 %0, %0, %4\n\t"                                                              \
        "v_max_i32 %2, %2, %5\n\t"                                                              \
        "s_nop 0\n\t"
    asm volatile(
        "s_nop 1\n\t"
        MP_STEP("row_shr:1") MP_STEP("row_shr:2") MP_STEP("row_shr:4") MP_STEP("row_shr:8")
        MP_BCAST("row_bcast:15", "0xa") MP_BCAST("row_bcast:31", "0xc")
        "s_nop 0"
        : "+v"(Aq), "+v"(Bq), "+v"(Ar), "+v"(Br), "=&v"(tq), "=&v"(tr));
#undef MP_STEP
#undef MP_BCAST
    a.A = (Aq >= MP_OFF) ? Aq - MP_OFF : S_NEG; a.B = (Bq < 0) ? -1 : Bq;
    b.A = (Ar >= MP_OFF) ? Ar - MP_OFF : S_NEG; b.B = (Br < 0) ? -1 : Br;
}

// ===========================================================================
// K1s: *striped* banded forward sweep (W = 64 cells, one wavefront per alignment).
//
// The window origin of each plane is constant over stripes of FS_K truth rows, so inside a stripe lane l
// owns the same cell x = lo_p + l of every row: its distance, base, swap source and exit mask are
// registers; the diagonal neighbour is a DPP wave_shr:1, the swap source a ds_bpermute, the INS chain the
// DPP prefix-min.  Only the first row of a stripe re-aligns (bpermute by the origin shift).  Same
// exactness argument as k_fwd_band: exit_min collects the distance of every cell with an edge leaving
// the window (the last row of a stripe is tested against the next stripe's window).
// ===========================================================================
#define FS_K 8
#define FS_W 64

// value of lane `src` (any lane index; out-of-range sources give `fill`)
__device__ __forceinline__ int lane_get(int src, int v, int fill) {
    const int r = __builtin_amdgcn_ds_bpermute((src & 63) << 2, v);
    return (src >= 0 && src < 64) ? r : fill;
}

// stripe origin: the window is centred between the reference coordinates of the stripe's first and last
// truth row, so a jump of the diagonal inside the stripe (a truth indel) costs at most half its size of margin
__device__ __forceinline__ void stripe_origin(const int32_t *t2r, const uint16_t *tj, const int32_t *r2q, int s, int n_stripes, int Lt,
                                              int Lq, int Lr, int &loQ, int &loR) {
    loQ = 0; loR = 0;
    if (s > 0 && s < n_stripes) {   // stripe 0 starts at the origin
        const int ta = s * FS_K, tb = min(ta + FS_K - 1, Lt - 1);
        const int ra = t2r[ta], rb = t2r[tb];
        const int qa = query_center(t2r, tj, r2q, ta, Lr), qb = query_center(t2r, tj, r2q, tb, Lr);
        loR = max(0, min((ra + rb) / 2 - FS_W / 2, Lr - min(FS_W, Lr)));
        loQ = max(0, min((qa + qb) / 2 - FS_W / 2, Lq - min(FS_W, Lq)));
    }
}

// Exit test of the window kernels (see the header).  An optimal path that leaves the window does so over an edge from an
// in-window cell b (distance D inside the window) to a cell c outside; it costs at least D + cost(edge) + LB(c), where LB(c)
// bounds every continuation from c = (p, x, t) to an end cell.  With rho(c) the reference coordinate of c's plane position, a
// continuation has to cover remR = rho(end) - rho(c) reference bases in remT = Lt - 1 - t truth rows.  Every edge but INS takes a
// truth row, every edge but DEL a plane position; a plane step advances rho by one -- except the steps of the query hap's
// indels: a step onto an inserted base leaves rho where it is, the step across a deletion adds its length.  So along any
// continuation      #INS - #DEL = (remR - remT) - S,      S = (deleted bases crossed) - (inserted bases crossed),
// and its cost is at least |#INS - #DEL|.  A path crosses a query-hap step at most once and only steps it can still reach:
// behind x from the QUERY plane, and from the REF plane those behind the hap position a swap from x or later lands on -- a
// deletion is crossed only by the swap in front of it, so a path that stands on one of its bases (or walks into it along the
// REF plane, as the successors of the cell in front of a deletion shared with the truth do) cannot use it (Bref, k_prep_xb).
// With I / Dl the inserted / deleted bases among those steps, S lies in [-I, Dl] and
//     LB(c) = max(0, v - Dl, -I - v),     v = remR - remT.
// (Until round 6 the bound was |rho - tau| - (I + Dl) - (all indel bases of the truth hap behind t), tau the truth row's
// reference coordinate: symmetric, so that an insertion shared by the query and the truth hap -- thousands of bases with an
// SV -- made every bound behind it zero, and a cell at distance 0 that a plane swap reaches PAST the query's insertion (its
// first bases match the reference behind it one time in four; always inside a tandem repeat) failed every window although
// nothing leads from it to the end: the truth's steps are not a budget, a continuation takes ALL of them, and their signed sum
// is in remT.  The alignments of such superclusters climbed to the dense level: joint_synth)
// The end cells' own coordinates: rho(end) = t2r[Lt - 1] for both planes where the strings end on a shared reference base
// (generate_ptrs_strs); `slack` = how far the two planes' last positions are from it otherwise, taken off every bound.
// Successors of b = (p, x, t), rho = rho(b), bud = {own, other} of xb_q / xb_r (pairs, I | Dl << 16):
//     INS   (p, x + 1, t)        cost 1   v = remR - 1 - remT       own budget   (QUERY: rho(x + 1) = rho + 1 + w with the step w
//     diag  (p, x + 1, t + 1)    cost 0   v = remR - remT           own budget    still in the budget: a weaker bound, never a wrong one)
//     DEL   (p, x, t + 1)        cost 1   v = remR - remT + 1       bud.x
//     swap  (1 - p, z, t + 1)    cost 0   v = remR - remT           other budget
// ex: which of the four leave the window (bits 1, 2, 4, 8).  Returns the smallest D + cost + LB over them, D_INF for none.
struct ExitEnd { int endT, slack; };      // rho of the end cells, and the slack above
__device__ __forceinline__ ExitEnd exit_end(const int32_t *q2r, const int32_t *t2r, int Lq, int Lr, int Lt) {
    ExitEnd e;
    e.endT = t2r[Lt - 1];
    const int a = q2r[Lq - 1] - e.endT, b = Lr - 1 - e.endT;
    e.slack = max(a < 0 ? -a : a, b < 0 ? -b : b);
    return e;
}
__device__ __forceinline__ int exit_lb(int v, int packed, int slack) {
    int ins = packed & 0xffff, del = int(uint32_t(packed) >> 16);
    ins = ins == 0xffff ? (1 << 28) : ins;
    del = del == 0xffff ? (1 << 28) : del;
    return max(max(v - del, -ins - v) - slack, 0);
}
__device__ __forceinline__ int exit_key(int ex, int p, int D, int rho, int2 bud, int remT, ExitEnd E) {
    const int b_same = p ? bud.y : bud.x, b_swap = p ? bud.x : bud.y;
    const int v = (E.endT - rho) - remT;
    const int k_ins = 1 + exit_lb(v - 1, b_same, E.slack);
    const int k_dg = exit_lb(v, b_same, E.slack);
    const int k_del = 1 + exit_lb(v + 1, bud.x, E.slack);
    const int k_sw = exit_lb(v, b_swap, E.slack);
    int k = D_INF;
    k = (ex & 1) ? min(k, k_ins) : k;
    k = (ex & 2) ? min(k, k_dg) : k;
    k = (ex & 4) ? min(k, k_del) : k;
    k = (ex & 8) ? min(k, k_sw) : k;
    return ex ? D + k : D_INF;
}

// 16 bytes per lane from LDS to global memory, issued without the compiler's knowledge: it keeps no record of a store in
// flight, so it places no s_waitcnt vmcnt(0) for it later on (a latency chain of one wave pays a full store round trip, ~0.5 us,
// for each of those; gfx9 reads a store's data registers at issue, so reusing them at once is safe -- LLVM relies on the same).
// The data becomes visible at the latest when the kernel ends.
__device__ __forceinline__ void lds_to_global16(const void *lds_src, void *dst) {
    uint4 tmp;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tglobal_store_dwordx4 %2, %0, off"
                 : "=&v"(tmp) : "v"(uint32_t(uintptr_t(lds_src))), "v"(dst) : "memory");
}

// The forward sweep of one alignment at the 64-cell level: one wavefront, stripe after stripe, row after row; results to outs[a].
// (Rounds 3 - 5 carried a block-parallel variant of this sweep -- tentative runs from guessed rows, fix-up runs until the rows
// meet, pr_fwdpar.hip -- behind VPR_PAR_FWD: exact, 0.43 instead of 6.2 ms for a 9 288-row alignment of ordinary sequence, no
// gain inside tandem repeats where the runs never meet, never the default.  Removed in round 6; `git show 303eef9:vcfdist_amd/csrc/pr_fwdpar.hip`.)
__device__ __forceinline__ void fwd_stripe_range(const DevBatch &B, const AlnDesc &d, const int a, uint8_t *__restrict__ ws,
                                                 int32_t *__restrict__ blo_all, AlnOut *__restrict__ outs,
                                                 const int64_t save_delta = 0) {
    const int lane = threadIdx.x;
    const int Lq = d.Lq, Lr = d.Lr, Lt = d.Lt;
    const int Lp[2] = {Lq, Lr};
    const uint8_t *Ts = B.hap_seq[d.ts] + d.t_off;
    const uint8_t *Tf = B.hap_flag[d.ts] + d.t_off;
    const int32_t *t2r = B.hap_ptr[d.ts] + d.t_off;
    const uint16_t *tjp = B.tj[d.ts - 2] + d.t_off;
    const int32_t *r2q = B.ref_ptr[d.qs] + d.r_off;
    const int2 *fk[2] = {B.fk_q[d.qs] + d.q_off, B.fk_r[d.qs] + d.r_off};
    const int4 *cand[2] = {B.cand_q[d.qs] + d.q_off, B.cand_r[d.qs] + d.r_off};
    const int4 *cand2[2] = {B.cand2_q[d.qs] + d.q_off, B.cand2_r[d.qs] + d.r_off};
    const int32_t *q2r = B.hap_ptr[d.qs] + d.q_off;
    const int2 *xbp[2] = {B.xb_q[d.qs] + d.q_off, B.xb_r[d.qs] + d.r_off};   // free-shift budgets of the exit test (pr_device.h)
    const ExitEnd xend = exit_end(q2r, t2r, Lq, Lr, Lt);      // end-cell coordinates of the exit test (exit_key)
    uint8_t *mat[2] = {ws + d.mat_off[0], ws + d.mat_off[1]};
    int32_t *blo = blo_all + d.blo_off;
    const int n_stripes = (Lt + FS_K - 1) / FS_K;
    // flag bytes of a stripe (FS_K rows x pitch <= 64 B per plane), flushed with one 16-byte store per lane -- a stripe LATER
    // (two buffers), so that the store is long complete when the wave next waits for a load: see lds_to_global16
    __shared__ __align__(16) uint8_t fbuf2[2][2][FS_K * FS_W];
    auto flush_stripe = [&](int s_) {
        const int ta = s_ * FS_K, nr = min(FS_K, Lt - ta);
#pragma unroll
        for (int p = 0; p < 2; p++)
            if (lane * 16 < nr * d.pitch[p]) {
                lds_to_global16(&fbuf2[s_ & 1][p][lane * 16], mat[p] + size_t(ta) * d.pitch[p] + lane * 16);
                // (k_fwd_stripe_save: a second copy the backward sweep does not overwrite, for k_restore_stripe)
                if (save_delta) lds_to_global16(&fbuf2[s_ & 1][p][lane * 16], mat[p] + save_delta + size_t(ta) * d.pitch[p] + lane * 16);
            }
    };

    // stripe origins, 64 stripes per register chunk (lane l <-> stripe c0 + l); next chunk prefetched
    int cbQ, cbR, nbQ, nbR;
    stripe_origin(t2r, tjp, r2q, lane, n_stripes, Lt, Lq, Lr, cbQ, cbR);
    stripe_origin(t2r, tjp, r2q, 64 + lane, n_stripes, Lt, Lq, Lr, nbQ, nbR);
    uint32_t tchunk = 0, tlast = 0;
    if (lane < Lt) tchunk = uint32_t(Ts[lane]) | (uint32_t(Tf[lane]) << 8);

    int exit_min = D_INF, min_tie = D_INF;
    int Dp[2] = {lane, lane};            // row 0: D = x along the INS chain (origin 0)
    int lo[2] = {0, 0}, hi[2] = {min(Lq, FS_W) - 1, min(Lr, FS_W) - 1};
    int plo[2] = {0, 0};                 // origins of the previous stripe
    int nlo[2] = {0, 0}, nhi[2] = {0, 0};
    int2 kc[2], kn[2];                   // packed constants of this / the next stripe
    int rhoc[2], rhon[2];                // reference coordinate of the lane's cell
    int2 vac[2], van[2];                 // its free-shift budgets (xb_q / xb_r)
#pragma unroll
    for (int p = 0; p < 2; p++) {
        kc[p] = make_int2(-1, int(0xffffffffu));
        rhoc[p] = lane; vac[p] = make_int2(0, 0);
        if (lo[p] + lane <= hi[p]) {
            const int x0 = lo[p] + lane;
            kc[p] = fk[p][x0];
            rhoc[p] = (p == 0) ? q2r[x0] : x0;
            vac[p] = xbp[p][x0];
        }
    }
    int s_last = 0;                      // last stripe swept

    for (int s = 0; s < n_stripes; s++) {
        const int t0 = s * FS_K;
        const int rows = min(FS_K, Lt - t0);
        uint8_t (*fbuf)[FS_K * FS_W] = fbuf2[s & 1];
        if (s > 0) flush_stripe(s - 1);
        s_last = s;
        // ---- next stripe's window, prefetch of its constants
        const bool has_next = s + 1 < n_stripes;
        if (has_next) {
            if (((s + 1) & 63) == 0) { nlo[0] = __builtin_amdgcn_readlane(nbQ, 0); nlo[1] = __builtin_amdgcn_readlane(nbR, 0); }
            else { nlo[0] = __builtin_amdgcn_readlane(cbQ, (s + 1) & 63); nlo[1] = __builtin_amdgcn_readlane(cbR, (s + 1) & 63); }
        } else { nlo[0] = lo[0]; nlo[1] = lo[1]; }
        nhi[0] = min(Lq - 1, nlo[0] + FS_W - 1);
        nhi[1] = min(Lr - 1, nlo[1] + FS_W - 1);
#pragma unroll
        for (int p = 0; p < 2; p++) {
            kn[p] = make_int2(-1, int(0xffffffffu));
            rhon[p] = 0; van[p] = make_int2(0, 0);
            if (has_next && nlo[p] + lane <= nhi[p]) {
                const int xn = nlo[p] + lane;
                kn[p] = fk[p][xn];
                rhon[p] = (p == 0) ? q2r[xn] : xn;
                van[p] = xbp[p][xn];
            }
        }
        if (lane < rows) { blo[t0 + lane] = lo[0]; blo[Lt + t0 + lane] = lo[1]; }   // read by K2 / K3

        // ---- per-lane constants of this stripe
        int s0[2];
        uint32_t base[2];
        bool multi[2];
        // which edges of the lane's cell leave the window (bits: 1 INS, 2 diagonal, 4 DEL, 8 swap): towards a row of this
        // stripe / from the stripe's last row towards the next stripe's window
        int ex_in[2], ex_last[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int o = 1 - p;
            const int x = lo[p] + lane;
            const bool valid = x <= hi[p];
            s0[p] = (kc[p].x < 0) ? -1 : (kc[p].x & (FK_MULTI - 1));
            multi[p] = kc[p].x >= 0 && (kc[p].x & FK_MULTI);
            base[p] = valid ? (uint32_t(kc[p].y) >> 24) : 0xffu;
            const int z = kc[p].y & 0xffffff;
            const bool zok = valid && z != FK_NONE24 && z < Lp[o];
            const bool ins_out = valid && x == hi[p] && hi[p] < Lp[p] - 1;
            ex_in[p] = (ins_out ? 3 : 0) | ((zok && (z < lo[o] || z > hi[o])) ? 8 : 0);
            ex_last[p] = (ins_out ? 1 : 0) |
                         ((has_next && valid && x + 1 <= Lp[p] - 1 && (x + 1 < nlo[p] || x + 1 > nhi[p])) ? 2 : 0) |
                         ((has_next && valid && (x < nlo[p] || x > nhi[p])) ? 4 : 0) |
                         ((has_next && zok && (z < nlo[o] || z > nhi[o])) ? 8 : 0);
        }
        int rowo[2] = {lane, lane};   // byte offset of this lane's cell inside the stripe's LDS block
        const bool st_ok[2] = {lane < d.pitch[0], lane < d.pitch[1]};
        // the first swap source of every cell of the stripe sits one window column to the left in the other plane (no indel inside
        // the two windows): nearly every stripe
        const bool swap_near = __all((s0[0] < 0 || s0[0] - lo[1] == lane - 1) && (s0[1] < 0 || s0[1] - lo[0] == lane - 1));

        for (int r = 0; r < rows; r++) {
            const int t = t0 + r;
            const bool last = (r == rows - 1);
            if (t == 0) {   // row 0, dist.cpp:300-305,397-405
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    if (st_ok[p]) fbuf[p][rowo[p]] = (lane == 0) ? F_MAT : F_INS;
                    rowo[p] += d.pitch[p];
                    exit_min = min(exit_min, exit_key(last ? ex_last[p] : ex_in[p], p, lane, rhoc[p], vac[p], Lt - 1, xend));
                }
                continue;
            }
            if ((t & 63) == 0) {
                tlast = __builtin_amdgcn_readlane(tchunk, 63);
                const int tt = t + lane;
                tchunk = 0;
                if (tt < Lt) {
                    tchunk = uint32_t(Ts[tt]) | (uint32_t(Tf[tt]) << 8);
                }
            }
            const uint32_t cur = __builtin_amdgcn_readlane(tchunk, t & 63);
            const uint32_t prv = ((t & 63) == 0) ? tlast : uint32_t(__builtin_amdgcn_readlane(tchunk, (t - 1) & 63));
            const uint32_t Tt = cur & 0xff;
            const bool at = fwd_allow(int((prv >> 8) & 0xff));
            const bool first = (r == 0);   // the previous row belongs to the previous stripe (origins plo)

            int v[2], up[2], dg[2], sw[2];
            uint32_t mk[2];
            bool match[2], need_multi = false;
#pragma unroll
            for (int p = 0; p < 2; p++) {
                if (first) {
                    const int sh = lo[p] - plo[p];
                    up[p] = lane_get(lane + sh, Dp[p], D_INF);
                    dg[p] = lane_get(lane + sh - 1, Dp[p], D_INF);
                } else {
                    up[p] = Dp[p];
                    dg[p] = wave_shr1(Dp[p], D_INF);
                }
            }
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int o = 1 - p;
                match[p] = base[p] == Tt;
                const bool on = match[p] && at && s0[p] >= 0;
                // (swap_near: every swap source of the stripe is the other plane's left neighbour column -- its value is that plane's
                // diagonal value, already here; the permute through LDS was on the row-to-row dependency chain)
                const int sv = (swap_near && !first) ? dg[o] : lane_get(s0[p] - (first ? plo[o] : lo[o]), Dp[o], D_INF);
                sw[p] = on ? sv : D_INF;
                need_multi = need_multi || (on && multi[p]);
            }
            uint32_t swbits[2] = {0, 0};
            if (__builtin_expect(__any(need_multi), 0)) {
                // rare: several allowed swap sources (insertion / deletion boundary); keep the highest index
                // among the optimal ones and remember ties (VPR_ST_SWAP_TIE)
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const int o = 1 - p;
                    const bool need = match[p] && at && s0[p] >= 0 && multi[p];
                    int4 cc = make_int4(-1, -1, -1, -1);
                    if (need) cc = cand[p][lo[p] + lane];
                    const int olo = first ? plo[o] : lo[o];
                    const int srcs[3] = {cc.y, cc.z, cc.w};
                    int choice = 0;
                    bool tie = false;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const int val0 = lane_get(srcs[k] - olo, Dp[o], D_INF);
                        const int val = (need && srcs[k] >= 0) ? val0 : D_INF;
                        if (need && srcs[k] >= 0 && val <= sw[p]) { tie = (val == sw[p]); sw[p] = val; choice = k + 1; }
                    }
                    if (__builtin_expect(__any(need && cc.w >= 0), 0)) {      // sources five to eight
                        int4 c2 = make_int4(-1, -1, -1, -1);
                        if (need && cc.w >= 0) c2 = cand2[p][lo[p] + lane];
                        const int more[4] = {c2.x, c2.y, c2.z, c2.w};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int val = lane_get(more[k] - olo, Dp[o], D_INF);
                            if (more[k] >= 0 && val <= sw[p]) { tie = (val == sw[p]); sw[p] = val; choice = k + 4; }
                        }
                    }
                    swbits[p] = f_choice_bits(choice) | (tie ? F_TIE : 0);
                    if (tie && sw[p] < D_INF) min_tie = min(min_tie, sw[p]);
                }
            }
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int cm = dg[p] + (match[p] ? 0 : 1);
                const int up1 = up[p] + 1;
                const int b = min(min(cm, up1), sw[p]);
                uint32_t m = (cm == b) ? (match[p] ? F_MAT : F_SUB) : 0;
                m |= (up1 == b) ? F_DEL : 0;
                m |= (sw[p] == b && sw[p] < D_INF) ? (F_SWP | swbits[p]) : 0;
                mk[p] = m;
                v[p] = b - lane;
            }
            int iq = v[0], ir = v[1];
            wave_prefix_min2(iq, ir);
            const int inc[2] = {iq, ir};
            bool may_exit = false;
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int carry = wave_shr1(inc[p], D_INF);
                const int Dn = inc[p] + lane;                       // inclusive prefix-min + x
                uint32_t f = (v[p] <= carry) ? mk[p] : 0;
                const int left = wave_shr1(Dn, D_INF);
                f |= (left + 1 == Dn) ? F_INS : 0;
                if (st_ok[p]) fbuf[p][rowo[p]] = uint8_t(f);
                rowo[p] += d.pitch[p];
                // first tier of the exit test: an exit key is D + cost + a bound >= D, so a cell whose D has reached the lane's
                // smallest key so far cannot lower it
                may_exit = may_exit || ((last ? ex_last[p] : ex_in[p]) != 0 && Dn < exit_min);
                Dp[p] = Dn;
            }
            // (second tier, the keys themselves -- 50 of the row's ~250 instructions --, only in rows where some cell can still
            // lower its lane's minimum: the cells at the window's edge keep about the same D from row to row)
            if (__any(may_exit)) {
#pragma unroll
                for (int p = 0; p < 2; p++)
                    exit_min = min(exit_min, exit_key(last ? ex_last[p] : ex_in[p], p, Dp[p], rhoc[p], vac[p], Lt - 1 - t, xend));
            }
        }
        asm volatile("" ::: "memory");      // (the stripe's flag rows leave LDS at the start of the next stripe)
        // ---- advance to the next stripe
        plo[0] = lo[0]; plo[1] = lo[1];
        lo[0] = nlo[0]; lo[1] = nlo[1]; hi[0] = nhi[0]; hi[1] = nhi[1];
        kc[0] = kn[0]; kc[1] = kn[1];
        rhoc[0] = rhon[0]; rhoc[1] = rhon[1]; vac[0] = van[0]; vac[1] = van[1];
        if (((s + 1) & 63) == 0) {
            cbQ = nbQ; cbR = nbR;
            stripe_origin(t2r, tjp, r2q, s + 1 + 64 + lane, n_stripes, Lt, Lq, Lr, nbQ, nbR);
        }
    }
    flush_stripe(s_last);
    // end cells: row Lt-1 was computed with origins plo (the last stripe's)
    const int eq = Lq - 1 - plo[0], er = Lr - 1 - plo[1];
    const int dq = (eq >= 0 && eq < 64) ? __builtin_amdgcn_readlane(Dp[0], eq & 63) : D_INF;
    const int dr = (er >= 0 && er < 64) ? __builtin_amdgcn_readlane(Dp[1], er & 63) : D_INF;
    int em = exit_min, mt = min_tie;
    wave_prefix_min2(em, mt);
    if (lane == 63) {
        outs[a].dist_q = dq;
        outs[a].dist_r = dr;
        outs[a].exit_min = em;
        // smallest distance of a cell with tied swap sources, + 1 (0: none): the host replays the container order of an
        // alignment speculatively when this is <= s, before its backward sweep has said whether a tie is consulted
        // (AlnOut::path_len is free until the walk)
        outs[a].path_len = (mt < D_INF) ? mt + 1 : 0;
    }
}

__global__ void __launch_bounds__(64) k_fwd_stripe(DevBatch B, const AlnDesc *__restrict__ descs,
                                                   const int32_t *__restrict__ work, uint8_t *__restrict__ ws,
                                                   int32_t *__restrict__ blo_all, AlnOut *__restrict__ outs) {
    __builtin_amdgcn_s_setprio(2);      // a latency chain (rows are sequential): win issue arbitration against the bulk kernels
    const int a = work[blockIdx.x];
    const AlnDesc d = descs[a];
    fwd_stripe_range(B, d, a, ws, blo_all, outs);
}
// The forward sweep of round 0's long part: the flag bytes are also written save_delta bytes further on (a region as large as
// the part's workspace).  The backward sweep replaces the flags by path_ptr bytes in place, and a tie round needs the flags again
// (it patches the decided cells and repeats the backward sweep): instead of repeating this sweep -- a chain of up to 9 288
// sequential rows, 3.3 ms on the lone step's critical path -- it copies them back (k_restore_stripe).
__global__ void __launch_bounds__(64) k_fwd_stripe_save(DevBatch B, const AlnDesc *__restrict__ descs,
                                                        const int32_t *__restrict__ work, uint8_t *__restrict__ ws,
                                                        int32_t *__restrict__ blo_all, AlnOut *__restrict__ outs, int64_t save_delta) {
    __builtin_amdgcn_s_setprio(2);
    const int a = work[blockIdx.x];
    const AlnDesc d = descs[a];
    fwd_stripe_range(B, d, a, ws, blo_all, outs, save_delta);
}
// where round 0 left an alignment's forward flags (the saved copy) and stripe origins, relative to its workspace
struct RestoreJob {
    int64_t old_mat[2];     // byte offsets of the two planes' flag rows in round 0's workspace
    int64_t old_blo;        // int32 offset of the stripe origins there
};
// one workgroup per alignment of a tie round's list: the saved flags and the origins into the round's own workspace (the
// descriptor the round staged); dist_q / dist_r / exit_min / path_len of AlnOut are still the forward sweep's
__global__ void __launch_bounds__(256) k_restore_stripe(const AlnDesc *__restrict__ descs, const int32_t *__restrict__ work,
                                                        const RestoreJob *__restrict__ jobs, const uint8_t *__restrict__ old_ws,
                                                        int64_t save_delta, uint8_t *__restrict__ ws, int32_t *__restrict__ blo_all) {
    const int a = work[blockIdx.x];
    const AlnDesc d = descs[a];
    const RestoreJob J = jobs[blockIdx.x];
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int64_t n16 = int64_t(d.Lt) * d.pitch[p] / 16;        // (pitch: a multiple of 16; both layouts start 16-byte aligned)
        const uint4 *src = reinterpret_cast<const uint4 *>(old_ws + save_delta + J.old_mat[p]);
        uint4 *dst = reinterpret_cast<uint4 *>(ws + d.mat_off[p]);
        for (int64_t k = tid; k < n16; k += 256) dst[k] = src[k];
    }
    const int32_t *bs = reinterpret_cast<const int32_t *>(old_ws) + J.old_blo;
    int32_t *bd = blo_all + d.blo_off;
    for (int k = tid; k < 2 * d.Lt; k += 256) bd[k] = bs[k];
}
// ===========================================================================
// K2s: striped banded backward max-TP sweep (W = 64), the mirror image of k_fwd_stripe: lanes are
// reversed (lane l owns x = lo + 63 - l) so "x+1" is lane l-1 (DPP wave_shr:1) and the suffix composition of
// the max-plus maps is a DPP prefix scan; scores and forward flags of row t+1 live in registers, the swap
// successor comes through ds_bpermute, only the first row of a stripe re-aligns by the origin shift.
// Uses the stripe origins k_fwd_stripe stored in blo (constant over FS_K rows).
// ===========================================================================
__global__ void __launch_bounds__(64) k_bwd_stripe(DevBatch B, const AlnDesc *__restrict__ descs,
                                                   const int32_t *__restrict__ work, uint8_t *__restrict__ ws,
                                                   const int32_t *__restrict__ blo_all, AlnOut *__restrict__ outs, int tag) {
    __builtin_amdgcn_s_setprio(2);      // a latency chain (rows are sequential): win issue arbitration against the bulk kernels
    const int a = work[blockIdx.x];
    const AlnDesc d = descs[a];
    if (outs[a].band_ok != tag || d.band_pad != tag) return;   // rejected by the exit test (re-run wider) or another round's
    const int lane = threadIdx.x;
    const int Lq = d.Lq, Lr = d.Lr, Lt = d.Lt;
    const int Lp[2] = {Lq, Lr};
    const int32_t *bk[2] = {B.bk_q[d.qs] + d.q_off, B.bk_r[d.qs] + d.r_off};
    uint8_t *mat[2] = {ws + d.mat_off[0], ws + d.mat_off[1]};
    const int32_t *blo = blo_all + d.blo_off;
    const int end_plane = outs[a].end_plane;
    const int n_stripes = (Lt + FS_K - 1) / FS_K;
    const int col = 63 - lane;                     // window column of this lane
    const bool st_ok[2] = {col < d.pitch[0], col < d.pitch[1]};
    // forward flags of the current / next-lower stripe (double buffered, 16-byte loads one stripe ahead) and
    // the path_ptr rows produced for the current stripe (flushed with 16-byte stores)
    __shared__ __align__(16) uint8_t fin[2][2][FS_K * FS_W];
    __shared__ __align__(16) uint8_t fout[2][FS_K * FS_W];
    uint4 pfv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    auto stage_load = [&](int s_) {    // request stripe s_'s forward-flag rows (16 B per lane and plane)
        if (s_ < 0) return;
        const int ta = s_ * FS_K, nr = min(ta + FS_K, Lt) - ta;
#pragma unroll
        for (int p = 0; p < 2; p++)
            if (lane * 16 < nr * d.pitch[p])
                pfv[p] = *reinterpret_cast<const uint4 *>(mat[p] + size_t(ta) * d.pitch[p] + lane * 16);
    };
    auto stage_commit = [&](int buf) {  // ... and park them in LDS once they are needed (a stripe later)
#pragma unroll
        for (int p = 0; p < 2; p++)
            if (lane * 16 < FS_K * FS_W) *reinterpret_cast<uint4 *>(&fin[buf][p][lane * 16]) = pfv[p];
    };

    // stripe origins, 64 stripes per register chunk (lane l <-> stripe c0 + l); the chunk below is prefetched
    auto load_chunk = [&](int c0, int &bq, int &br) {
        bq = 0; br = 0;
        const int s = c0 + lane;
        if (c0 >= 0 && s < n_stripes) { bq = blo[s * FS_K]; br = blo[Lt + s * FS_K]; }
    };
    int cbQ, cbR, lbQ, lbR;
    const int c_top = (n_stripes - 1) & ~63;
    load_chunk(c_top, cbQ, cbR);
    load_chunk(c_top - 64, lbQ, lbR);

    int lo[2], hi[2], plo[2] = {0, 0};
    lo[0] = __builtin_amdgcn_readlane(cbQ, (n_stripes - 1) & 63);
    lo[1] = __builtin_amdgcn_readlane(cbR, (n_stripes - 1) & 63);
    int bkc[2], bkn[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        hi[p] = min(Lp[p] - 1, lo[p] + FS_W - 1);
        bkc[p] = int(FK_NONE24);
        if (lo[p] + col <= hi[p]) bkc[p] = bk[p][lo[p] + col];
    }
    int sc1[2] = {S_NEG, S_NEG};     // scores of row t+1 (own column)
    int f1[2] = {0, 0};              // forward flags of row t+1
    stage_load(n_stripes - 1);
    stage_commit((n_stripes - 1) & 1);
    uint32_t tie_used = 0;
    asm volatile("" ::"v"(bkc[0]), "v"(bkc[1]), "v"(lbQ), "v"(lbR) : "memory");      // (no load in flight at the loop's entry: see the stripe's end)

    for (int s = n_stripes - 1; s >= 0; s--) {
        const int t0 = s * FS_K, t1 = min(t0 + FS_K, Lt) - 1;
        // ---- the stripe below: origins and constants (prefetched)
        int nlo[2] = {0, 0};
        if (s > 0) {
            if ((s & 63) == 0) { nlo[0] = __builtin_amdgcn_readlane(lbQ, 63); nlo[1] = __builtin_amdgcn_readlane(lbR, 63); }
            else { nlo[0] = __builtin_amdgcn_readlane(cbQ, (s - 1) & 63); nlo[1] = __builtin_amdgcn_readlane(cbR, (s - 1) & 63); }
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            bkn[p] = int(FK_NONE24);
            const int xn = nlo[p] + col;
            if (s > 0 && xn <= min(Lp[p] - 1, nlo[p] + FS_W - 1)) bkn[p] = bk[p][xn];
        }
        // ---- per-lane constants of this stripe
        bool valid[2];
        int tp_own[2], tp_right[2], zl[2];
        uint32_t zkey[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            valid[p] = lo[p] + col <= hi[p];
            tp_own[p] = (bkc[p] >> 24) & 1;
            tp_right[p] = wave_shr1(tp_own[p], 0);
            zl[p] = bkc[p] & 0xffffff;   // swap target (absolute index in the other plane) or FK_NONE24
            zkey[p] = f_swp_key(rank_of(uint32_t(bkc[p]) >> 24));
        }
        const int sh[2] = {plo[0] - lo[0], plo[1] - lo[1]};   // origin shift against the stripe above (first row)
        // every swap target of the stripe sits in the lane next door of the other plane's row (no indel inside the two windows)
        const bool swap_near = __all((zl[0] == int(FK_NONE24) || 63 - (zl[0] - lo[1]) == lane - 1) &&
                                     (zl[1] == int(FK_NONE24) || 63 - (zl[1] - lo[0]) == lane - 1));
        stage_load(s - 1);                                     // prefetch the stripe below into registers
        asm volatile("" ::: "memory");
        const uint8_t *fcur[2] = {fin[s & 1][0], fin[s & 1][1]};
        int rowo[2] = {(t1 - t0) * d.pitch[0] + col, (t1 - t0) * d.pitch[1] + col};

        for (int t = t1; t >= t0; t--) {
            const bool first = (t == t1) && (s != n_stripes - 1);   // row t+1 is aligned to the stripe above
            int best[2], lk[2], f0[2];
            uint32_t bm[2];
            MP g[2];
            int up_sv[2], up_fv[2], dn_sv[2], dn_fv[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                if (first) {
                    up_sv[p] = lane_get(lane + sh[p] - 1, sc1[p], S_NEG);
                    up_fv[p] = lane_get(lane + sh[p] - 1, f1[p], 0);
                    dn_sv[p] = lane_get(lane + sh[p], sc1[p], S_NEG);
                    dn_fv[p] = lane_get(lane + sh[p], f1[p], 0);
                } else {
                    up_sv[p] = wave_shr1(sc1[p], S_NEG);
                    up_fv[p] = wave_shr1(f1[p], 0);
                    dn_sv[p] = sc1[p];
                    dn_fv[p] = f1[p];
                }
            }
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int o = 1 - p;
                const int up_s = up_sv[p], up_f = up_fv[p], dn_s = dn_sv[p], dn_f = dn_fv[p];
                int b = S_NEG;
                uint32_t m = 0;
                if (f_diag(up_f)) { b = up_s + tp_right[p]; m = f_diag(up_f); }
                if (dn_f & F_DEL) {
                    if (dn_s > b) { b = dn_s; m = F_DEL; } else if (dn_s == b) m |= F_DEL;
                }
                // swap successor z = (other plane, zl, t+1): its lane in the alignment of row t+1.  (swap_near: in every cell of
                // the stripe it is the other plane's diagonal successor -- the value one lane over, already here: no permute through
                // LDS on the row-to-row dependency chain)
                const int olo = first ? plo[o] : lo[o];
                const int zlane = 63 - (zl[p] - olo);
                const bool znone = zl[p] == int(FK_NONE24);
                int zf, zs;
                if (swap_near && !first) {
                    zf = znone ? 0 : up_fv[o];
                    zs = znone ? S_NEG : up_sv[o];
                } else {
                    zf = lane_get(znone ? -1 : zlane, f1[o], 0);
                    zs = lane_get(znone ? -1 : zlane, sc1[o], S_NEG);
                }
                if ((uint32_t(zf) & F_SWP_KEY_MASK) == zkey[p]) {
                    const int v = zs + ((bkc[p] >> 27) & 1);
                    if (v >= 0 && (zf & F_TIE)) tie_used++;
                    if (v > b) { b = v; m = F_SWP; } else if (v == b) m |= F_SWP;
                }
                if (t == Lt - 1 && p == end_plane && lo[p] + col == Lp[p] - 1) { b = 0; m = F_MAT; }   // dist.cpp:538-546
                if (!valid[p]) { b = S_NEG; m = 0; }
                best[p] = b;
                bm[p] = m;
                f0[p] = st_ok[p] ? int(fcur[p][rowo[p]]) : 0;   // forward flags of (x, t), staged a stripe ago
                const int f0r = wave_shr1(f0[p], 0);   // forward flags of (x+1, t)
                lk[p] = (f0r & F_INS) ? tp_right[p] : -1;
                g[p].A = b; g[p].B = lk[p];
            }
            // The in-row INS chain (a score flows from (x + 1, t) to (x, t) where the former carries F_INS) is a max-plus scan
            // over the row: 60 of the row's ~210 instructions.  It moves something only when a cell that IS on an optimal path
            // (best >= 0) was entered by an INS edge -- a row of the walk's insertion steps; everywhere else the row's scores
            // are `best` as they stand.
            int inc[2] = {S_NEG, S_NEG};
            if (__any((best[0] >= 0 && (f0[0] & F_INS)) || (best[1] >= 0 && (f0[1] & F_INS)))) {
                MP hq = g[0], hr = g[1];
                wave_prefix_mp2(hq, hr);
                inc[0] = wave_shr1(hq.A, S_NEG); inc[1] = wave_shr1(hr.A, S_NEG);
            }
            uint32_t outm[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                int v = best[p];
                uint32_t m = bm[p];
                if (lk[p] >= 0) {
                    const int w = inc[p] + lk[p];
                    if (w > v) { v = w; m = F_INS; } else if (w == v) m |= F_INS;
                }
                if (v < 0) { v = S_NEG; m = 0; }
                outm[p] = m;
                sc1[p] = v;          // becomes the "row t+1" score of the next iteration
                f1[p] = f0[p];       // ... and its forward flags
                if (st_ok[p]) fout[p][rowo[p]] = m ? uint8_t(m | (uint32_t(f0[p]) & F_KEEP)) : uint8_t(0);
                rowo[p] -= d.pitch[p];
            }
            (void)outm;
        }
        stage_commit((s - 1) & 1);   // the prefetched rows of the stripe below (requested a stripe ago)
        // (... and its constants: every load of the stripe has landed before the stores below go out.  The compiler keeps no
        // record across the loop's back edge of which loads are done: with a store or a load still in its books there it put a
        // `s_waitcnt vmcnt(0)` in front of the first use of bkc -- right behind the NEXT stripe's loads: a full round trip per
        // stripe of eight rows, and another one per store in front of the next store's data)
        asm volatile("" ::"v"(bkn[0]), "v"(bkn[1]) : "memory");
        // ---- flush this stripe's path_ptr rows (in place of the forward flags), 16 bytes per lane (lds_to_global16: issued
        // without the compiler's knowledge; the rows are next read by the walk, another kernel)
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int nbytes = (t1 - t0 + 1) * d.pitch[p];
            if (lane * 16 < nbytes) lds_to_global16(&fout[p][lane * 16], mat[p] + size_t(t0) * d.pitch[p] + lane * 16);
        }
        asm volatile("" ::: "memory");
        // ---- advance to the stripe below
        plo[0] = lo[0]; plo[1] = lo[1];
        lo[0] = nlo[0]; lo[1] = nlo[1];
        hi[0] = min(Lq - 1, lo[0] + FS_W - 1); hi[1] = min(Lr - 1, lo[1] + FS_W - 1);
        bkc[0] = bkn[0]; bkc[1] = bkn[1];
        if ((s & 63) == 0 && s > 0) {
            cbQ = lbQ; cbR = lbR;
            load_chunk(((s - 1) & ~63) - 64, lbQ, lbR);
        }
    }
    // (QUERY, 0, 0) is column 0 of row 0 = lane 63 (stripe 0 starts at the origin)
    const int bs = __builtin_amdgcn_readlane(sc1[0], 63);
    if (lane == 0) outs[a].beg_plane = (bs >= 0) ? VPR_PLANE_QUERY : VPR_PLANE_REF;   // dist.cpp:811-814
    if (tie_used) { atomicOr(&outs[a].status, VPR_ST_SWAP_TIE); outs[a].band_ok = TIE_MARK(tag); atomicAdd(&outs[a].n_sec, int(tie_used)); }
}

// ===========================================================================
// K3r: the forward walk of get_prec_recall_path_sync (dist.cpp:865-998) as a sweep over truth rows, one
// wavefront per alignment, for the striped 64-cell layout of k_fwd_stripe / k_bwd_stripe.
//
// Inside one row the walk can only take INS moves (every other move consumes a truth base), and it takes
// one exactly when INS is the highest-priority move left in the cell's path_ptr byte (priority
// REF-plane swap > MAT > SUB > INS > DEL > QUERY-plane swap, dist.cpp:907-935).  So the cells visited in row t
// are the entry cell e plus the run of "INS-only" cells that follows it: a ballot over the window row and a
// count-trailing-ones give the run, the lanes of the run store their path entries side by side (coalesced),
// and one readlane of the run's last cell decides the move into row t+1.  Rows are read once, coalesced, from
// the stripe's 16-byte LDS staging - no per-step pointer chase through HBM.
// ===========================================================================
// The walk over stripes [s_begin, s_end) of one alignment from a known entry state (plane hi, column e, the move mv_in that
// entered the first row, path index n): the whole alignment (k_walk_rows) or one segment of it (k_wseg_emit, pr_walkseg.hip).
// whole: this call covers the alignment: it also stores path_len.
__device__ __forceinline__ void walk_rows_range(const DevBatch &B, const AlnDesc &d, AlnOut &O, const uint8_t *__restrict__ ws,
                                                const int32_t *__restrict__ blo_all, PathEnt *__restrict__ paths,
                                                int s_begin, int s_end, int hi, int e, int mv_in, int64_t n, bool whole) {
    const int lane = threadIdx.x;
    const int Lq = d.Lq, Lr = d.Lr, Lt = d.Lt;
    const int Lp[2] = {Lq, Lr};
    // packed walk constants (k_prep_wk): {pointer to the other coordinate system, flags | "an insertion at this reference
    // position" per hap slot << 8}: one load per position instead of a pointer load and dependent has_ins loads
    const int2 *wq_ = B.wk_q[d.qs] + d.q_off, *wr_ = B.wk_r[d.qs] + d.r_off, *wt_ = B.wk_t[d.ts - 2] + d.t_off;
    const int insmask = ((1 << d.qs) | (1 << d.ts)) << 8;
    const uint8_t *mat[2] = {ws + d.mat_off[0], ws + d.mat_off[1]};
    const int32_t *blo = blo_all + d.blo_off;
    PathEnt *path = paths + d.path_off;
    const int n_stripes = (Lt + FS_K - 1) / FS_K;
    __shared__ __align__(16) uint8_t pin[2][2][FS_K * FS_W];   // [buffer][plane] path_ptr rows of a stripe

    // ---- staging of the path_ptr rows, one stripe ahead (16 B per lane and plane)
    uint4 pfv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    auto stage_load = [&](int s_) {
        if (s_ >= n_stripes) return;
        const int ta = s_ * FS_K, nr = min(ta + FS_K, Lt) - ta;
#pragma unroll
        for (int p = 0; p < 2; p++)
            if (lane * 16 < nr * d.pitch[p])
                pfv[p] = *reinterpret_cast<const uint4 *>(mat[p] + size_t(ta) * d.pitch[p] + lane * 16);
    };
    auto stage_commit = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; p++)
            if (lane * 16 < FS_K * FS_W) *reinterpret_cast<uint4 *>(&pin[buf][p][lane * 16]) = pfv[p];
    };
    // ---- per-row constants of 64 truth rows at a time (lane l <-> row (t & ~63) + l), the next 64 prefetched
    const int t_first = s_begin * FS_K;
    int2 rcur = make_int2(0, 0), rnxt = make_int2(0, 0);
    {
        const int rb = t_first & ~63;
        if (rb + lane < Lt) rcur = wt_[rb + lane];
        if (rb + 64 + lane < Lt) rnxt = wt_[rb + 64 + lane];
    }
    // ---- stripe origins, 64 stripes per register chunk (lane l <-> stripe c0 + l), the next chunk prefetched
    int cbQ = 0, cbR = 0, nbQ = 0, nbR = 0;
    auto load_org = [&](int c0, int &bq, int &br) {
        bq = 0; br = 0;
        const int s_ = c0 + lane;
        if (s_ < n_stripes) { bq = blo[s_ * FS_K]; br = blo[Lt + s_ * FS_K]; }
    };
    load_org(s_begin & ~63, cbQ, cbR);
    load_org((s_begin & ~63) + 64, nbQ, nbR);
    // ---- per-column constants of the current stripe (lane l <-> column lo_p + l) and, requested a stripe ahead, of the next
    int2 ccq = make_int2(0, 0), ccr = make_int2(0, 0), ncq = make_int2(0, 0), ncr = make_int2(0, 0);
    auto load_cols = [&](int loQ, int loR, int2 &cq, int2 &cr) {
        cq = make_int2(0, 0); cr = make_int2(0, 0);
        const int xq = loQ + lane, xr = loR + lane;
        if (xq < Lq) cq = wq_[xq];
        if (xr < Lr) cr = wr_[xr];
    };

    stage_load(s_begin);
    stage_commit(s_begin & 1);
    int lo[2] = {__builtin_amdgcn_readlane(cbQ, s_begin & 63), __builtin_amdgcn_readlane(cbR, s_begin & 63)};
    load_cols(lo[0], lo[1], ccq, ccr);

    uint32_t status = 0;
    uint32_t edit_in = (mv_in & (F_SUB | F_DEL)) ? 1u : 0u;
    bool ok = true;

    for (int s = s_begin; s < s_end && ok; s++) {
        const int t0 = s * FS_K, rows = min(FS_K, Lt - t0);
        stage_load(s + 1);                 // next stripe's rows into registers
        int nlo[2] = {0, 0};
        if (s + 1 < n_stripes) {
            if (((s + 1) & 63) == 0) { nlo[0] = __builtin_amdgcn_readlane(nbQ, 0); nlo[1] = __builtin_amdgcn_readlane(nbR, 0); }
            else { nlo[0] = __builtin_amdgcn_readlane(cbQ, (s + 1) & 63); nlo[1] = __builtin_amdgcn_readlane(cbR, (s + 1) & 63); }
            load_cols(nlo[0], nlo[1], ncq, ncr);      // (consumed when the stripe is done)
        }
        asm volatile("" ::: "memory");
        const uint8_t *pcurQ = pin[s & 1][0], *pcurR = pin[s & 1][1];
        for (int r = 0; r < rows; r++) {
            const int t = t0 + r;
            if ((t & 63) == 0 && t != t_first) {
                rcur = rnxt;
                rnxt = make_int2(0, 0);
                if (t + 64 + lane < Lt) rnxt = wt_[t + 64 + lane];
            }
            const int el = e - lo[hi];
            if (el < 0 || el > 63 || e >= Lp[hi]) { status |= VPR_ST_ERR_NO_PTR; ok = false; break; }
            const int trv = __builtin_amdgcn_readlane(rcur.x, t & 63);
            // sync flag of the entry cell, dist.cpp:949-968 (only a diagonal move can make a sync point)
            uint32_t sync_in = 1;
            if (mv_in != 0) {
                const int rwy = __builtin_amdgcn_readlane(rcur.y, t & 63);
                const int tflv = rwy & 0xff;
                int qflv = 0, qr, cwy;
                if (hi == 0) {
                    cwy = __builtin_amdgcn_readlane(ccq.y, el);
                    qflv = cwy & 0xff;
                    qr = __builtin_amdgcn_readlane(ccq.x, el);
                } else {
                    qr = e;
                    cwy = __builtin_amdgcn_readlane(ccr.y, el);
                }
                const bool in_t = (tflv & PV) && !(tflv & PB);
                const bool in_q = (qflv & PV) && !(qflv & PB);
                sync_in = (!in_t && !in_q && !((rwy | cwy) & insmask) && trv == qr && (mv_in & (F_MAT | F_SWP | F_SUB))) ? 1u : 0u;
            }
            // the run of INS-only cells that starts at the entry cell
            const int pq = (lane < d.pitch[0]) ? int(pcurQ[r * d.pitch[0] + lane]) : 0;
            const int pr = (lane < d.pitch[1]) ? int(pcurR[r * d.pitch[1] + lane]) : 0;
            const int pc = (hi == 0) ? pq : pr;
            const bool ins_only = (pc & F_INS) && !(pc & (F_MAT | F_SUB)) && !(hi == 1 && (pc & F_SWP));
            const unsigned long long m = __ballot(ins_only) >> el;
            const int k = (~m == 0ull) ? 64 : __builtin_ctzll(~m);
            const int c = e + k;                                  // last cell visited in this row
            if (c - lo[hi] > 63 || c >= Lp[hi]) { status |= VPR_ST_ERR_NO_PTR; ok = false; break; }
            if (n + k + 1 > d.path_cap) { status |= VPR_ST_ERR_LIMIT; ok = false; break; }
            if (lane >= el && lane <= el + k) {
                const int x = lo[hi] + lane;
                PathEnt pe;
                pe.a = uint32_t(x) | (uint32_t(hi) << 31);
                pe.b = uint32_t(t) | ((lane == el) ? ((sync_in << 31) | (edit_in << 30)) : (1u << 30));
                pe.qref = (hi == 0) ? ccq.x : x;
                pe.tref = trv;
                path[n + (lane - el)] = pe;
            }
            n += k + 1;
            if (t == Lt - 1) {                                    // the walk ends at the end cell of its plane
                if (c != Lp[hi] - 1) { status |= VPR_ST_ERR_NO_PTR; ok = false; }
                break;
            }
            // move out of the row from cell c, by priority
            const int cl = c - lo[hi];
            const int p = __builtin_amdgcn_readlane(pc, cl) & 31;
            if (hi == 1 && (p & F_SWP)) { mv_in = F_SWP; e = __builtin_amdgcn_readlane(ccr.x, cl) + 1; hi = 0; edit_in = 0; }
            else if (p & F_MAT) { mv_in = F_MAT; e = c + 1; edit_in = 0; }
            else if (p & F_SUB) { mv_in = F_SUB; e = c + 1; edit_in = 1; }
            else if (p & F_DEL) { mv_in = F_DEL; e = c; edit_in = 1; }
            else if (hi == 0 && (p & F_SWP)) { mv_in = F_SWP; e = __builtin_amdgcn_readlane(ccq.x, cl) + 1; hi = 1; edit_in = 0; }
            else { status |= VPR_ST_ERR_NO_PTR; ok = false; break; }
        }
        // advance to the next stripe: commit its staged rows, take over its per-column constants
        stage_commit((s + 1) & 1);
        asm volatile("" ::: "memory");
        if (s + 1 < n_stripes) { lo[0] = nlo[0]; lo[1] = nlo[1]; ccq = ncq; ccr = ncr; }
        if (((s + 1) & 63) == 0) {
            cbQ = nbQ; cbR = nbR;
            load_org(s + 1 + 64, nbQ, nbR);
        }
    }
    if (lane == 0) {
        if (whole) O.path_len = int32_t(n);
        if (!ok) { O.n_sec = 0; }
        if (status) atomicOr(&O.status, status);
    }
}

__global__ void __launch_bounds__(64) k_walk_rows(DevBatch B, const AlnDesc *__restrict__ descs,
                                                  const int32_t *__restrict__ work, int n_work,
                                                  const uint8_t *__restrict__ ws, const int32_t *__restrict__ blo_all,
                                                  AlnOut *__restrict__ outs, PathEnt *__restrict__ paths, int tag) {
    if (int(blockIdx.x) >= n_work) return;
    __builtin_amdgcn_s_setprio(2);      // a latency chain (rows are sequential): win issue arbitration against the bulk kernels
    const int a = work[blockIdx.x];
    const AlnDesc d = descs[a];
    AlnOut &O = outs[a];
    if (O.band_ok != tag || d.band_pad != tag) return;
    walk_rows_range(B, d, O, ws, blo_all, paths, 0, (d.Lt + FS_K - 1) / FS_K, O.beg_plane, 0, 0, 0, true);
}
