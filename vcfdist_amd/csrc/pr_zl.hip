// pr_zl.hip -- the zero-distance level of the short part: ONE LANE PER ALIGNMENT, forward expansion + backward max-TP
// pass + walk in one kernel (k_zero_lane).
//
// On whole-genome input nine alignments in ten have distance 0, and for those the reference's own algorithm is tiny:
// wave 0 of calc_prec_recall_aln (dist.cpp:312-381) is a breadth-first expansion over MAT and SWP edges only, every
// edge advances the truth row by one, and a row holds the one or two cells (seldom more) whose diagonal still matches.
// A window kernel spends a 16-lane row group on the 32 cells around them; here a lane owns an alignment and keeps the
// reached cells of the current row in registers -- up to four diagonals per plane ("slots": a MAT edge keeps its slot,
// a SWP edge lands in a slot of the other plane).  The 64 lanes of a wave walk 64 alignments of (nearly) equal length in
// lockstep, row by row:
//
//   forward   row t -> t+1: MAT child of every live slot (next base of its plane == next truth base), SWP child
//             (fwd_allow at the source and the truth row, dist.cpp:336-339, base at ptr + 1 of the other plane == next
//             truth base).  The row's cells go to the log, one byte each: "entered by MAT / by SWP from slot s" and the
//             two bits of the cell's own position word the backward pass needs (is_tp, bwd_allow).
//   backward  calc_prec_recall_path (dist.cpp:536-687, zero-cost moves only) over the log rows in reverse: score of a
//             cell = max over its successors of (their score + is_tp), path_ptr bits = the moves that reach the max;
//             they and the slot of the SWP successor are a second, 4-byte record per row.
//   walk      get_prec_recall_path_sync (dist.cpp:865-998): from the begin cell along the path_ptr bits by the
//             reference's priorities (REF plane: SWP first; then MAT; QUERY plane: SWP last), sync flag per step; the
//             steps are 8-byte records for k_zero_credit (16-byte path entries on request, VPR_CFG_KEEP_PATHS).
//
// What it does not take: an alignment whose end cells are not reached at distance 0 (s > 0), a row with more than four
// diagonals on a plane, lengths that do not fit the 16-bit position fields, and a cell that receives a SWP edge from two
// different sources (the reference keeps the last writer, dist.cpp:347,376 -- at distance 0 that is plain FIFO order,
// but such cells are rare and the general kernels + the container-order replay already decide them).  All of these are
// *rejected* exactly like a failed window test: band_ok = 0, the alignment re-runs in place with the 16-cell kernels.
//
// Memory access is the point of the layout: a lane streams through its own alignment, so with the batch's CSR arrays
// every load of a wave would touch 64 different cache lines.  k_prep_zl therefore writes, once per batch, a
// WAVE-INTERLEAVED copy of what the kernel reads: for the 64 alignments of a wave, position x of alignment l is word
// (x * 64 + l) of the wave's Q / R / T block -- the lanes are at (nearly) the same position at the same time, so a wave
// load is a few consecutive 256-byte rows.  The logs have the same shape (row t of lane l at (t * 64 + l) * record size),
// and their records are as small as they can be -- 8 + 4 + 8 bytes per row -- because the kernel is bound by the bytes it
// moves: every byte is written once and read once, by a later pass, when it has long left the caches.
#ifndef PR_ZL_HIP_
#define PR_ZL_HIP_
#include <type_traits>

// (struct ZlWave -- per wave of 64 alignments, the offsets of its interleaved input block and of its log blocks: pr_device.h)

// position word
#define ZW_PTR(w) (int((w) & 0xffffu) - 1)
#define ZW_BASE(w) (((w) >> 16) & 0x7fu)
#define ZW_PV (1u << 23)
#define ZW_PB (1u << 24)
#define ZW_PE (1u << 25)
#define ZW_INS (1u << 26)
#define ZW_TP (1u << 27)
__device__ __forceinline__ bool zw_fwd_allow(uint32_t w) { return !(w & ZW_PV) || (w & ZW_PE); }
__device__ __forceinline__ bool zw_bwd_allow(uint32_t w) { return !(w & ZW_PV) || (w & ZW_PB); }

// log entry of a cell: one byte (a row's eight slots are one 8-byte record, byte p * 4 + s)
#define ZE_ALIVE 1u
#define ZE_HASMAT 2u
#define ZE_HASSWP 4u
#define ZE_PSLOT_SHIFT 3       // 2 bits: slot of the SWP predecessor (other plane, previous row)
#define ZE_TP 32u
#define ZE_BWD 64u

#define ZL_MAXLEN 65000
#define ZL_CLEAN_REJECT (-1)     // AlnOut::exit_min of a reject whose wave 0 is complete: the distance-1 lane kernel's input (pr_d1.hip)

// K0z: wave-interleaved position words.  One workgroup per wave of 64 alignments.  The arrays are read the way they lie
// (64 consecutive positions of ONE alignment per wave load), turned in LDS, and written the way the lane kernel reads them
// (position x of the 64 alignments = one 256-byte row): both sides of the transpose are whole cache lines.
//   hdr[w], list: the waves of one chunk's short part (list[64 w + l] = alignment of lane l, -1 beyond the end)
__global__ void __launch_bounds__(256) k_prep_zl(DevBatch B, const AlnDesc *__restrict__ descs, const int32_t *__restrict__ list,
                                                 int n_list, const ZlWave *__restrict__ hdr, uint32_t *__restrict__ zin) {
    __shared__ uint32_t tile[64][65];       // [position in the tile][alignment]
    __shared__ int64_t s_qo[64], s_ro[64], s_to[64];
    __shared__ int32_t s_lq[64], s_lr[64], s_lt[64];
    __shared__ uint8_t s_qs[64], s_ts[64];
    const int w = blockIdx.x;
    const ZlWave H = hdr[w];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    if (threadIdx.x < 64) {
        const int wi = w * 64 + lane;
        const int a = wi < n_list ? list[wi] : -1;
        if (a >= 0) {
            const AlnDesc *d = descs + a;
            s_qo[lane] = d->q_off; s_ro[lane] = d->r_off; s_to[lane] = d->t_off;
            s_lq[lane] = d->Lq; s_lr[lane] = d->Lr; s_lt[lane] = d->Lt;
            s_qs[lane] = uint8_t(d->qs); s_ts[lane] = uint8_t(d->ts);
        } else {
            s_qo[lane] = s_ro[lane] = s_to[lane] = 0;
            s_lq[lane] = s_lr[lane] = s_lt[lane] = 0;
            s_qs[lane] = 0; s_ts[lane] = 2;
        }
    }
    __syncthreads();
    auto flagbits = [](int f) -> uint32_t { return ((f & PV) ? ZW_PV : 0u) | ((f & PB) ? ZW_PB : 0u) | ((f & PE) ? ZW_PE : 0u); };
    uint32_t *out = zin + H.in_off;
    for (int arr = 0; arr < 3; arr++) {                 // Q, R, T blocks
        const int mlen = arr == 0 ? H.mq : (arr == 1 ? H.mr : H.mt);
        for (int x0 = 0; x0 < mlen; x0 += 64) {
            const int x = x0 + lane;
#pragma unroll 16
            for (int k = 0; k < 16; k++) {              // this wave's 16 alignments, 64 positions each
                const int l = sub * 16 + k;
                const int qs = s_qs[l], ts = s_ts[l];
                const int Lr = s_lr[l];
                const int len = arr == 0 ? s_lq[l] : (arr == 1 ? Lr : s_lt[l]);
                uint32_t v = 0x7f0000u;
                if (x < len) {
                    // (pointer, flags and "an insertion of hap slot s sits at this position's reference base" come in one 8-byte
                    // record of the walk constants, k_prep_wk: no second, dependent round trip for the insertion marks)
                    const int64_t ro = s_ro[l];
                    const int slot = arr == 2 ? ts : qs;
                    const int64_t o = (arr == 0 ? s_qo[l] : (arr == 1 ? ro : s_to[l])) + x;
                    const uint8_t *seq = arr == 1 ? B.ref_seq : B.hap_seq[slot];
                    const int2 *wk = arr == 0 ? B.wk_q[qs] : (arr == 1 ? B.wk_r[qs] : B.wk_t[ts - 2]);
                    const int2 k2 = wk[o];
                    const int p = k2.x, f = arr == 1 ? int(B.ref_flag[qs][o]) : (k2.y & 0xff);
                    const uint32_t ins = ((k2.y >> 8) & ((1 << qs) | (1 << ts))) ? ZW_INS : 0u;
                    v = uint32_t((p + 1) & 0xffff) | (uint32_t(seq[o] & 0x7f) << 16) | flagbits(f) | ins;
                    if (arr == 0 && x > 0 && ((p != wk[o - 1].x + 1) || (f & PB))) v |= ZW_TP;       // dist.cpp:572-574
                    (void)Lr;
                }
                tile[lane][l] = v;
            }
            __syncthreads();
            const int rows = min(64, mlen - x0);
            for (int r = sub; r < rows; r += 4) out[int64_t(x0 + r) * 64 + lane] = tile[r][lane];
            __syncthreads();
        }
        out += int64_t(mlen) * 64;
    }
}

// ===========================================================================
// KZ: forward + backward + walk of the zero-distance alignments, one lane per alignment.
// All memory traffic goes through two buffer descriptors built from wave-uniform values (the wave's input block and its
// log block): a lane's address is a 32-bit byte offset, and an offset behind the block -- what a lane without work for
// the access gets -- is answered with zeros by the bounds check instead of costing a branch or a cache line.
// ===========================================================================
typedef __attribute__((ext_vector_type(4))) unsigned int zl_u4;
typedef __attribute__((ext_vector_type(2))) unsigned int zl_u2;
#define ZL_OOB 0xfffffff0u

__global__ void __launch_bounds__(64, 6) k_zero_lane(const AlnDesc *__restrict__ descs, const int32_t *__restrict__ list, int n_list,
                                                  const ZlWave *__restrict__ hdr, const uint32_t *__restrict__ zin,
                                                  uint4 *__restrict__ zlog, AlnOut *__restrict__ outs, PathEnt *__restrict__ paths,
                                                  int keep_paths, int d1_max_rows, int prio_rows) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const ZlWave H = hdr[w];
    // The launch lasts as long as its longest wave (the work list is sorted longest first: ~1 000 rows x three dependent passes),
    // and a wave that shares its SIMD with five others advances at half a lone wave's pace: the long ones go first.
    if (H.mt >= prio_rows) __builtin_amdgcn_s_setprio(2);
    const int wi = w * 64 + lane;
    const int a_ = wi < n_list ? list[wi] : -1;
    const bool live = a_ >= 0;
    const int a = max(a_, 0);
    const AlnDesc *dp = descs + a;
    const int Lq = live ? dp->Lq : 1, Lr = live ? dp->Lr : 1, Lt = live ? dp->Lt : 0;
    const int L[2] = {Lq, Lr};
    // input block: position x of plane p at ((pos0[p] + x) * 64 + lane) * 4, truth row t at ((post + t) * 64 + lane) * 4
    const auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(zin + H.in_off), 0, 256 * (H.mq + H.mr + H.mt), 0x00020000);
    // log block: the cells' flag bytes (8 B per row and lane), the path_ptr words (4 B), the walk's steps (8 B)
    const auto rlog = __builtin_amdgcn_make_buffer_rsrc(zlog + H.log_off, 0, 1280 * H.mt, 0x00020000);
    const uint32_t lane4 = uint32_t(lane) << 2, lane8 = uint32_t(lane) << 3;
    const uint32_t pos0[2] = {0u, uint32_t(H.mq) << 8}, post = uint32_t(H.mq + H.mr) << 8;
    const uint32_t logP0 = uint32_t(H.mt) << 9;
    auto in_at = [&](uint32_t off) -> uint32_t { return __builtin_amdgcn_raw_buffer_load_b32(rin, off, 0, 0); };
    // (a one-row alignment only exists where a region was cut at the contig end, include/vcfdist_pr.h: the reference's
    // backward pass never terminates on it -- left to the general kernels)
    bool ok = live && Lq < ZL_MAXLEN && Lr < ZL_MAXLEN && Lt < ZL_MAXLEN && Lt >= 2;
    const int rows = ok ? Lt : 0;
    int tmax = rows;
#pragma unroll
    for (int o = 32; o; o >>= 1) tmax = max(tmax, __shfl_xor(tmax, o));
    if (tmax == 0) {
        if (live) { outs[a].dist_q = D_INF; outs[a].dist_r = D_INF; outs[a].exit_min = 0; }
        return;
    }

    // ---------------- forward: the cells of wave 0, row by row (dist.cpp:317-381)
    int qri[2][4];
    uint32_t pw[2][4];
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int s = 0; s < 4; s++) { qri[p][s] = -1; pw[p][s] = 0; }
    }
    uint32_t tw0 = in_at(ok ? post + lane4 : ZL_OOB);
    if (ok) {     // the two start cells, dist.cpp:300-305
        qri[0][0] = 0; pw[0][0] = in_at(pos0[0] + lane4);
        qri[1][0] = 0; pw[1][0] = in_at(pos0[1] + lane4);
        zl_u2 e0;
        e0.x = ZE_ALIVE | ZE_HASMAT | ((pw[0][0] & ZW_TP) ? ZE_TP : 0u) | (zw_bwd_allow(pw[0][0]) ? ZE_BWD : 0u);
        e0.y = ZE_ALIVE | ZE_HASMAT | (zw_bwd_allow(pw[1][0]) ? ZE_BWD : 0u);
        __builtin_amdgcn_raw_buffer_store_b64(e0, rlog, lane8, 0, 0);
    }
    // One row step over the first S slots of each plane (S = 2: no lane of the wave has a third diagonal -- nearly every
    // row; S = 4 otherwise).  Returns false when S = 2 was not enough (a SWP child found both slots of its plane taken):
    // nothing has been committed then and the caller repeats the row with S = 4.
    // Software pipeline of the two-slot path: the children's words of the cells of row t + 1 -- what the NEXT row step compares --
    // are requested as soon as those cells are known, in the middle of this row step, and the rest of the step (the log entries,
    // their store, the loop) runs while they are on their way.  pf_*: the words requested by the previous two-slot step for
    // slots 0 and 1 (have_pf: there are such; a four-slot step leaves none).  The SWP child is requested wherever the cell
    // allows a swap at all; whether the truth row does is only known in the next step, which then ignores the word.
    uint32_t pf_mw[2][2] = {{0, 0}, {0, 0}}, pf_sw[2][2] = {{0, 0}, {0, 0}};
    bool have_pf = false;
    auto fwd_row = [&](auto Sc, int t, bool act, uint32_t tw1) -> bool {
        constexpr int S = decltype(Sc)::value;
        const bool t_allow = zw_fwd_allow(tw0);
        // children's position words: MAT child x + 1 of the own plane, SWP child ptr + 1 of the other plane
        uint32_t mw[2][S], sw[2][S];
        int sz[2][S];
        const bool use_pf = S == 2 && have_pf;
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
            for (int s = 0; s < S; s++) {
                const bool alive = act && qri[p][s] >= 0;
                const int c = qri[p][s] + 1;
                const int z = ZW_PTR(pw[p][s]) + 1;
                const bool sok = alive && t_allow && zw_fwd_allow(pw[p][s]) && z < L[1 - p];
                sz[p][s] = sok ? z : -1;
                if (use_pf) {
                    mw[p][s] = (alive && c < L[p]) ? pf_mw[p][s & 1] : 0u;
                    sw[p][s] = sok ? pf_sw[p][s & 1] : 0u;
                } else {
                    mw[p][s] = in_at((alive && c < L[p]) ? pos0[p] + (uint32_t(c) << 8) + lane4 : ZL_OOB);
                    sw[p][s] = in_at(sok ? pos0[1 - p] + (uint32_t(z) << 8) + lane4 : ZL_OOB);
                }
            }
        }
        const uint32_t Tb = ZW_BASE(tw1);
        int nq[2][S];
        uint32_t nw[2][S], ne[2][S];
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
            for (int s = 0; s < S; s++) {
                const bool hit = act && qri[p][s] >= 0 && ZW_BASE(mw[p][s]) == Tb;      // (a word from behind the block has base 0)
                nq[p][s] = hit ? qri[p][s] + 1 : -1;
                nw[p][s] = mw[p][s];
                ne[p][s] = hit ? ZE_HASMAT : 0u;
            }
        }
        bool bad = false, full = false;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int o = 1 - p;
#pragma unroll
            for (int s = 0; s < S; s++) {
                const bool hit = sz[p][s] >= 0 && ZW_BASE(sw[p][s]) == Tb;
                if (!__any(hit)) continue;
                const int z = sz[p][s];
                // the cell (o, z) of row t + 1: already there (by MAT, or by another source's SWP), else the first free slot
                int at = -1;
#pragma unroll
                for (int k = S - 1; k >= 0; k--) at = (nq[o][k] < 0) ? k : at;
#pragma unroll
                for (int k = 0; k < S; k++) at = (nq[o][k] == z) ? k : at;
                full = full || (hit && at < 0);                   // S = 4: a fifth diagonal on the plane
#pragma unroll
                for (int k = 0; k < S; k++) {
                    const bool here = hit && at == k;
                    bad = bad || (here && (ne[o][k] & ZE_HASSWP));   // a second SWP source of the cell: left to the general kernels
                    nq[o][k] = here ? z : nq[o][k];
                    nw[o][k] = here ? sw[p][s] : nw[o][k];
                    ne[o][k] = here ? (ne[o][k] | ZE_HASSWP | (uint32_t(s) << ZE_PSLOT_SHIFT)) : ne[o][k];
                }
            }
        }
        if (S < 4 && __any(full)) return false;
        ok = ok && !bad && !full;
        if (S == 2) {       // the next step's words (see pf_*)
#pragma unroll
            for (int p = 0; p < 2; p++) {
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    const bool alive = act && ok && nq[p][s] >= 0;
                    const int c = nq[p][s] + 1;
                    const int z = ZW_PTR(nw[p][s]) + 1;
                    pf_mw[p][s] = in_at((alive && c < L[p]) ? pos0[p] + (uint32_t(c) << 8) + lane4 : ZL_OOB);
                    pf_sw[p][s] = in_at((alive && zw_fwd_allow(nw[p][s]) && z < L[1 - p]) ? pos0[1 - p] + (uint32_t(z) << 8) + lane4 : ZL_OOB);
                }
            }
        }
        have_pf = S == 2;
        // the row's log entries
        {
            zl_u2 ev;
            uint32_t wq = 0, wr_ = 0;
#pragma unroll
            for (int sl = 0; sl < S; sl++) {
                wq |= (nq[0][sl] < 0 ? 0u : (ZE_ALIVE | ne[0][sl] | ((nw[0][sl] & ZW_TP) ? ZE_TP : 0u) | (zw_bwd_allow(nw[0][sl]) ? ZE_BWD : 0u))) << (8 * sl);
                wr_ |= (nq[1][sl] < 0 ? 0u : (ZE_ALIVE | ne[1][sl] | (zw_bwd_allow(nw[1][sl]) ? ZE_BWD : 0u))) << (8 * sl);
            }
            ev.x = wq; ev.y = wr_;
            __builtin_amdgcn_raw_buffer_store_b64(ev, rlog, (act && ok) ? (uint32_t(t + 1) << 9) + lane8 : ZL_OOB, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
            for (int s = 0; s < S; s++) { qri[p][s] = act ? nq[p][s] : qri[p][s]; pw[p][s] = act ? nw[p][s] : pw[p][s]; }
        }
        tw0 = act ? tw1 : tw0;
        return true;
    };
    // (tried: the truth words four rows ahead as well -- three more registers spill at six waves per SIMD, and at five the launch
    // is 10 % slower than before: 3.15 against 2.87 ms)
    for (int t = 0; t + 1 < tmax; t++) {
        const bool act = ok && t + 1 < rows;
        const uint32_t tw1 = in_at(act ? post + (uint32_t(t + 1) << 8) + lane4 : ZL_OOB);
        bool upper = false;
#pragma unroll
        for (int p = 0; p < 2; p++) upper = upper || (act && (qri[p][2] >= 0 || qri[p][3] >= 0));
        if (__any(upper) || !fwd_row(std::integral_constant<int, 2>{}, t, act, tw1))
            (void)fwd_row(std::integral_constant<int, 4>{}, t, act, tw1);
    }
    // end cells (dist.cpp:390-391, 436-439: the QUERY plane is preferred)
    int endq = -1, endr = -1;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (qri[0][s] == Lq - 1) endq = s;
        if (qri[1][s] == Lr - 1) endr = s;
    }
    const bool clean = ok && Lt <= d1_max_rows;       // wave 0 is complete and without a tie: all that can be wrong now is s > 0 (pr_d1.hip)
    ok = ok && (endq >= 0 || endr >= 0);
    if (live) {
        AlnOut &o = outs[a];
        o.dist_q = (ok && endq >= 0) ? 0 : D_INF;
        o.dist_r = (ok && endr >= 0) ? 0 : D_INF;
        o.exit_min = ok ? D_INF : (clean ? ZL_CLEAN_REJECT : 0);          // k_fwd_band_finish: accepted iff s = 0 here
    }
    if (!__any(ok)) return;

    // ---------------- backward: max-TP scores over the zero-cost moves (dist.cpp:550-681), rows Lt-1 .. 1.  The path_ptr
    // bits of a row's cells are one word: four bits per slot (MAT, SWP, slot of the SWP successor)
    const int nrow = ok ? Lt : 0;
    int sc[2][4];
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int s = 0; s < 4; s++) sc[p][s] = -1;
    }
    int bmax = nrow;
#pragma unroll
    for (int o = 32; o; o >>= 1) bmax = max(bmax, __shfl_xor(bmax, o));
    auto log_row = [&](int t, bool on) -> zl_u2 {
        return __builtin_amdgcn_raw_buffer_load_b64(rlog, on ? (uint32_t(t) << 9) + lane8 : ZL_OOB, 0, 0);
    };
    zl_u2 cur = zl_u2{0, 0};      // entries of row t
    auto bwd_row = [&](auto Sc, int t, bool act, zl_u2 pre) {
        constexpr int S = decltype(Sc)::value;
        int best[2][S];
        uint32_t pp[2][S];      // path_ptr nibble of the row t - 1 cells: 1 MAT, 2 SWP, bits 2-3 slot of the SWP successor
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
            for (int s = 0; s < S; s++) { best[p][s] = -1; pp[p][s] = 0; }
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int o = 1 - p;
#pragma unroll
            for (int s = 0; s < S; s++) {
                const bool on = act && sc[p][s] >= 0;
                const uint32_t e = ((p ? cur.y : cur.x) >> (8 * s)) & 0xffu;
                const int tp = (e & ZE_TP) ? 1 : 0;
                {      // MAT predecessor: same plane, same slot
                    const bool m = on && (e & ZE_HASMAT);
                    const int v = sc[p][s] + tp;
                    const bool gt = m && v > best[p][s], eq = m && v == best[p][s];
                    best[p][s] = gt ? v : best[p][s];
                    pp[p][s] = gt ? 1u : (eq ? (pp[p][s] | 1u) : pp[p][s]);
                }
                {      // SWP predecessor (bwd_allow at this cell, dist.cpp:599-602); leaving a REF cell scores 0 (dist.cpp:614)
                    const bool m = on && (e & ZE_HASSWP) && (e & ZE_BWD);
                    const int v = sc[p][s] + (p == 1 ? 0 : tp);
                    const int ps = int((e >> ZE_PSLOT_SHIFT) & 3u);
                    const uint32_t mine = 2u | (uint32_t(s) << 2);
#pragma unroll
                    for (int k = 0; k < S; k++) {
                        const bool h_ = m && ps == k;
                        const bool gt = h_ && v > best[o][k], eq = h_ && v == best[o][k];
                        best[o][k] = gt ? v : best[o][k];
                        pp[o][k] = gt ? mine : (eq ? ((pp[o][k] & 1u) | mine) : pp[o][k]);
                    }
                }
            }
        }
        uint32_t ppw = 0;
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                if (s < S) {
                    ppw |= (best[p][s] >= 0 ? pp[p][s] : 0u) << (4 * (p * 4 + s));
                    sc[p][s] = act ? best[p][s] : sc[p][s];
                } else {
                    sc[p][s] = act ? -1 : sc[p][s];       // (no cell in the upper slots of either row)
                }
            }
        }
        __builtin_amdgcn_raw_buffer_store_b32(ppw, rlog, act ? logP0 + (uint32_t(t - 1) << 8) + lane4 : ZL_OOB, 0, 0);
        if (act) cur = pre;
    };
    // The rows' entries are requested four rows ahead of their use (their addresses do not depend on anything: a lane's row r is
    // at r * 512 + lane * 8), a lane's LAST row with them (it used to be a load of its own, waited for, in up to 64 different
    // iterations of a wave): a row step no longer waits for memory.
    auto row_at = [&](int r) -> zl_u2 { return log_row(r, r >= 0 && r < nrow); };
    // The four rows in flight live in four FIXED registers, the loop is unrolled four times and each copy of the row step names its
    // own: as a queue (q0 = q1; q1 = q2; q2 = load) the compiler rotated the registers with moves, a move of a register whose load
    // is in flight waits for it -- `s_waitcnt vmcnt(0)` at the head of every row: the newest request, and the path_ptr store behind
    // it: the look-ahead bought nothing (round 6, found in the ISA).
    zl_u2 lastpre = row_at(bmax - 1);
    zl_u2 rA = row_at(bmax - 2), rB = row_at(bmax - 3), rC = row_at(bmax - 4), rD = zl_u2{0, 0};
    auto bwd_step = [&](int t, const zl_u2 &pre_reg, zl_u2 &next_reg) {     // pre_reg: row t - 1; next_reg: free, receives row t - 4
        const bool act = t >= 1 && t < nrow;
        const bool first = t == nrow - 1 && t >= 1;     // this lane's last row: the end cell has score 0 (dist.cpp:538-546)
        if (first) {
            const int ep = endq >= 0 ? 0 : 1, es = endq >= 0 ? endq : endr;
            cur = lastpre;
#pragma unroll
            for (int p = 0; p < 2; p++) {
#pragma unroll
                for (int s = 0; s < 4; s++) sc[p][s] = (p == ep && s == es) ? 0 : -1;
            }
        }
        next_reg = row_at(t - 4);
        const zl_u2 pre = pre_reg;            // row t - 1
        // upper slots in use in either row (by any lane)?
        const bool upper = act && (((cur.x | cur.y | pre.x | pre.y) & 0xffff0000u) != 0u);
        if (__any(upper)) bwd_row(std::integral_constant<int, 4>{}, t, act, pre);
        else bwd_row(std::integral_constant<int, 2>{}, t, act, pre);
        lastpre = pre;
    };
    for (int t = bmax - 1; t >= 1; t -= 4) {      // (the copies past row 1 of the last round do nothing: act is false, their loads are out of range)
        bwd_step(t, rA, rD);
        bwd_step(t - 1, rB, rA);
        bwd_step(t - 2, rC, rB);
        bwd_step(t - 3, rD, rC);
    }
    // (QUERY, 0, 0) on a path to the end?  dist.cpp:811-814
    const int beg_plane = sc[0][0] >= 0 ? VPR_PLANE_QUERY : VPR_PLANE_REF;

    // (the walk -- a third pass of dependent loads, the address of a row's position word comes out of the row before -- is a
    // kernel of its own, k_zero_walk, on the side stream of the credit walks: this launch is on the short part's chain, that one
    // runs beside the distance-1 level and the in-place 16-cell round)
    if (ok) outs[a].beg_plane = beg_plane;
}

// ===========================================================================
// KZw: the walk of the alignments k_zero_lane finished (get_prec_recall_path_sync, dist.cpp:905-982, sync flags :949-968), one
// lane per alignment over the wave's log: path_ptr words in, step records out (what k_zero_credit reads).
// ===========================================================================
__global__ void __launch_bounds__(64, 8) k_zero_walk(const AlnDesc *__restrict__ descs, const int32_t *__restrict__ list, int n_list,
                                                  const ZlWave *__restrict__ hdr, const uint32_t *__restrict__ zin,
                                                  uint4 *__restrict__ zlog, AlnOut *__restrict__ outs, PathEnt *__restrict__ paths,
                                                  int keep_paths, int tag, int prio_rows) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const ZlWave H = hdr[w];
    if (H.mt >= prio_rows) __builtin_amdgcn_s_setprio(2);
    const int wi = w * 64 + lane;
    const int a_ = wi < n_list ? list[wi] : -1;
    const bool live = a_ >= 0;
    const int a = max(a_, 0);
    const AlnDesc *dp = descs + a;
    const auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(zin + H.in_off), 0, 256 * (H.mq + H.mr + H.mt), 0x00020000);
    const auto rlog = __builtin_amdgcn_make_buffer_rsrc(zlog + H.log_off, 0, 1280 * H.mt, 0x00020000);
    const uint32_t lane4 = uint32_t(lane) << 2, lane8 = uint32_t(lane) << 3;
    const uint32_t pos0[2] = {0u, uint32_t(H.mq) << 8}, post = uint32_t(H.mq + H.mr) << 8;
    const uint32_t logP0 = uint32_t(H.mt) << 9, logS0 = logP0 + (uint32_t(H.mt) << 8);
    auto in_at = [&](uint32_t off) -> uint32_t { return __builtin_amdgcn_raw_buffer_load_b32(rin, off, 0, 0); };
    // finished by k_zero_lane: accepted by its exit test (k_fwd_band_finish) at this level
    const bool ok = live && dp->band_pad == tag && outs[a].band_ok == tag;
    const int nrow = ok ? dp->Lt : 0;
    int bmax = nrow;
#pragma unroll
    for (int o = 32; o; o >>= 1) bmax = max(bmax, __shfl_xor(bmax, o));
    if (bmax == 0) return;
    const int beg_plane = ok ? outs[a].beg_plane : 0;
    // ---------------- walk (dist.cpp:905-982) + sync flags (dist.cpp:949-968)
    PathEnt *path = paths + dp->path_off;
    int hi = beg_plane, slot = 0, mv_in = 0, x = 0;
    uint32_t status = 0;
    bool wok = ok;
    for (int t = 0; t < bmax; t++) {
        const bool act = wok && t < nrow;
        const uint32_t ppw = __builtin_amdgcn_raw_buffer_load_b32(rlog, (act && t + 1 < nrow) ? logP0 + (uint32_t(t) << 8) + lane4 : ZL_OOB, 0, 0);
        const uint32_t tw = in_at(act ? post + (uint32_t(t) << 8) + lane4 : ZL_OOB);
        const uint32_t cw = in_at(act ? pos0[hi] + (uint32_t(x) << 8) + lane4 : ZL_OOB);
        const int trv = ZW_PTR(tw), qref = hi ? x : ZW_PTR(cw);
        uint32_t sync = 1;
        if (mv_in) {
            const bool in_t = (tw & ZW_PV) && !(tw & ZW_PB);
            const bool in_q = hi == 0 && (cw & ZW_PV) && !(cw & ZW_PB);
            sync = (!in_t && !in_q && !((tw | cw) & ZW_INS) && trv == qref) ? 1u : 0u;
        }
        // "the reference base at this step's reference coordinate equals its truth base": what the credit walk compares for the
        // one-base sections between neighbouring sync points (nearly every step), from the words at hand instead of two scattered
        // byte loads per lane and step in k_zero_credit.  On the REF plane the cell's base IS that reference base; on the QUERY
        // plane it is wherever the hap position lies outside a variant (generate_ptrs_strs copies the reference there); inside
        // one the bit is marked invalid and the credit walk loads the bases.
        const uint32_t eqb = (ZW_BASE(cw) == ZW_BASE(tw) ? 1u : 0u) | ((hi || !(cw & ZW_PV)) ? 2u : 0u);
        zl_u2 st;           // the step as k_zero_credit reads it
        st.x = uint32_t(x) | (uint32_t(hi) << 16) | (sync << 17) | (eqb << 18);
        st.y = uint32_t(qref + 1) | (uint32_t(trv + 1) << 16);
        __builtin_amdgcn_raw_buffer_store_b64(st, rlog, act ? logS0 + (uint32_t(t) << 9) + lane8 : ZL_OOB, 0, 0);
        if ((keep_paths & 1) && act) {       // VPR_CFG_KEEP_PATHS: also as a 16-byte path entry (vpr_download_path)
            uint4 pe;
            pe.x = uint32_t(x) | (uint32_t(hi) << 31);
            pe.y = uint32_t(t) | (sync << 31);
            pe.z = uint32_t(qref);
            pe.w = uint32_t(trv);
            *reinterpret_cast<uint4 *>(path + t) = pe;
        }
        if (act && t + 1 < nrow) {      // the move out of the cell, by priority; a SWP edge lands behind the cell's pointer
            const uint32_t nb = (ppw >> (4 * ((hi << 2) | slot))) & 15u;
            if (hi == 1 && (nb & 2u)) { hi = 0; slot = int(nb >> 2); x = ZW_PTR(cw) + 1; }
            else if (nb & 1u) { x = x + 1; }
            else if (hi == 0 && (nb & 2u)) { hi = 1; slot = int(nb >> 2); x = ZW_PTR(cw) + 1; }
            else { status |= VPR_ST_ERR_NO_PTR; wok = false; }
            mv_in = 1;
        }
    }
    if (ok) {
        AlnOut &o = outs[a];
        o.path_len = wok ? nrow : 0;
        if (!wok) o.n_sec = 0;
        if (status) atomicOr(&o.status, status);
    }
}

// ===========================================================================
// KZc: the credit sections of the alignments k_zero_lane finished (integer part of calc_prec_recall, dist.cpp:1035-1400:
// credit_walk), one lane per alignment.  The path comes from the walk's step records: row t of lane l is 8 bytes at
// (t * 64 + l) * 8 of the wave's third log region, so a wave's load of a step is one 512-byte stretch instead of 64 lines
// of 64 different path blocks.
// ===========================================================================
struct ZlFetch {
    const uint2 *log;       // this lane's column of the wave's step records
    int64_t pre_i;
    uint2 pre;
    __device__ PathEnt operator()(int64_t i) {
        const uint2 v = (i == pre_i) ? pre : log[i * 64];
        if (i > 0) { pre_i = i - 1; pre = log[(i - 1) * 64]; }    // entries are read in descending order
        PathEnt e;
        e.a = (v.x & 0xffffu) | (((v.x >> 16) & 1u) << 31);
        e.b = uint32_t(i) | (((v.x >> 17) & 1u) << 31) | (((v.x >> 18) & 3u) << 28);      // bits 28, 29: see credit_walk
        e.qref = int(v.y & 0xffffu) - 1;
        e.tref = int(v.y >> 16) - 1;
        return e;
    }
};

__global__ void __launch_bounds__(64) k_zero_credit(DevBatch B, const AlnDesc *__restrict__ descs, const int32_t *__restrict__ list,
                                                    int n_list, const ZlWave *__restrict__ hdr, const uint4 *__restrict__ zlog,
                                                    AlnOut *__restrict__ outs, Section *__restrict__ secs,
                                                    int32_t *const *__restrict__ fp_group, EdJob *__restrict__ jobs,
                                                    int32_t *__restrict__ n_jobs, int32_t jobs_cap, int tag) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const int wi = w * 64 + lane;
    if (wi >= n_list) return;
    const int a = list[wi];
    if (a < 0) return;
    const AlnDesc d = descs[a];
    AlnOut &O = outs[a];
    if (d.band_pad != tag || O.band_ok != tag) return;       // not finished by k_zero_lane: the in-place 16-cell round has it
    if (O.status & (VPR_ST_ERR_NO_PTR | VPR_ST_ERR_LIMIT)) return;
    const ZlWave H = hdr[w];
    ZlFetch f{reinterpret_cast<const uint2 *>(zlog + H.log_off + 48 * int64_t(H.mt)) + lane, -1, make_uint2(0, 0)};     // behind 512 + 256 bytes per row
    credit_walk<false, ZlFetch, true>(B, d, O, a, nullptr, int64_t(O.path_len), 0u, secs, fp_group, jobs, n_jobs, jobs_cap, true, f);
}

#endif
