// pr_d1.hip -- distance ONE at the lane level: what the zero-distance lane kernel (pr_zl.hip) rejects only because its end cells
// are not reached at distance 0 -- 86 % of its rejects have distance exactly 1 -- is finished here, one lane per alignment
// again, instead of by the 16-cell window kernels (k_fwd_q16 / k_bwd_q16 / k_walk_q16 / k_credit<lane>).
//
// The reference's algorithm at s = 1 (calc_prec_recall_aln, dist.cpp:286-442):
//   wave 0   F0 = everything reachable from the two start cells over MAT / SWP edges (what k_zero_lane computes); no end cell in it;
//   wave 1   every F0 cell sends its INS / DEL / SUB edge (dist.cpp:395-426) into cells that are not in F0 ("landings"), and the
//            landings are extended over MAT / SWP edges again (dist.cpp:317-381); a cell first reached now has D = 1 and its flag
//            byte holds exactly the edges that arrive from F0 cells (edits) and from D = 1 cells (MAT, SWP + swap_pred);
//   stop     an end cell has D = 1; the QUERY plane's is preferred (dist.cpp:436-439).
// The backward pass (calc_prec_recall_path, dist.cpp:517-823) starts at that end cell E and only ever sees the D = 1 cells from
// which E is reachable over zero-cost edges -- call the set of ALL cells with a zero-cost path to E "B0" (it cannot meet F0, or
// s would be 0) -- then the edit edges back into F0, then F0's own flags down to the start.  So, per lane:
//
//   pass 1   B0 of the QUERY end cell, rows Lt-1 .. 0 (a breadth-first walk backwards over MAT and SWP edges; the sources of a
//            SWP edge into a position come with the position: the first two entries of K0's candidate lists, cand_q / cand_r);
//            up to four cells per plane and row, their positions go to the log.
//   pass 2   rows 0 .. Lt-1: F0 as in k_zero_lane, and beside it G1 = the B0 cells with D = 1: a B0 cell of row t is one iff
//            an F0 cell sits at its INS / DEL / SUB source, or a G1 cell of row t-1 has a MAT / SWP edge into it.  That is the
//            reference's flag byte of the cell; the F0 slot of each edit source and the G1 slot of the SWP source ride along.
//            Two different G1 sources of one SWP edge (the reference keeps the last writer, in wave 1 that is the iteration
//            order of its unordered_set): rejected, as a second source is at s = 0.  s = 1 iff the end cell is in G1.
//   pass 3   max-TP scores backwards over both layers (dist.cpp:550-806; moves out of a cell: MAT, SWP, and from F0 cells SUB /
//            INS / DEL into G1), and per cell THE move the walk will take: among the moves that reach the maximum the first by
//            the walk's priority (dist.cpp:907-935), with the slot it lands in.  One byte per cell.
//   pass 4   the walk + sync flags (dist.cpp:865-998) along those bytes; the steps are the 8-byte records of k_zero_lane plus an
//            edit bit and "truth row = step - 1" behind an INS step; k_one_credit reads them like k_zero_credit.
//
// Not taken (the lane stays rejected and re-runs in place with the 16-cell kernels, as before): s >= 2; s = 1 only at the REF
// plane's end cell; a fifth B0 / F0 cell of a plane in one row; a position with three or more swap sources; the tie above.
//
// Layout: the rejects are scattered over the zero level's waves, so their position words are interleaved again for waves of 64
// REJECTS (k_d1_hdr / k_d1_scan size the blocks on the device from the device-built fail list, k_prep_d1 writes them: the words
// of k_prep_zl plus one word per QUERY / REF position with its two swap sources).  Log per row and lane: 16 B (B0 positions,
// later the move bytes) + 8 B (F0 flag bytes, later the steps) + 16 B (G1 flag halfwords).
#ifndef PR_D1_HIP_
#define PR_D1_HIP_

#define D1_TAG 9            // AlnOut::band_ok of an alignment finished here
#define D1_CLEAN ZL_CLEAN_REJECT       // AlnOut::exit_min of a zero-level reject whose F0 is complete (s > 0 is all that is wrong with it)

// G1 flag halfword of a B0 slot
#define GE_ALIVE 1u
#define GE_MAT 2u
#define GE_SWP 4u
#define GE_PSLOT_SHIFT 3        // 2 bits: G1 slot (other plane, previous row) of the SWP source
#define GE_SUB 32u
#define GE_SUB_SHIFT 6          // 2 bits: F0 slot (same plane, previous row) of the SUB source
#define GE_INS 256u
#define GE_INS_SHIFT 9          // 2 bits: F0 slot (same plane, same row) of the INS source
#define GE_DEL 2048u
#define GE_DEL_SHIFT 12         // 2 bits: F0 slot (same plane, previous row) of the DEL source
#define GE_TP 16384u
#define GE_BWD 32768u

// move byte of a cell (pass 3 -> pass 4): rank by the walk's priority | slot the move lands in << 3; 0xff: not on a path
#define MV_SWP_R 0      // REF plane: swap first (dist.cpp:907)
#define MV_MAT 1
#define MV_SUB 2
#define MV_INS 3
#define MV_DEL 4
#define MV_SWP_Q 5      // QUERY plane: swap last (dist.cpp:931)

// ---- headers of the waves of 64 rejects, built on the device: largest lengths ...
__global__ void __launch_bounds__(64) k_d1_hdr(const AlnDesc *__restrict__ descs, const int32_t *__restrict__ list,
                                               const int32_t *__restrict__ n_dev, int n_cap, const AlnOut *__restrict__ outs,
                                               ZlWave *__restrict__ hdr) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const int n = min(*n_dev, n_cap);
    const int wi = w * 64 + lane;
    const int a = wi < n ? list[wi] : -1;
    int lq = 0, lr = 0, lt = 0;
    if (a >= 0 && outs[a].exit_min == D1_CLEAN) {
        const AlnDesc *d = descs + a;
        lq = d->Lq; lr = d->Lr; lt = d->Lt;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        lq = max(lq, __shfl_xor(lq, o)); lr = max(lr, __shfl_xor(lr, o)); lt = max(lt, __shfl_xor(lt, o));
    }
    if (lane == 0) {
        ZlWave W;
        W.in_off = 0; W.log_off = 0; W.mq = lq; W.mr = lr; W.mt = lt; W.pad = 0;
        hdr[w] = W;
    }
}

// ... and their offsets (one workgroup; a wave that does not fit the blocks any more is dropped: mt = 0, its lanes stay rejected)
//   info[0] = waves with work, info[1] = waves dropped
#define D1_SCAN_NT 256      // (four waves find a compute unit beside a device full of long-lived one-wave workgroups: k_fails_scan, pr_api.hip)
__global__ void __launch_bounds__(D1_SCAN_NT) k_d1_scan(ZlWave *__restrict__ hdr, int n_waves, int64_t in_cap, int64_t log_cap,
                                                  int32_t *__restrict__ info) {
    __shared__ int64_t s_in[D1_SCAN_NT], s_log[D1_SCAN_NT];
    __shared__ int s_used, s_drop;
    const int tid = threadIdx.x;
    if (tid == 0) { s_used = 0; s_drop = 0; }
    const int per = (n_waves + D1_SCAN_NT - 1) / D1_SCAN_NT;
    const int b = min(n_waves, tid * per), e = min(n_waves, b + per);
    int64_t si = 0, sl = 0;
    for (int k = b; k < e; k++) {
        const ZlWave W = hdr[k];
        if (W.mt > 0) { si += 64 * (2 * int64_t(W.mq) + 2 * int64_t(W.mr) + W.mt); sl += 160 * (int64_t(W.mt) + 1); }
    }
    s_in[tid] = si; s_log[tid] = sl;
    __syncthreads();
    for (int o = 1; o < D1_SCAN_NT; o <<= 1) {
        const int64_t ai = tid >= o ? s_in[tid - o] : 0, al = tid >= o ? s_log[tid - o] : 0;
        __syncthreads();
        s_in[tid] += ai; s_log[tid] += al;
        __syncthreads();
    }
    int64_t oi = s_in[tid] - si, ol = s_log[tid] - sl;
    int used = 0, drop = 0;
    for (int k = b; k < e; k++) {
        ZlWave W = hdr[k];
        if (W.mt <= 0) continue;
        const int64_t ni = 64 * (2 * int64_t(W.mq) + 2 * int64_t(W.mr) + W.mt), nl = 160 * (int64_t(W.mt) + 1);
        if (oi + ni > in_cap || ol + nl > log_cap) { W.mq = W.mr = W.mt = 0; drop++; }
        else { W.in_off = oi; W.log_off = ol; used++; }
        oi += ni; ol += nl;
        hdr[k] = W;
    }
    if (used) atomicAdd(&s_used, used);
    if (drop) atomicAdd(&s_drop, drop);
    __syncthreads();
    if (tid == 0) { info[0] = s_used; info[1] = s_drop; }
}

// ---- position words of a wave of rejects: QUERY words, QUERY swap sources, REF words, REF swap sources, truth words
// (k_prep_zl's transpose; the source word of a position = its first two allowed swap sources + 1, 0xffff in the upper half
// where there are more than two)
__global__ void __launch_bounds__(256) k_prep_d1(DevBatch B, const AlnDesc *__restrict__ descs, const int32_t *__restrict__ list,
                                                 const int32_t *__restrict__ n_dev, int n_cap, const AlnOut *__restrict__ outs,
                                                 const ZlWave *__restrict__ hdr, uint32_t *__restrict__ zin) {
    __shared__ uint32_t tile[64][65];
    __shared__ int64_t s_qo[64], s_ro[64], s_to[64];
    __shared__ int32_t s_lq[64], s_lr[64], s_lt[64];
    __shared__ uint8_t s_qs[64], s_ts[64];
    const int w = blockIdx.x;
    const ZlWave H = hdr[w];
    if (H.mt <= 0) return;
    const int n_list = min(*n_dev, n_cap);
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    if (threadIdx.x < 64) {
        const int wi = w * 64 + lane;
        int a = wi < n_list ? list[wi] : -1;
        if (a >= 0 && outs[a].exit_min != D1_CLEAN) a = -1;
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
    for (int arr = 0; arr < 5; arr++) {                 // QW, QC, RW, RC, TW
        const int pl = arr >> 1;                        // 0 QUERY, 1 REF, 2 truth
        const bool src_words = (arr & 1) && arr < 4;
        const int mlen = pl == 0 ? H.mq : (pl == 1 ? H.mr : H.mt);
        for (int x0 = 0; x0 < mlen; x0 += 64) {
            const int x = x0 + lane;
#pragma unroll 16
            for (int k = 0; k < 16; k++) {
                const int l = sub * 16 + k;
                const int qs = s_qs[l], ts = s_ts[l];
                const int Lr = s_lr[l];
                const int len = pl == 0 ? s_lq[l] : (pl == 1 ? Lr : s_lt[l]);
                uint32_t v = src_words ? 0u : 0x7f0000u;
                if (x < len) {
                    const int64_t ro = s_ro[l];
                    if (src_words) {
                        const int4 c = pl == 0 ? B.cand_q[qs][s_qo[l] + x] : B.cand_r[qs][ro + x];
                        v = uint32_t(c.x >= 0 ? c.x + 1 : 0) | (c.z >= 0 ? 0xffff0000u : (uint32_t(c.y >= 0 ? c.y + 1 : 0) << 16));
                    } else {
                        const int slot = pl == 2 ? ts : qs;
                        const int64_t o = (pl == 0 ? s_qo[l] : (pl == 1 ? ro : s_to[l])) + x;
                        const uint8_t *seq = pl == 1 ? B.ref_seq : B.hap_seq[slot];
                        const int2 *wk = pl == 0 ? B.wk_q[qs] : (pl == 1 ? B.wk_r[qs] : B.wk_t[ts - 2]);      // (see k_prep_zl)
                        const int2 k2 = wk[o];
                        const int p = k2.x, f = pl == 1 ? int(B.ref_flag[qs][o]) : (k2.y & 0xff);
                        const uint32_t ins = ((k2.y >> 8) & ((1 << qs) | (1 << ts))) ? ZW_INS : 0u;
                        v = uint32_t((p + 1) & 0xffff) | (uint32_t(seq[o] & 0x7f) << 16) | flagbits(f) | ins;
                        if (pl == 0 && x > 0 && ((p != wk[o - 1].x + 1) || (f & PB))) v |= ZW_TP;       // dist.cpp:572-574
                        (void)Lr;
                    }
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
// K1L: the four passes, one lane per alignment.  Every pass's row step exists for S = 2 and S = 4 slots per plane and layer
// (nearly every row of every lane holds one or two cells per plane: the step over two slots is a third of the instructions,
// and a lone wavefront's row time is its instruction count); a row in which any lane of the wave needs a third slot is
// stepped with four.
// ===========================================================================
__global__ void __launch_bounds__(64, 3) k_one_lane(const AlnDesc *__restrict__ descs, const int32_t *__restrict__ list,
                                                    const int32_t *__restrict__ n_dev, int n_cap, const ZlWave *__restrict__ hdr,
                                                    const uint32_t *__restrict__ zin, uint4 *__restrict__ zlog,
                                                    AlnOut *__restrict__ outs, PathEnt *__restrict__ paths, int keep_paths,
                                                    int32_t *__restrict__ info, int prio_rows) {
    // (the fail list keeps the order of the zero level's work list -- longest first: the longest chains of dependent rows start
    // first and run beside the throughput work of the others)
    const int w = blockIdx.x, lane = threadIdx.x;
    const ZlWave H = hdr[w];
    if (H.mt <= 0) return;
    if (H.mt >= prio_rows) __builtin_amdgcn_s_setprio(2);
    const int n_list = min(*n_dev, n_cap);
    const int wi = w * 64 + lane;
    int a_ = wi < n_list ? list[wi] : -1;
    if (a_ >= 0 && outs[a_].exit_min != D1_CLEAN) a_ = -1;
    const bool live = a_ >= 0;
    {       // info[3]: alignments looked at, info[4]: finished here (vpr_timing::n_lane1_*)
        const int nl = __popcll(__ballot(live));
        if (lane == 0 && nl) atomicAdd(info + 3, nl);
    }
    const int a = max(a_, 0);
    const AlnDesc *dp = descs + a;
    const int Lq = live ? dp->Lq : 1, Lr = live ? dp->Lr : 1, Lt = live ? dp->Lt : 0;
    const int L[2] = {Lq, Lr};
    const auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(zin + H.in_off), 0, 256 * (2 * H.mq + 2 * H.mr + H.mt), 0x00020000);
    const auto rlog = __builtin_amdgcn_make_buffer_rsrc(zlog + H.log_off, 0, 2560 * (H.mt + 1), 0x00020000);
    const uint32_t lane4 = uint32_t(lane) << 2, lane8 = uint32_t(lane) << 3, lane16 = uint32_t(lane) << 4;
    const uint32_t posW[2] = {0u, uint32_t(2 * H.mq) << 8}, posC[2] = {uint32_t(H.mq) << 8, uint32_t(2 * H.mq + H.mr) << 8};
    const uint32_t post = uint32_t(2 * H.mq + 2 * H.mr) << 8;
    // log regions: A 16 B per row and lane, B1 8 B, B2 16 B; mt + 1 rows each
    const uint32_t logB1 = uint32_t(H.mt + 1) << 10, logB2 = logB1 + (uint32_t(H.mt + 1) << 9);
    auto in_at = [&](uint32_t off) -> uint32_t { return __builtin_amdgcn_raw_buffer_load_b32(rin, off, 0, 0); };
    auto word_at = [&](int p, int x, bool on) -> uint32_t { return in_at(on ? posW[p] + (uint32_t(x) << 8) + lane4 : ZL_OOB); };
    auto t_at = [&](int t, bool on) -> uint32_t { return in_at(on ? post + (uint32_t(t) << 8) + lane4 : ZL_OOB); };
    bool ok = live && Lt >= 2;
    const int rows = ok ? Lt : 0;
    int tmax = rows;
#pragma unroll
    for (int o = 32; o; o >>= 1) tmax = max(tmax, __shfl_xor(tmax, o));
    if (tmax == 0) return;
    using S2 = std::integral_constant<int, 2>;
    using S4 = std::integral_constant<int, 4>;
#ifdef D1_CLOCKS
    long long ck0 = wall_clock64();
    auto lapck = [&](int k) { const long long c = wall_clock64(); if (lane == 0) { atomicMax(info + 8 + k, int(c - ck0)); } ck0 = c; };
#else
    auto lapck = [&](int) {};
#endif

    // ---------------- pass 1: B0 of the QUERY end cell, rows Lt-1 .. brow0
    int brow0 = 1 << 30;
    {
        int bq[2][4];
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
            for (int s = 0; s < 4; s++) bq[p][s] = -1;
        }
        auto store_row = [&](int t, bool on) {
            zl_u4 v;
            v.x = uint32_t(bq[0][0] + 1) | (uint32_t(bq[0][1] + 1) << 16);
            v.y = uint32_t(bq[0][2] + 1) | (uint32_t(bq[0][3] + 1) << 16);
            v.z = uint32_t(bq[1][0] + 1) | (uint32_t(bq[1][1] + 1) << 16);
            v.w = uint32_t(bq[1][2] + 1) | (uint32_t(bq[1][3] + 1) << 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rlog, on ? (uint32_t(t) << 10) + lane16 : ZL_OOB, 0, 0);
        };
        // row t -> row t - 1 over the first S slots of each plane; false: S = 2 was not enough (nothing committed)
        auto b0_row = [&](auto Sc, int t, bool act, uint32_t tw1, uint32_t tw0) -> bool {
            constexpr int S = decltype(Sc)::value;
            const bool t_allow = zw_fwd_allow(tw0);
            const uint32_t Tb = ZW_BASE(tw1);
            int nq[2][S];
            uint32_t cw[2][S];
            bool hit[2][S];
#pragma unroll
            for (int p = 0; p < 2; p++) {
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const bool alive = act && bq[p][s] >= 0;
                    const uint32_t ww = word_at(p, bq[p][s], alive);
                    cw[p][s] = in_at(alive ? posC[p] + (uint32_t(bq[p][s]) << 8) + lane4 : ZL_OOB);
                    hit[p][s] = alive && ZW_BASE(ww) == Tb;
                    nq[p][s] = (hit[p][s] && bq[p][s] >= 1) ? bq[p][s] - 1 : -1;       // the MAT source keeps the slot
                }
            }
            bool full = false, over = false;
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int o = 1 - p;
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const bool sw = hit[p][s] && t_allow && cw[p][s] != 0u;
                    if (!__any(sw)) continue;
                    over = over || (sw && (cw[p][s] >> 16) == 0xffffu);          // three or more sources
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const int z = int((cw[p][s] >> (16 * c)) & 0xffffu) - 1;
                        const bool on = sw && z >= 0 && z < 0xfffe;
                        if (c == 1 && !__any(on)) continue;
                        int at = -1;
#pragma unroll
                        for (int k = S - 1; k >= 0; k--) at = (nq[o][k] < 0) ? k : at;
#pragma unroll
                        for (int k = 0; k < S; k++) at = (nq[o][k] == z) ? k : at;
                        full = full || (on && at < 0);
#pragma unroll
                        for (int k = 0; k < S; k++) nq[o][k] = (on && at == k) ? z : nq[o][k];
                    }
                }
            }
            if (S < 4 && __any(full)) return false;
            ok = ok && !full && !over;
            bool any = false;
#pragma unroll
            for (int p = 0; p < 2; p++) {
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if (s < S) { bq[p][s] = act ? nq[p][s] : bq[p][s]; any = any || nq[p][s] >= 0; }
                    else bq[p][s] = act ? -1 : bq[p][s];
                }
            }
            // (no cell of row t - 1 reaches the end cell at no cost: B0 ends at row t -- the rows in front are not stored)
            if (act && ok && any) brow0 = t - 1;
            store_row(t - 1, act && ok && any);
            return true;
        };
#ifdef D1_CLOCKS
        int n_rows_ = 0, n_s4_ = 0;
#endif
        for (int t = tmax - 1; t >= 1; t--) {           // row t -> row t - 1
            const bool first = ok && t == rows - 1;
            if (__any(first)) {
                if (first) { bq[0][0] = Lq - 1; brow0 = t; }
                store_row(t, first);
            }
            const bool act = ok && t < rows && brow0 <= t;
            if (!__any(act)) continue;
            const uint32_t tw1 = t_at(t, act), tw0 = t_at(t - 1, act);
            const bool upper = act && (bq[0][2] >= 0 || bq[0][3] >= 0 || bq[1][2] >= 0 || bq[1][3] >= 0);
#ifdef D1_CLOCKS
            n_rows_++;
            if (__any(upper)) n_s4_++;
#endif
            if (__any(upper) || !b0_row(S2{}, t, act, tw1, tw0)) (void)b0_row(S4{}, t, act, tw1, tw0);
        }
#ifdef D1_CLOCKS
        if (lane == 0) { atomicAdd(info + 12, n_rows_); atomicAdd(info + 13, n_s4_); }
#endif
    }
    { const int c_ = __popcll(__ballot(ok)); if (lane == 0 && c_) atomicAdd(info + 5, c_); }
    lapck(0);
    if (!__any(ok)) return;

    // ---------------- pass 2: F0 (dist.cpp:317-381, as k_zero_lane) and G1 (the B0 cells with D = 1), rows 0 .. Lt-1
    {
        int fq[2][4], gq[2][4];
        uint32_t fw[2][4], gw[2][4];
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
            for (int s = 0; s < 4; s++) { fq[p][s] = -1; fw[p][s] = 0; gq[p][s] = -1; gw[p][s] = 0; }
        }
        uint32_t tw0 = 0;
        // builds row t from row t - 1; bp: the B0 positions of row t.  false: S = 2 was not enough (nothing committed)
        auto fg_row = [&](auto Sc, int t, bool act, uint32_t tw1, const zl_u4 bp) -> bool {
            constexpr int S = decltype(Sc)::value;
            const uint32_t Tb = ZW_BASE(tw1);
            const bool t_allow = zw_fwd_allow(tw0);      // of row t - 1
            int nq[2][S];
            uint32_t nw[2][S], ne[2][S];
            bool bad = false, full = false;
            // (the words of the row's B0 cells: their positions come with the row, so they are requested in front of the F0 step and
            // arrive with its words instead of a round trip later)
            int by[2][S];
            by[0][0] = int(bp.x & 0xffffu) - 1; by[0][1] = int(bp.x >> 16) - 1;
            by[1][0] = int(bp.z & 0xffffu) - 1; by[1][1] = int(bp.z >> 16) - 1;
            if constexpr (S == 4) {
                by[0][2] = int(bp.y & 0xffffu) - 1; by[0][3] = int(bp.y >> 16) - 1;
                by[1][2] = int(bp.w & 0xffffu) - 1; by[1][3] = int(bp.w >> 16) - 1;
            }
            uint32_t ywv[2][S];
#pragma unroll
            for (int p = 0; p < 2; p++) {
#pragma unroll
                for (int s = 0; s < S; s++) ywv[p][s] = word_at(p, by[p][s], act && by[p][s] >= 0);
            }
            if (t == 0) {
#pragma unroll
                for (int p = 0; p < 2; p++) {
#pragma unroll
                    for (int s = 0; s < S; s++) { nq[p][s] = -1; nw[p][s] = 0; ne[p][s] = 0; }
                    nq[p][0] = act ? 0 : -1;
                    nw[p][0] = word_at(p, 0, act);
                    ne[p][0] = ZE_HASMAT;
                }
            } else {
                uint32_t mw[2][S], sw[2][S];
                int sz[2][S];
#pragma unroll
                for (int p = 0; p < 2; p++) {
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const bool alive = act && fq[p][s] >= 0;
                        const int c = fq[p][s] + 1;
                        const int z = ZW_PTR(fw[p][s]) + 1;
                        const bool sok = alive && t_allow && zw_fwd_allow(fw[p][s]) && z < L[1 - p];
                        sz[p][s] = sok ? z : -1;
                        mw[p][s] = word_at(p, c, alive && c < L[p]);
                        sw[p][s] = word_at(1 - p, z, sok);
                    }
                }
#pragma unroll
                for (int p = 0; p < 2; p++) {
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const bool h_ = act && fq[p][s] >= 0 && fq[p][s] + 1 < L[p] && ZW_BASE(mw[p][s]) == Tb;
                        nq[p][s] = h_ ? fq[p][s] + 1 : -1;
                        nw[p][s] = mw[p][s];
                        ne[p][s] = h_ ? ZE_HASMAT : 0u;
                    }
                }
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const int o = 1 - p;
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const bool h_ = sz[p][s] >= 0 && ZW_BASE(sw[p][s]) == Tb;
                        if (!__any(h_)) continue;
                        const int z = sz[p][s];
                        int at = -1;
#pragma unroll
                        for (int k = S - 1; k >= 0; k--) at = (nq[o][k] < 0) ? k : at;
#pragma unroll
                        for (int k = 0; k < S; k++) at = (nq[o][k] == z) ? k : at;
                        full = full || (h_ && at < 0);
#pragma unroll
                        for (int k = 0; k < S; k++) {
                            const bool here = h_ && at == k;
                            bad = bad || (here && (ne[o][k] & ZE_HASSWP));
                            nq[o][k] = here ? z : nq[o][k];
                            nw[o][k] = here ? sw[p][s] : nw[o][k];
                            ne[o][k] = here ? (ne[o][k] | ZE_HASSWP | (uint32_t(s) << ZE_PSLOT_SHIFT)) : ne[o][k];
                        }
                    }
                }
                if (S < 4 && __any(full)) return false;
            }
            // the B0 cells of row t and which of them have D = 1
            int ngq[2][S];
            uint32_t ngw[2][S], ge[2][S];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int o = 1 - p;
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const int y = by[p][s];
                    const bool on = act && y >= 0;
                    uint32_t e = 0;
                    const uint32_t yw = ywv[p][s];
                    if (__any(on)) {
                        const bool bm = on && ZW_BASE(yw) == Tb;
#pragma unroll
                        for (int k = 0; k < S; k++) {
                            if (t > 0) {
                                if (on && fq[p][k] >= 0 && fq[p][k] == y) e |= GE_DEL | (uint32_t(k) << GE_DEL_SHIFT);
                                if (on && fq[p][k] >= 0 && fq[p][k] == y - 1) e |= GE_SUB | (uint32_t(k) << GE_SUB_SHIFT);
                            }
                            if (on && nq[p][k] >= 0 && nq[p][k] == y - 1) e |= GE_INS | (uint32_t(k) << GE_INS_SHIFT);
                            bad = bad || (on && nq[p][k] == y);          // (a B0 cell inside F0: cannot be, s would be 0)
                        }
                        if (t > 0) {
                            if (bm && gq[p][s] >= 0 && gq[p][s] == y - 1) e |= GE_MAT;
#pragma unroll
                            for (int k = 0; k < S; k++) {
                                const bool src = bm && t_allow && gq[o][k] >= 0 && zw_fwd_allow(gw[o][k]) && ZW_PTR(gw[o][k]) + 1 == y;
                                bad = bad || (src && (e & GE_SWP));       // two D = 1 sources: the reference's container order decides
                                e = src ? (e | GE_SWP | (uint32_t(k) << GE_PSLOT_SHIFT)) : e;
                            }
                        }
                        if (e) e |= GE_ALIVE | ((p == 0 && (yw & ZW_TP)) ? GE_TP : 0u) | (zw_bwd_allow(yw) ? GE_BWD : 0u);
                    }
                    ge[p][s] = e;
                    ngq[p][s] = e ? y : -1;
                    ngw[p][s] = yw;
                }
            }
            ok = ok && !bad && !full;
            {
                zl_u2 ev;
                uint32_t wq = 0, wr_ = 0;
#pragma unroll
                for (int sl = 0; sl < S; sl++) {
                    wq |= (nq[0][sl] < 0 ? 0u : (ZE_ALIVE | ne[0][sl] | ((nw[0][sl] & ZW_TP) ? ZE_TP : 0u) | (zw_bwd_allow(nw[0][sl]) ? ZE_BWD : 0u))) << (8 * sl);
                    wr_ |= (nq[1][sl] < 0 ? 0u : (ZE_ALIVE | ne[1][sl] | (zw_bwd_allow(nw[1][sl]) ? ZE_BWD : 0u))) << (8 * sl);
                }
                ev.x = wq; ev.y = wr_;
                __builtin_amdgcn_raw_buffer_store_b64(ev, rlog, (act && ok) ? logB1 + (uint32_t(t) << 9) + lane8 : ZL_OOB, 0, 0);
                zl_u4 gv;
                gv.x = ge[0][0] | (ge[0][1] << 16); gv.z = ge[1][0] | (ge[1][1] << 16);
                gv.y = 0; gv.w = 0;
                if constexpr (S == 4) { gv.y = ge[0][2] | (ge[0][3] << 16); gv.w = ge[1][2] | (ge[1][3] << 16); }
                __builtin_amdgcn_raw_buffer_store_b128(gv, rlog, (act && ok) ? logB2 + (uint32_t(t) << 10) + lane16 : ZL_OOB, 0, 0);
            }
#pragma unroll
            for (int p = 0; p < 2; p++) {
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if (s < S) {
                        fq[p][s] = act ? nq[p][s] : fq[p][s]; fw[p][s] = act ? nw[p][s] : fw[p][s];
                        gq[p][s] = act ? ngq[p][s] : gq[p][s]; gw[p][s] = act ? ngw[p][s] : gw[p][s];
                    } else {
                        fq[p][s] = act ? -1 : fq[p][s]; gq[p][s] = act ? -1 : gq[p][s];
                    }
                }
            }
            tw0 = act ? tw1 : tw0;
            return true;
        };
        // (the B0 positions and the truth word of a row are known in advance: requested one row ahead)
        auto b_row = [&](int t) -> zl_u4 { return __builtin_amdgcn_raw_buffer_load_b128(rlog, (ok && t < rows && t >= brow0) ? (uint32_t(t) << 10) + lane16 : ZL_OOB, 0, 0); };
        zl_u4 bp_n = b_row(0);
        uint32_t tw_n = t_at(0, ok && 0 < rows);
        for (int t = 0; t < tmax; t++) {
            const bool act = ok && t < rows;
            const uint32_t tw1 = tw_n;
            const zl_u4 bp = bp_n;
            bp_n = b_row(t + 1);
            tw_n = t_at(t + 1, ok && t + 1 < rows);
            bool upper = act && ((bp.y | bp.w) != 0u);
#pragma unroll
            for (int p = 0; p < 2; p++) upper = upper || (act && (fq[p][2] >= 0 || fq[p][3] >= 0 || gq[p][2] >= 0 || gq[p][3] >= 0));
            if (__any(upper) || !fg_row(S2{}, t, act, tw1, bp)) (void)fg_row(S4{}, t, act, tw1, bp);
        }
        { const int c_ = __popcll(__ballot(ok)); if (lane == 0 && c_) atomicAdd(info + 6, c_); }
        // s = 1 iff the QUERY end cell (slot 0 of its row's B0 cells) has D = 1
        ok = ok && gq[0][0] == Lq - 1;
        { const int c_ = __popcll(__ballot(ok)); if (lane == 0 && c_) atomicAdd(info + 7, c_); }
    }
    lapck(1);
    if (live) {
        AlnOut &o = outs[a];
        o.dist_q = ok ? 1 : D_INF;
        o.dist_r = D_INF;                      // (not looked at: the QUERY plane's end cell is preferred, dist.cpp:436-439)
        o.exit_min = ok ? D_INF : 0;           // k_fwd_band_finish: accepted iff s = 1 was found here
    }
    if (!__any(ok)) return;

    // ---------------- pass 3: max-TP scores over both layers, rows Lt-1 .. 0, and the walk's move out of every cell.
    // A cell's state is one key: score << 5 | (7 - rank of the move by the walk's priority) << 2 | slot the move lands in;
    // the largest key is the best score and among equal scores the move the walk prefers (dist.cpp:907-935).  -1: no path.
    const int nrow = ok ? Lt : 0;
    int bmax = nrow;
#pragma unroll
    for (int o = 32; o; o >>= 1) bmax = max(bmax, __shfl_xor(bmax, o));
    int beg_plane = VPR_PLANE_REF;
    {
        int kf[2][4], kg[2][4];         // keys of the F0 / G1 cells of row t + 1
        uint32_t cf[2] = {0, 0};        // F0 flag bytes of row t + 1
        uint32_t cg[2][2] = {{0, 0}, {0, 0}};      // G1 flag halfwords of row t + 1: [plane][slots 0-1 / 2-3]
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
            for (int s = 0; s < 4; s++) { kf[p][s] = -1; kg[p][s] = -1; }
        }
        auto key = [](int from, int tp, int rank, int slot) -> int { return (((from >> 5) + tp) << 5) | ((7 - rank) << 2) | slot; };
        auto sc_row = [&](auto Sc, int t, bool act, bool first, const zl_u2 pf, const zl_u4 pgv) {
            constexpr int S = decltype(Sc)::value;
            int bf[2][S], bg[2][S];
#pragma unroll
            for (int p = 0; p < 2; p++) {
#pragma unroll
                for (int s = 0; s < S; s++) { bf[p][s] = -1; bg[p][s] = -1; }
            }
            if (first) bg[0][0] = 0x1f;      // the end cell: score 0 (dist.cpp:538-546); its own move byte is never read
            if (__any(act && !first)) {
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const int o = 1 - p;
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        {       // moves into the G1 cell (p, s) of row t + 1
                            const uint32_t e = (cg[p][s >> 1] >> (16 * (s & 1))) & 0xffffu;
                            const bool on = act && !first && kg[p][s] >= 0;
                            if (__any(on)) {
                                const int tp = (e & GE_TP) ? 1 : 0;
                                if (on && (e & GE_MAT)) bg[p][s] = max(bg[p][s], key(kg[p][s], tp, MV_MAT, s));
                                const bool sw = on && (e & GE_SWP) && (e & GE_BWD);        // dist.cpp:599-602
                                const int ps = int((e >> GE_PSLOT_SHIFT) & 3u);
                                const int ksw = key(kg[p][s], p == 1 ? 0 : tp, o == 1 ? MV_SWP_R : MV_SWP_Q, s);   // leaving a REF cell scores 0 (dist.cpp:614)
                                const bool su = on && (e & GE_SUB), de = on && (e & GE_DEL);
                                const int ss = int((e >> GE_SUB_SHIFT) & 3u), ds = int((e >> GE_DEL_SHIFT) & 3u);
                                const int ksu = key(kg[p][s], tp, MV_SUB, s), kde = key(kg[p][s], 0, MV_DEL, s);
#pragma unroll
                                for (int k = 0; k < S; k++) {
                                    bg[o][k] = (sw && ps == k) ? max(bg[o][k], ksw) : bg[o][k];
                                    bf[p][k] = (su && ss == k) ? max(bf[p][k], ksu) : bf[p][k];
                                    bf[p][k] = (de && ds == k) ? max(bf[p][k], kde) : bf[p][k];
                                }
                            }
                        }
                        {       // moves into the F0 cell (p, s) of row t + 1 (as k_zero_lane's backward pass)
                            const uint32_t e = (cf[p] >> (8 * s)) & 0xffu;
                            const bool on = act && !first && kf[p][s] >= 0;
                            if (__any(on)) {
                                const int tp = (e & ZE_TP) ? 1 : 0;
                                if (on && (e & ZE_HASMAT)) bf[p][s] = max(bf[p][s], key(kf[p][s], tp, MV_MAT, s));
                                const bool sw = on && (e & ZE_HASSWP) && (e & ZE_BWD);
                                const int ps = int((e >> ZE_PSLOT_SHIFT) & 3u);
                                const int ksw = key(kf[p][s], p == 1 ? 0 : tp, o == 1 ? MV_SWP_R : MV_SWP_Q, s);
#pragma unroll
                                for (int k = 0; k < S; k++) bf[o][k] = (sw && ps == k) ? max(bf[o][k], ksw) : bf[o][k];
                            }
                        }
                    }
                }
            }
            const uint32_t pgw[2][2] = {{pgv.x, pgv.y}, {pgv.z, pgv.w}};
            // INS moves inside row t: from the F0 cell in front of a G1 cell
#pragma unroll
            for (int p = 0; p < 2; p++) {
#pragma unroll
                for (int s = 0; s < S; s++) {
                    const uint32_t e = (pgw[p][s >> 1] >> (16 * (s & 1))) & 0xffffu;
                    const bool in = act && bg[p][s] >= 0 && (e & GE_INS);
                    if (!__any(in)) continue;
                    const int is = int((e >> GE_INS_SHIFT) & 3u);
                    const int kin = key(bg[p][s], (e & GE_TP) ? 1 : 0, MV_INS, s);
#pragma unroll
                    for (int k = 0; k < S; k++) bf[p][k] = (in && is == k) ? max(bf[p][k], kin) : bf[p][k];
                }
            }
            // the move bytes of row t: F0 cells, then G1 cells
            {
                zl_u4 mv;
                uint32_t b[4] = {0xffff0000u, 0xffff0000u, 0xffff0000u, 0xffff0000u};
                if constexpr (S == 4) { b[0] = b[1] = b[2] = b[3] = 0u; }
#pragma unroll
                for (int p = 0; p < 2; p++) {
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const uint32_t mf = bf[p][s] < 0 ? 0xffu : (uint32_t(7 - ((bf[p][s] >> 2) & 7)) | (uint32_t(bf[p][s] & 3) << 3));
                        const uint32_t mg = bg[p][s] < 0 ? 0xffu : (uint32_t(7 - ((bg[p][s] >> 2) & 7)) | (uint32_t(bg[p][s] & 3) << 3));
                        b[p] |= mf << (8 * s);
                        b[2 + p] |= mg << (8 * s);
                    }
                }
                mv.x = b[0]; mv.y = b[1]; mv.z = b[2]; mv.w = b[3];
                __builtin_amdgcn_raw_buffer_store_b128(mv, rlog, act ? (uint32_t(t) << 10) + lane16 : ZL_OOB, 0, 0);
            }
            if (act) {
#pragma unroll
                for (int p = 0; p < 2; p++) {
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        if (s < S) { kf[p][s] = bf[p][s]; kg[p][s] = bg[p][s]; }
                        else { kf[p][s] = -1; kg[p][s] = -1; }
                    }
                    cg[p][0] = pgw[p][0]; cg[p][1] = pgw[p][1];
                }
                cf[0] = pf.x; cf[1] = pf.y;
            }
        };
        // (the log entries of a row do not depend on what the rows behind it gave: requested one row ahead)
        auto f_row = [&](int t) -> zl_u2 { return __builtin_amdgcn_raw_buffer_load_b64(rlog, (t >= 0 && t < nrow) ? logB1 + (uint32_t(t) << 9) + lane8 : ZL_OOB, 0, 0); };
        auto g_row = [&](int t) -> zl_u4 { return __builtin_amdgcn_raw_buffer_load_b128(rlog, (t >= 0 && t < nrow) ? logB2 + (uint32_t(t) << 10) + lane16 : ZL_OOB, 0, 0); };
        zl_u2 pf_n = f_row(bmax - 1);
        zl_u4 pg_n = g_row(bmax - 1);
        for (int t = bmax - 1; t >= 0; t--) {
            const bool act = t < nrow;
            const bool first = act && t == nrow - 1;
            const zl_u2 pf = pf_n;
            const zl_u4 pgv = pg_n;
            pf_n = f_row(t - 1);
            pg_n = g_row(t - 1);
            // upper slots in use in either row (by any lane)?
            const bool upper = act && ((((cf[0] | cf[1] | pf.x | pf.y) & 0xffff0000u) | cg[0][1] | cg[1][1] | pgv.y | pgv.w) != 0u);
#ifdef D1_CLOCKS
            if (lane == 0) { atomicAdd(info + 14, 1); if (__any(upper)) atomicAdd(info + 15, 1); }
#endif
            if (__any(upper)) sc_row(S4{}, t, act, first, pf, pgv);
            else sc_row(S2{}, t, act, first, pf, pgv);
        }
        // (QUERY, 0, 0) on a path to the end?  dist.cpp:811-814
        beg_plane = kf[0][0] >= 0 ? VPR_PLANE_QUERY : VPR_PLANE_REF;
        ok = ok && (kf[0][0] >= 0 || kf[1][0] >= 0);
    }

    lapck(2);
    // (pass 4, the walk, is a kernel of its own -- k_one_walk, on the side stream in front of k_one_credit -- like the zero level's:
    // this launch is on the short part's chain)
    if (ok) { AlnOut &o = outs[a]; o.beg_plane = beg_plane; o.path_len = -1; }       // (-1: walk pending)
}

// ===========================================================================
// K1Lw: pass 4 of the distance-1 lane level -- the walk + sync flags (dist.cpp:865-998) along the move bytes k_one_lane left; the
// steps are the 8-byte records of k_zero_walk plus an edit bit and "truth row one behind the index" (after an INS step)
// ===========================================================================
__global__ void __launch_bounds__(64, 8) k_one_walk(const AlnDesc *__restrict__ descs, const int32_t *__restrict__ list,
                                                    const int32_t *__restrict__ n_dev, int n_cap, const ZlWave *__restrict__ hdr,
                                                    const uint32_t *__restrict__ zin, uint4 *__restrict__ zlog,
                                                    AlnOut *__restrict__ outs, PathEnt *__restrict__ paths, int keep_paths,
                                                    int32_t *__restrict__ info, int prio_rows) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const ZlWave H = hdr[w];
    if (H.mt <= 0) return;
    if (H.mt >= prio_rows) __builtin_amdgcn_s_setprio(2);
    const int n_list = min(*n_dev, n_cap);
    const int wi = w * 64 + lane;
    const int a_ = wi < n_list ? list[wi] : -1;
    const int a = max(a_, 0);
    const AlnDesc *dp = descs + a;
    const bool ok = a_ >= 0 && outs[a].band_ok == D1_TAG && outs[a].path_len == -1;
    const int Lq = ok ? dp->Lq : 1, Lr = ok ? dp->Lr : 1;
    const int L[2] = {Lq, Lr};
    const int nrow = ok ? dp->Lt : 0;
    int bmax = nrow;
#pragma unroll
    for (int o = 32; o; o >>= 1) bmax = max(bmax, __shfl_xor(bmax, o));
    if (bmax == 0) return;
    const auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(zin + H.in_off), 0, 256 * (2 * H.mq + 2 * H.mr + H.mt), 0x00020000);
    const auto rlog = __builtin_amdgcn_make_buffer_rsrc(zlog + H.log_off, 0, 2560 * (H.mt + 1), 0x00020000);
    const uint32_t lane4 = uint32_t(lane) << 2, lane8 = uint32_t(lane) << 3, lane16 = uint32_t(lane) << 4;
    const uint32_t posW[2] = {0u, uint32_t(2 * H.mq) << 8};
    const uint32_t post = uint32_t(2 * H.mq + 2 * H.mr) << 8;
    const uint32_t logB1 = uint32_t(H.mt + 1) << 10;
    auto in_at = [&](uint32_t off) -> uint32_t { return __builtin_amdgcn_raw_buffer_load_b32(rin, off, 0, 0); };
    auto word_at = [&](int p, int x, bool on) -> uint32_t { return in_at(on ? posW[p] + (uint32_t(x) << 8) + lane4 : ZL_OOB); };
    auto t_at = [&](int t, bool on) -> uint32_t { return in_at(on ? post + (uint32_t(t) << 8) + lane4 : ZL_OOB); };
    const int beg_plane = ok ? outs[a].beg_plane : 0;
    auto lapck = [&](int) {};
    {
        PathEnt *path = paths + dp->path_off;
        const uint32_t logS0 = logB1;           // the steps replace the F0 flag bytes
        int hi = beg_plane, slot = 0, layer = 0, x = 0, ti = 0, mv_in = -1;
        uint32_t status = 0;
        bool wok = ok, done = !ok;
        int plen = 0;
        for (int i = 0; i <= bmax; i++) {
            const bool act = wok && !done;
            if (!__any(act)) break;
            const uint32_t mvw = __builtin_amdgcn_raw_buffer_load_b32(rlog, act ? (uint32_t(ti) << 10) + lane16 + (uint32_t(layer) << 3) + (uint32_t(hi) << 2) : ZL_OOB, 0, 0);
            const uint32_t tw = t_at(ti, act);
            const uint32_t cw = word_at(hi, x, act);
            const int trv = ZW_PTR(tw), qref = hi ? x : ZW_PTR(cw);
            uint32_t sync = 1, edit = 0;
            if (mv_in >= 0) {
                // dist.cpp:949-968.  An INS step consumes no reference base: a variant's first base then counts as inside it --
                // and the step is no sync point anyway, like a DEL step
                const bool diag = mv_in != MV_INS && mv_in != MV_DEL;
                const bool in_t = (tw & ZW_PV) && !(tw & ZW_PB);
                const bool in_q = hi == 0 && (cw & ZW_PV) && !(cw & ZW_PB);
                sync = (diag && !in_t && !in_q && !((tw | cw) & ZW_INS) && trv == qref) ? 1u : 0u;
                edit = (mv_in == MV_SUB || mv_in == MV_INS || mv_in == MV_DEL) ? 1u : 0u;
            }
            const uint32_t eqb = (ZW_BASE(cw) == ZW_BASE(tw) ? 1u : 0u) | ((hi || !(cw & ZW_PV)) ? 2u : 0u);
            zl_u2 st;
            st.x = uint32_t(x) | (uint32_t(hi) << 16) | (sync << 17) | (eqb << 18) | (edit << 20) | (uint32_t(i - ti) << 21);
            st.y = uint32_t(qref + 1) | (uint32_t(trv + 1) << 16);
            __builtin_amdgcn_raw_buffer_store_b64(st, rlog, act ? logS0 + (uint32_t(i) << 9) + lane8 : ZL_OOB, 0, 0);
            if ((keep_paths & 1) && act) {
                uint4 pe;
                pe.x = uint32_t(x) | (uint32_t(hi) << 31);
                pe.y = uint32_t(ti) | (sync << 31) | (edit << 30);
                pe.z = uint32_t(qref);
                pe.w = uint32_t(trv);
                *reinterpret_cast<uint4 *>(path + i) = pe;
            }
            if (act) {
                if (layer == 1 && hi == 0 && x == Lq - 1 && ti == nrow - 1) { done = true; plen = i + 1; }
                else {
                    const uint32_t m = (mvw >> (8 * slot)) & 0xffu;
                    const int rank = int(m & 7u), to = int((m >> 3) & 3u);
                    if (m == 0xffu) { status |= VPR_ST_ERR_NO_PTR; wok = false; }
                    else if (rank == MV_MAT) { x++; ti++; }
                    else if (rank == MV_SWP_R || rank == MV_SWP_Q) { x = ZW_PTR(cw) + 1; hi = 1 - hi; slot = to; ti++; }
                    else if (rank == MV_SUB) { x++; ti++; layer = 1; slot = to; }
                    else if (rank == MV_INS) { x++; layer = 1; slot = to; }
                    else { ti++; layer = 1; slot = to; }          // MV_DEL
                    mv_in = rank;
                    if (ti >= nrow || x >= L[hi]) { status |= VPR_ST_ERR_NO_PTR; wok = false; }
                }
            }
        }
        lapck(3);
        {
            const int nf = __popcll(__ballot(ok && wok && done));
            if (lane == 0 && nf) atomicAdd(info + 4, nf);
        }
        if (ok) {
            AlnOut &o = outs[a];
            wok = wok && done;
            if (!wok) status |= VPR_ST_ERR_NO_PTR;
            o.beg_plane = beg_plane;
            o.path_len = wok ? plen : 0;
            if (!wok) o.n_sec = 0;
            if (status) atomicOr(&o.status, status);
        }
    }
}

// ===========================================================================
// K1Lc: credit sections of the alignments k_one_lane finished (as k_zero_credit; a step carries its edit bit and how far its
// truth row lies behind its index -- one behind an INS step)
// ===========================================================================
struct D1Fetch {
    const uint2 *log;
    int64_t pre_i;
    uint2 pre;
    __device__ PathEnt operator()(int64_t i) {
        const uint2 v = (i == pre_i) ? pre : log[i * 64];
        if (i > 0) { pre_i = i - 1; pre = log[(i - 1) * 64]; }
        PathEnt e;
        e.a = (v.x & 0xffffu) | (((v.x >> 16) & 1u) << 31);
        e.b = uint32_t(i - int64_t((v.x >> 21) & 1u)) | (((v.x >> 17) & 1u) << 31) | (((v.x >> 20) & 1u) << 30) | (((v.x >> 18) & 3u) << 28);
        e.qref = int(v.y & 0xffffu) - 1;
        e.tref = int(v.y >> 16) - 1;
        return e;
    }
};

__global__ void __launch_bounds__(64) k_one_credit(DevBatch B, const AlnDesc *__restrict__ descs, const int32_t *__restrict__ list,
                                                   const int32_t *__restrict__ n_dev, int n_cap, const ZlWave *__restrict__ hdr,
                                                   const uint4 *__restrict__ zlog, AlnOut *__restrict__ outs, Section *__restrict__ secs,
                                                   int32_t *const *__restrict__ fp_group, EdJob *__restrict__ jobs,
                                                   int32_t *__restrict__ n_jobs, int32_t jobs_cap, int ztag) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const ZlWave H = hdr[w];
    if (H.mt <= 0) return;
    const int wi = w * 64 + lane;
    if (wi >= min(*n_dev, n_cap)) return;
    const int a = list[wi];
    if (a < 0) return;
    const AlnDesc d = descs[a];
    AlnOut &O = outs[a];
    if (d.band_pad != ztag || O.band_ok != D1_TAG) return;
    if (O.status & (VPR_ST_ERR_NO_PTR | VPR_ST_ERR_LIMIT)) return;
    D1Fetch f{reinterpret_cast<const uint2 *>(zlog + H.log_off + 64 * (int64_t(H.mt) + 1)) + lane, -1, make_uint2(0, 0)};     // behind region A
    credit_walk<false, D1Fetch, true>(B, d, O, a, nullptr, int64_t(O.path_len), 0u, secs, fp_group, jobs, n_jobs, jobs_cap, true, f);
}

#endif
