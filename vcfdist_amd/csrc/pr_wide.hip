// pr_wide.hip -- the striped forward sweep for the wide windows (256 / 1024 cells): NW wavefronts per alignment.
//
// A single wave pays ~4 cycles per instruction on a dependent chain, so a window of C cells per lane costs C times
// the row latency of the 64-cell kernel (k_fwd_band<4>: ~1.3 us per truth row, and one long alignment that needs
// 256 cells is the critical path of a whole batch).  Here the window is spread over NW waves of one workgroup, one
// cell per lane and plane exactly as in k_fwd_stripe, with ONE workgroup barrier per truth row:
//   * every wave scans its 64 cells (DPP prefix-min), publishes the scan values (rowbuf) and its total, barrier;
//   * after the barrier each wave folds the totals of the waves below it into its carry;
//   * the value any other wave needs from the previous row (swap source, re-alignment at a stripe boundary) is
//     reconstructed as min(rowbuf[col], carry of col's wave) + col, and the diagonal neighbour of a wave's first
//     lane is simply carry + col - 1: no second barrier.
// Same window origins (constant over stripes of WD_K rows), exit test and flag layout ([t][pitch] rows, one origin
// per row and plane) as the ring kernels it replaces, so K2b / K3 consume its output unchanged.
#ifndef PR_WIDE_HIP_
#define PR_WIDE_HIP_

#define WD_K 8

template <int W>
__device__ __forceinline__ void wide_origin(const int32_t *t2r, const uint16_t *tj, const int32_t *r2q, int s, int n_stripes, int Lt,
                                            int Lq, int Lr, int &loQ, int &loR) {
    loQ = 0; loR = 0;
    if (s > 0 && s < n_stripes) {   // stripe 0 starts at the origin
        const int ta = s * WD_K, tb = min(ta + WD_K - 1, Lt - 1);
        const int ra = t2r[ta], rb = t2r[tb];
        const int qa = query_center(t2r, tj, r2q, ta, Lr), qb = query_center(t2r, tj, r2q, tb, Lr);
        loR = max(0, min((ra + rb) / 2 - W / 2, Lr - min(W, Lr)));
        loQ = max(0, min((qa + qb) / 2 - W / 2, Lq - min(W, Lq)));
    }
}

template <int NW>
__global__ void __launch_bounds__(NW * 64) k_fwd_wide(DevBatch B, const AlnDesc *__restrict__ descs,
                                                      const int32_t *__restrict__ work, uint8_t *__restrict__ ws,
                                                      int32_t *__restrict__ blo_all, AlnOut *__restrict__ outs) {
    constexpr int W = NW * 64;
    __builtin_amdgcn_s_setprio(2);      // a latency chain (rows are sequential): win issue arbitration against the bulk kernels
    const int a = work[blockIdx.x];
    const AlnDesc d = descs[a];
    const int c = threadIdx.x;                     // window column of this thread
    const int lane = c & 63;
    const int wave = __builtin_amdgcn_readfirstlane(c >> 6);
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
    const int2 *xbp[2] = {B.xb_q[d.qs] + d.q_off, B.xb_r[d.qs] + d.r_off};     // budgets of the exit test (pr_device.h)
    const ExitEnd xend = exit_end(q2r, t2r, Lq, Lr, Lt);      // end-cell coordinates of the exit test (exit_key)
    uint8_t *mat[2] = {ws + d.mat_off[0], ws + d.mat_off[1]};
    int32_t *blo = blo_all + d.blo_off;
    const int n_stripes = (Lt + WD_K - 1) / WD_K;
    __shared__ int32_t rowbuf[2][2][W];            // [row parity][plane][column]: wave-local inclusive scan of D - column
    __shared__ int32_t totals[2][2][16];           // [row parity][plane][wave]: the wave's scan total
    __shared__ __align__(16) uint8_t fbuf[2][NW][WD_K * 64];   // a stripe's flag bytes, one 64-column slab per wave
    __shared__ int32_t red[NW + 2];

    // stripe origins, 64 stripes per register chunk (lane l <-> stripe c0 + l), replicated per wave
    int cbQ, cbR, nbQ, nbR;
    wide_origin<W>(t2r, tjp, r2q, lane, n_stripes, Lt, Lq, Lr, cbQ, cbR);
    wide_origin<W>(t2r, tjp, r2q, 64 + lane, n_stripes, Lt, Lq, Lr, nbQ, nbR);
    uint32_t tchunk = 0, tlast = 0;
    if (lane < Lt) {
        tchunk = uint32_t(Ts[lane]) | (uint32_t(Tf[lane]) << 8);
    }

    int exit_min = D_INF;
    int Dp[2] = {c, c};                  // row 0: D = x along the INS chain (origin 0)
    int lo[2] = {0, 0}, hi[2] = {min(Lq, W) - 1, min(Lr, W) - 1};
    int plo[2] = {0, 0};                 // origins of the previous stripe
    int nlo[2] = {0, 0}, nhi[2] = {0, 0};
    int2 kc[2], kn[2];
    int rhoc[2], rhon[2];
    int2 vac[2], van[2];
    int pin[2] = {D_INF, D_INF};         // lane l < NW: min of the scan totals of waves 0..l of the previous row
    int carry_prev[2] = {D_INF, D_INF};  // this wave's carry of the previous row (uniform)
#pragma unroll
    for (int p = 0; p < 2; p++) {
        kc[p] = make_int2(-1, int(0xffffffffu));
        rhoc[p] = c; vac[p] = make_int2(0, 0);
        if (c <= hi[p]) {
            kc[p] = fk[p][c];
            rhoc[p] = (p == 0) ? q2r[c] : c;
            vac[p] = xbp[p][c];
        }
    }
    // D of window column cs (columns of the origin the previous row was computed with) in the previous row
    auto gather = [&](int par, int p, int cs) -> int {
        const bool ok = unsigned(cs) < unsigned(W);
        const int cc = ok ? cs : 0;
        const int loc = rowbuf[par][p][cc];
        const int wsrc = cc >> 6;
        const int car0 = __builtin_amdgcn_ds_bpermute(((wsrc - 1) & 63) << 2, pin[p]);
        const int car = wsrc > 0 ? car0 : D_INF;
        return ok ? min(loc, car) + cc : D_INF;
    };

    for (int s = 0; s < n_stripes; s++) {
        const int t0 = s * WD_K;
        const int rows = min(WD_K, Lt - t0);
        const bool has_next = s + 1 < n_stripes;
        if (has_next) {
            if (((s + 1) & 63) == 0) { nlo[0] = __builtin_amdgcn_readlane(nbQ, 0); nlo[1] = __builtin_amdgcn_readlane(nbR, 0); }
            else { nlo[0] = __builtin_amdgcn_readlane(cbQ, (s + 1) & 63); nlo[1] = __builtin_amdgcn_readlane(cbR, (s + 1) & 63); }
        } else { nlo[0] = lo[0]; nlo[1] = lo[1]; }
        nhi[0] = min(Lq - 1, nlo[0] + W - 1);
        nhi[1] = min(Lr - 1, nlo[1] + W - 1);
#pragma unroll
        for (int p = 0; p < 2; p++) {
            kn[p] = make_int2(-1, int(0xffffffffu));
            rhon[p] = 0; van[p] = make_int2(0, 0);
            if (has_next && nlo[p] + c <= nhi[p]) {
                const int xn = nlo[p] + c;
                kn[p] = fk[p][xn];
                rhon[p] = (p == 0) ? q2r[xn] : xn;
                van[p] = xbp[p][xn];
            }
        }
        if (wave == 0 && lane < rows) { blo[t0 + lane] = lo[0]; blo[Lt + t0 + lane] = lo[1]; }   // read by K2 / K3

        // ---- per-thread constants of this stripe
        int s0[2];
        uint32_t base[2];
        bool multi[2];
        int ex_in[2], ex_last[2];          // edges of the cell that leave the window (bits as in exit_key, k_fwd_stripe)
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int o = 1 - p;
            const int x = lo[p] + c;
            const bool valid = x <= hi[p];
            s0[p] = (kc[p].x < 0) ? -1 : (kc[p].x & (FK_MULTI - 1));
            multi[p] = (kc[p].x >= 0) & ((kc[p].x & FK_MULTI) != 0);
            base[p] = valid ? (uint32_t(kc[p].y) >> 24) : 0xffu;
            const int z = kc[p].y & 0xffffff;
            const bool zok = valid & (z < Lp[o]);
            const bool ins_out = valid & (x == hi[p]) & (hi[p] < Lp[p] - 1);
            const bool z_out = unsigned(z - lo[o]) > unsigned(hi[o] - lo[o]);
            const bool z_out_n = unsigned(z - nlo[o]) > unsigned(nhi[o] - nlo[o]);
            const bool dg_out_n = (x + 1 < Lp[p]) & ((x + 1 < nlo[p]) | (x + 1 > nhi[p]));
            const bool del_out_n = (x < nlo[p]) | (x > nhi[p]);
            ex_in[p] = (ins_out ? 3 : 0) | ((zok & z_out) ? 8 : 0);
            ex_last[p] = (ins_out ? 1 : 0) | ((has_next & valid & dg_out_n) ? 2 : 0) | ((has_next & valid & del_out_n) ? 4 : 0) |
                         ((has_next & zok & z_out_n) ? 8 : 0);
        }
        int rowo = lane;                   // byte offset of this thread's cell inside its wave's slab
        const bool st_ok[2] = {c < d.pitch[0], c < d.pitch[1]};

        for (int r = 0; r < rows; r++) {
            const int t = t0 + r;
            const int par = t & 1;
            const bool last = (r == rows - 1);
            if ((t & 63) == 0 && t > 0) {
                tlast = __builtin_amdgcn_readlane(tchunk, 63);
                const int tt = t + lane;
                tchunk = 0;
                if (tt < Lt) {
                    tchunk = uint32_t(Ts[tt]) | (uint32_t(Tf[tt]) << 8);
                }
            }
            int v[2] = {0, 0}, incl[2] = {0, 0};
            uint32_t mk[2] = {0, 0};
            if (t > 0) {
                const uint32_t cur = __builtin_amdgcn_readlane(tchunk, t & 63);
                const uint32_t prv = ((t & 63) == 0) ? tlast : uint32_t(__builtin_amdgcn_readlane(tchunk, (t - 1) & 63));
                const uint32_t Tt = cur & 0xff;
                const bool at = fwd_allow(int((prv >> 8) & 0xff));
                const bool first = (r == 0);   // the previous row belongs to the previous stripe (origins plo)
                const int pp = par ^ 1;        // parity of the previous row
                int up[2], dg[2], sw[2];
                bool match[2], need_multi = false;
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const int o = 1 - p;
                    if (first) {
                        const int sh = lo[p] - plo[p];
                        up[p] = gather(pp, p, c + sh);
                        dg[p] = gather(pp, p, c + sh - 1);
                    } else {
                        up[p] = Dp[p];
                        const int dn = wave_shr1(Dp[p], D_INF);
                        dg[p] = (lane == 0) ? ((wave > 0) ? carry_prev[p] + c - 1 : D_INF) : dn;
                    }
                    match[p] = base[p] == Tt;
                    const bool on = match[p] & at & (s0[p] >= 0);
                    const int sv = gather(pp, o, s0[p] - (first ? plo[o] : lo[o]));
                    sw[p] = on ? sv : D_INF;
                    need_multi = need_multi || (on && multi[p]);
                }
                uint32_t swbits[2] = {0, 0};
                if (__builtin_expect(__any(need_multi), 0)) {
                    // rare: several allowed swap sources; keep the highest index among the optimal ones, remember ties
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        const int o = 1 - p;
                        const bool need = match[p] && at && s0[p] >= 0 && multi[p];
                        int4 cc = make_int4(-1, -1, -1, -1);
                        if (need) cc = cand[p][lo[p] + c];
                        const int olo = first ? plo[o] : lo[o];
                        const int srcs[3] = {cc.y, cc.z, cc.w};
                        int choice = 0;
                        bool tie = false;
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            const int val0 = gather(pp, o, srcs[k] - olo);
                            const int val = (need && srcs[k] >= 0) ? val0 : D_INF;
                            if (need && srcs[k] >= 0 && val <= sw[p]) { tie = (val == sw[p]); sw[p] = val; choice = k + 1; }
                        }
                        if (__builtin_expect(__any(need && cc.w >= 0), 0)) {      // sources five to eight
                            int4 c2 = make_int4(-1, -1, -1, -1);
                            if (need && cc.w >= 0) c2 = cand2[p][lo[p] + c];
                            const int more[4] = {c2.x, c2.y, c2.z, c2.w};
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const int val0 = gather(pp, o, more[k] - olo);
                                const int val = more[k] >= 0 ? val0 : D_INF;
                                if (more[k] >= 0 && val <= sw[p]) { tie = (val == sw[p]); sw[p] = val; choice = k + 4; }
                            }
                        }
                        swbits[p] = f_choice_bits(choice) | (tie ? F_TIE : 0);
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
                    v[p] = b - c;
                }
                incl[0] = v[0]; incl[1] = v[1];
                wave_prefix_min2(incl[0], incl[1]);
            }
            // ---- publish the wave-local scan, one barrier, fold the lower waves' totals in
            rowbuf[par][0][c] = incl[0];
            rowbuf[par][1][c] = incl[1];
            if (lane == 63) { totals[par][0][wave] = incl[0]; totals[par][1][wave] = incl[1]; }
            lds_barrier<NW * 64>();
            int tq = (lane < NW) ? totals[par][0][lane & 15] : D_INF;
            int tr = (lane < NW) ? totals[par][1][lane & 15] : D_INF;
            row_prefix_min2(tq, tr);
            pin[0] = tq; pin[1] = tr;
            int carry[2];
            carry[0] = (wave > 0) ? __builtin_amdgcn_readlane(tq, (wave - 1) & 63) : D_INF;
            carry[1] = (wave > 0) ? __builtin_amdgcn_readlane(tr, (wave - 1) & 63) : D_INF;
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int excl_w = wave_shr1(incl[p], D_INF);            // wave-local exclusive scan
                const int excl = min(excl_w, carry[p]);
                const int Dn = min(incl[p], carry[p]) + c;               // inclusive prefix-min + x
                uint32_t f;
                if (t == 0) {   // row 0, dist.cpp:300-305,397-405
                    f = (c == 0) ? F_MAT : F_INS;
                } else {
                    f = (v[p] <= excl) ? mk[p] : 0;
                    const int left = excl + c - 1;                       // D of column c - 1
                    f |= (left + 1 == Dn) ? F_INS : 0;
                }
                if (st_ok[p]) fbuf[p][wave][rowo] = uint8_t(f);
                // exit test, see k_fwd_stripe
                exit_min = min(exit_min, exit_key(last ? ex_last[p] : ex_in[p], p, Dn, rhoc[p], vac[p], Lt - 1 - t, xend));
                Dp[p] = Dn;
                carry_prev[p] = carry[p];
            }
            rowo += 64;
        }
        // ---- flush the stripe's flag rows: rows x 64 bytes per wave and plane, 16 bytes per lane
        asm volatile("" ::: "memory");
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int row = lane >> 2, col = wave * 64 + (lane & 3) * 16;
            if (row < rows && col < d.pitch[p])
                *reinterpret_cast<uint4 *>(mat[p] + size_t(t0 + row) * d.pitch[p] + col) =
                    *reinterpret_cast<const uint4 *>(&fbuf[p][wave][lane * 16]);
        }
        asm volatile("" ::: "memory");
        // ---- advance to the next stripe
        plo[0] = lo[0]; plo[1] = lo[1];
        lo[0] = nlo[0]; lo[1] = nlo[1]; hi[0] = nhi[0]; hi[1] = nhi[1];
        kc[0] = kn[0]; kc[1] = kn[1];
        rhoc[0] = rhon[0]; rhoc[1] = rhon[1]; vac[0] = van[0]; vac[1] = van[1];
        if (((s + 1) & 63) == 0) {
            cbQ = nbQ; cbR = nbR;
            wide_origin<W>(t2r, tjp, r2q, s + 1 + 64 + lane, n_stripes, Lt, Lq, Lr, nbQ, nbR);
        }
    }
    // ---- end cells (row Lt-1 was computed with the last stripe's origins plo) and the window-wide exit minimum
    int em = exit_min, dummy = D_INF;
    wave_prefix_min2(em, dummy);
    if (c == 0) { red[NW] = D_INF; red[NW + 1] = D_INF; }
    if (lane == 63) red[wave] = em;
    lds_barrier<NW * 64>();
    if (c == Lq - 1 - plo[0]) red[NW] = Dp[0];
    if (c == Lr - 1 - plo[1]) red[NW + 1] = Dp[1];
    lds_barrier<NW * 64>();
    if (c == 0) {
        int m = D_INF;
        for (int w = 0; w < NW; w++) m = min(m, red[w]);
        outs[a].dist_q = red[NW];
        outs[a].dist_r = red[NW + 1];
        outs[a].exit_min = m;
    }
}

// ===========================================================================
// K2w: striped backward max-TP sweep for the wide windows, NW wavefronts per alignment (calc_prec_recall_path,
// dist.cpp:486-823); the multi-wave counterpart of k_bwd_stripe.  Thread i owns window column W-1-i (mirrored, so
// the suffix composition of the max-plus maps x -> max(A, x+B) that carries the in-row INS chain is a prefix scan in
// thread order).  One barrier per truth row: every wave publishes its local inclusive compositions (rowA/rowB) and
// its total map, after the barrier the totals of the lower waves are folded into the wave's carry = the final score
// of the column right of the wave.  Scores another wave needs from the previous row (swap successor, re-alignment
// at a stripe boundary) are reconstructed as apply(local composition, carry of that column's wave).  The forward
// flags of a stripe are staged in LDS (whole rows, prefetched one stripe ahead) and every "flags of a neighbour"
// read goes there; the path_ptr bytes are staged likewise and flushed per stripe, in place of the forward flags.
// ===========================================================================
template <int NW>
__global__ void __launch_bounds__(NW * 64) k_bwd_wide(DevBatch B, const AlnDesc *__restrict__ descs,
                                                      const int32_t *__restrict__ work, uint8_t *__restrict__ ws,
                                                      const int32_t *__restrict__ blo_all, AlnOut *__restrict__ outs, int tag) {
    constexpr int W = NW * 64;
    __builtin_amdgcn_s_setprio(2);
    const int a = work[blockIdx.x];
    const AlnDesc d = descs[a];
    if (outs[a].band_ok != tag || d.band_pad != tag) return;   // rejected by the exit test: re-run wider (uniform)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = W - 1 - tid;                   // window column of this thread
    const int Lq = d.Lq, Lr = d.Lr, Lt = d.Lt;
    const int Lp[2] = {Lq, Lr};
    const int32_t *bk[2] = {B.bk_q[d.qs] + d.q_off, B.bk_r[d.qs] + d.r_off};
    uint8_t *mat[2] = {ws + d.mat_off[0], ws + d.mat_off[1]};
    const int32_t *blo = blo_all + d.blo_off;
    const int end_plane = outs[a].end_plane;
    const int n_stripes = (Lt + WD_K - 1) / WD_K;
    const int pitch[2] = {d.pitch[0], d.pitch[1]};
    __shared__ __align__(16) uint8_t fin[2][2][WD_K * W];    // [buffer][plane] forward flags of a stripe, [row][pitch]
    __shared__ __align__(16) uint8_t fout[2][WD_K * W];      // [plane] path_ptr bytes of the current stripe
    __shared__ int2 rowM[2][2][W];                           // [row parity][plane][column] local inclusive compositions {A, B}: one 8-byte access
    __shared__ int4 totM[2][16];                             // [row parity][wave] the wave's total maps {A, B} of the two planes

    uint4 pfv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    auto stage_load = [&](int s_) {    // request stripe s_'s forward-flag rows (16 B per thread and plane)
        if (s_ < 0) return;
        const int ta = s_ * WD_K, nr = min(ta + WD_K, Lt) - ta;
#pragma unroll
        for (int p = 0; p < 2; p++)
            if (tid * 16 < nr * pitch[p])
                pfv[p] = *reinterpret_cast<const uint4 *>(mat[p] + size_t(ta) * pitch[p] + tid * 16);
    };
    auto stage_commit = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; p++)
            if (tid * 16 < WD_K * W) *reinterpret_cast<uint4 *>(&fin[buf][p][tid * 16]) = pfv[p];
    };
    auto origin_of = [&](int s_, int p) { return blo[p * Lt + s_ * WD_K]; };   // constant over the stripe's rows

    int lo[2], hi[2], plo[2] = {0, 0}, nlo[2] = {0, 0};
    int bkc[2], bkn[2], bkrc[2], bkrn[2];          // packed constants of this column / of column + 1
#pragma unroll
    for (int p = 0; p < 2; p++) {
        lo[p] = origin_of(n_stripes - 1, p);
        hi[p] = min(Lp[p] - 1, lo[p] + W - 1);
        bkc[p] = int(FK_NONE24); bkrc[p] = 0;
        if (lo[p] + col <= hi[p]) bkc[p] = bk[p][lo[p] + col];
        if (lo[p] + col + 1 <= hi[p]) bkrc[p] = bk[p][lo[p] + col + 1];
    }
    int sc1[2] = {S_NEG, S_NEG};     // final scores of row t+1 (own column)
    int f1[2] = {0, 0};              // forward flags of row t+1 (own column)
    int f1r[2] = {0, 0};             // ... and of the column right of it
    int pinA[2] = {S_NEG, S_NEG};    // lane l < NW: final score of the last column of wave l in the previous row
    int carry_prev[2] = {S_NEG, S_NEG};
    uint32_t tie_used = 0;
    stage_load(n_stripes - 1);
    stage_commit((n_stripes - 1) & 1);
    asm volatile("" ::"v"(bkc[0]), "v"(bkc[1]), "v"(bkrc[0]), "v"(bkrc[1]) : "memory");      // (no load in flight at the loop's entry: see the stripe's end)
    lds_barrier<NW * 64>();

    // final score of window column cz (of the origin row t+1 was computed with) in row t+1
    auto gather_s = [&](int par, int p, int cz) -> int {
        const bool ok = unsigned(cz) < unsigned(W);
        const int cc = ok ? cz : 0;
        const int2 AB = rowM[par][p][cc];
        const int A = AB.x, Bv = AB.y;
        const int wsrc = (W - 1 - cc) >> 6;
        const int car0 = __builtin_amdgcn_ds_bpermute(((wsrc - 1) & 63) << 2, pinA[p]);
        const int car = wsrc > 0 ? car0 : S_NEG;
        int v = max(A, (Bv >= 0) ? car + Bv : S_NEG);
        if (v < 0) v = S_NEG;
        return ok ? v : S_NEG;
    };

    for (int s = n_stripes - 1; s >= 0; s--) {
        const int t0 = s * WD_K, t1 = min(t0 + WD_K, Lt) - 1;
        const int cur = s & 1, above = cur ^ 1;
        if (s > 0) { nlo[0] = origin_of(s - 1, 0); nlo[1] = origin_of(s - 1, 1); }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            bkn[p] = int(FK_NONE24); bkrn[p] = 0;
            const int xn = nlo[p] + col, nh = min(Lp[p] - 1, nlo[p] + W - 1);
            if (s > 0 && xn <= nh) bkn[p] = bk[p][xn];
            if (s > 0 && xn + 1 <= nh) bkrn[p] = bk[p][xn + 1];
        }
        bool valid[2];
        int tp_right[2], zl[2];
        uint32_t zkey[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            valid[p] = lo[p] + col <= hi[p];
            tp_right[p] = (lo[p] + col + 1 <= hi[p]) ? ((bkrc[p] >> 24) & 1) : 0;
            zl[p] = bkc[p] & 0xffffff;   // swap target (absolute index in the other plane) or FK_NONE24
            zkey[p] = f_swp_key(rank_of(uint32_t(bkc[p]) >> 24));
        }
        stage_load(s - 1);                                     // prefetch the stripe below into registers
        // the swap target's column in the other plane's staged rows of this stripe, and this thread's own column
        int zc[2];
        bool zok[2], okc[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            zc[p] = zl[p] - lo[1 - p];
            zok[p] = zl[p] != int(FK_NONE24) && unsigned(zc[p]) < unsigned(W) && zc[p] < pitch[1 - p];
            okc[p] = valid[p] && col < pitch[p];
        }
        // forward flags of (plane p, absolute column x, row tt); tt == t+1 may lie in the stripe above
        auto fin_at = [&](int p, int x, int tt) -> int {
            if (tt >= Lt) return 0;
            const bool up_stripe = tt > t1;
            const int org = up_stripe ? plo[p] : lo[p];
            const int cx = x - org;
            const int rr = up_stripe ? 0 : tt - t0;
            const bool ok = (unsigned(cx) < unsigned(W)) & (cx < pitch[p]);
            const int byte = fin[up_stripe ? above : cur][p][ok ? rr * pitch[p] + cx : 0];
            return ok ? byte : 0;
        };

        for (int t = t1; t >= t0; t--) {
            const int par = t & 1, pp = par ^ 1;
            const bool first = (t == t1) && (s != n_stripes - 1);   // row t+1 is aligned to the stripe above
            int best[2], lk[2], f0[2], f0rr[2];
            uint32_t bm[2];
            MP g[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int o = 1 - p;
                const int x = lo[p] + col;
                int up_s, dn_s;
                if (first) {
                    up_s = gather_s(pp, p, x + 1 - plo[p]);
                    dn_s = gather_s(pp, p, x - plo[p]);
                } else {
                    const int ds_ = wave_shr1(sc1[p], S_NEG);
                    up_s = (lane == 0) ? carry_prev[p] : ds_;
                    dn_s = sc1[p];
                }
                // forward flags of (x+1, t+1) / (x, t+1): registers, except across a stripe boundary (new origin)
                const int up_f = first ? fin_at(p, x + 1, t + 1) : f1r[p];
                const int dn_f = first ? fin_at(p, x, t + 1) : f1[p];
                int b = S_NEG;
                uint32_t m = 0;
                if (f_diag(up_f)) { b = up_s + tp_right[p]; m = f_diag(up_f); }
                if (dn_f & F_DEL) {
                    if (dn_s > b) { b = dn_s; m = F_DEL; } else if (dn_s == b) m |= F_DEL;
                }
                // swap successor z = (other plane, zl, t+1)
                const bool hasz = zl[p] != int(FK_NONE24);
                // (inside a stripe the row below lies in the same staged rows at the same origin: no general lookup)
                const bool zin = zok[p] && t + 1 < Lt;
                const int zf = first ? (hasz ? fin_at(o, zl[p], t + 1) : 0)
                                     : int(fin[cur][o][zin ? (t + 1 - t0) * pitch[o] + zc[p] : 0]) & (zin ? 0xff : 0);
                const int zs = gather_s(pp, o, hasz ? zl[p] - (first ? plo[o] : lo[o]) : -1);
                if ((uint32_t(zf) & F_SWP_KEY_MASK) == zkey[p]) {
                    const int v = zs + ((bkc[p] >> 27) & 1);
                    if (v >= 0 && (zf & F_TIE)) tie_used++;
                    if (v > b) { b = v; m = F_SWP; } else if (v == b) m |= F_SWP;
                }
                if (t == Lt - 1 && p == end_plane && x == Lp[p] - 1) { b = 0; m = F_MAT; }   // dist.cpp:538-546
                if (!valid[p]) { b = S_NEG; m = 0; }
                best[p] = b;
                bm[p] = m;
                f0[p] = okc[p] ? int(fin[cur][p][(t - t0) * pitch[p] + col]) : 0;      // forward flags of (x, t)
                // forward flags of (x+1, t): the lane before (columns are mirrored) has just read them; the first lane of a wave
                // reads them itself.  (Behind the plane's last column both give "no INS link": a column that does not exist has no score)
                int f0r = wave_shr1(f0[p], 0);
                if (lane == 0) f0r = (x + 1 <= hi[p]) ? fin_at(p, x + 1, t) : 0;
                f0rr[p] = f0r;
                lk[p] = (f0r & F_INS) ? tp_right[p] : -1;
                g[p].A = b; g[p].B = lk[p];
            }
            MP hq = g[0], hr = g[1];
            wave_prefix_mp2(hq, hr);                                    // wave-local inclusive compositions
            rowM[par][0][col] = make_int2(hq.A, hq.B);
            rowM[par][1][col] = make_int2(hr.A, hr.B);
            if (lane == 63) totM[par][wave] = make_int4(hq.A, hq.B, hr.A, hr.B);
            lds_barrier<NW * 64>();
            MP tq, tr;
            const int4 tm = totM[par][lane & 15];
            tq.A = (lane < NW) ? tm.x : S_NEG; tq.B = (lane < NW) ? tm.y : -1;
            tr.A = (lane < NW) ? tm.z : S_NEG; tr.B = (lane < NW) ? tm.w : -1;
            row_prefix_mp2(tq, tr);
            pinA[0] = tq.A; pinA[1] = tr.A;
            int carry[2];
            carry[0] = (wave > 0) ? __builtin_amdgcn_readlane(tq.A, (wave - 1) & 63) : S_NEG;
            carry[1] = (wave > 0) ? __builtin_amdgcn_readlane(tr.A, (wave - 1) & 63) : S_NEG;
            const MP hl[2] = {hq, hr};
#pragma unroll
            for (int p = 0; p < 2; p++) {
                int vfin = max(hl[p].A, (hl[p].B >= 0) ? carry[p] + hl[p].B : S_NEG);
                if (vfin < 0) vfin = S_NEG;
                const int prevfin = wave_shr1(vfin, S_NEG);
                const int inc = (lane == 0) ? carry[p] : prevfin;       // final score of column x+1 in this row
                int v = best[p];
                uint32_t m = bm[p];
                if (lk[p] >= 0) {
                    const int w = inc + lk[p];
                    if (w > v) { v = w; m = F_INS; } else if (w == v) m |= F_INS;
                }
                if (v < 0) { v = S_NEG; m = 0; }
                sc1[p] = v;
                f1[p] = f0[p];
                f1r[p] = f0rr[p];
                carry_prev[p] = carry[p];
                if (col < pitch[p]) fout[p][(t - t0) * pitch[p] + col] = m ? uint8_t(m | (uint32_t(f0[p]) & F_KEEP)) : uint8_t(0);
            }
        }
        // ---- stripe end: flush the path_ptr rows in place of the forward flags, park the prefetched rows below
        lds_barrier<NW * 64>();
        stage_commit(above);     // the stripe below lands in the buffer the stripe above no longer needs (last read in this stripe's first row)
        // (every load of the stripe has landed before the stores go out, and the stores are issued without the compiler's knowledge:
        // see k_bwd_stripe -- the conservative waits cost this kernel three round trips per stripe of eight rows, 12.9 ms for a
        // 10 000-row alignment whose forward sweep takes 7)
        asm volatile("" ::"v"(bkn[0]), "v"(bkn[1]), "v"(bkrn[0]), "v"(bkrn[1]) : "memory");
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int nbytes = (t1 - t0 + 1) * pitch[p];
            if (tid * 16 < nbytes) lds_to_global16(&fout[p][tid * 16], mat[p] + size_t(t0) * pitch[p] + tid * 16);
        }
        lds_barrier<NW * 64>();
        plo[0] = lo[0]; plo[1] = lo[1];
        lo[0] = nlo[0]; lo[1] = nlo[1];
        hi[0] = min(Lq - 1, lo[0] + W - 1); hi[1] = min(Lr - 1, lo[1] + W - 1);
        bkc[0] = bkn[0]; bkc[1] = bkn[1];
        bkrc[0] = bkrn[0]; bkrc[1] = bkrn[1];
    }
    // (QUERY, 0, 0) is column 0 of row 0 (stripe 0 starts at the origin): the last thread
    if (tid == W - 1) outs[a].beg_plane = (sc1[0] >= 0) ? VPR_PLANE_QUERY : VPR_PLANE_REF;   // dist.cpp:811-814
    if (tie_used) { atomicOr(&outs[a].status, VPR_ST_SWAP_TIE); outs[a].band_ok = TIE_MARK(tag); atomicAdd(&outs[a].n_sec, int(tie_used)); }
}

#endif
