// pr_q16.hip -- the 16-cell window kernels: FOUR alignments per wavefront.
//
// On whole-genome small-variant input almost every supercluster alignment has distance 0..3 (truth and
// query agree), so the cells with D <= s hug the diagonal and a 64-cell window (pr_band.hip) spends most of
// its lanes on cells the reference never visits.  Here a DPP row (16 lanes) owns one alignment: lane gl of
// row g holds cell x = lo_p + gl of both planes, the window origin lo_p is constant over stripes of Q_K = 4
// truth rows, the in-row INS chain is a 4-step row_shr scan, and the four rows of the wave run four
// different alignments in lockstep (rows beyond an alignment's last truth row are masked).
//
// Same exactness contract as the wider windows: every cell with an edge leaving the window contributes
// D + (lower bound of what the diagonal offset still costs) to exit_min; the window is accepted only when
// exit_min > s, otherwise the alignment is re-run with the 64-cell kernels (then 256, 1024, dense).
//
// HBM layout of an alignment's flag bytes ("stripe-transposed"): record s = 128 bytes =
//   [plane][column 0..15][row 0..3 of stripe s], so a lane's four bytes of a stripe are ONE dword, a group's
// store / load of a stripe and plane is 64 contiguous bytes, and no LDS staging is needed.  The stripe
// origins are int2 {loQ, loR} per stripe at blo_off.
#ifndef PR_Q16_HIP_
#define PR_Q16_HIP_

#define Q_W 16
#define Q_K 4

// K0c: packed constants for these kernels.
//   which 0,1: fk4_q[which] (positions of query hap `which`)   {fk.x, fk.y, q2r[x], vs_hap[x-1]}
//   which 2,3: fk4_r[which-2] (ref positions, query hap h)     {fk.x, fk.y, x,      vs_ref[x-1]}
//   which 4,5: tk[which-4]    (positions of truth slot which-2) {t2r[t], base | fwd_allow(flag[t-1]) << 8 | vs[t-1] << 9}
__global__ void k_prep_q16(DevBatch B, int which, int64_t n_pos) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= n_pos) return;
    const int slot = which < 2 ? which : (which < 4 ? -1 : which - 2);
    const int64_t *off = slot >= 0 ? B.hap_off[slot] : B.ref_off;
    const int lo = (slot >= 0 ? B.sc_hap[slot] : B.sc_ref)[g];
    const int64_t beg = off[lo];
    const int64_t gm1 = g > beg ? g - 1 : beg;
    if (which < 2) {
        const int2 k = B.fk_q[which][g];
        B.fk4_q[which][g] = make_int4(k.x, k.y, B.hap_ptr[which][g], B.vs_hap[which][gm1]);
    } else if (which < 4) {
        const int h = which - 2;
        const int2 k = B.fk_r[h][g];
        B.fk4_r[h][g] = make_int4(k.x, k.y, int32_t(g - beg), B.vs_ref[h][gm1]);
    } else {
        const int f = g > beg ? int(B.hap_flag[slot][g - 1]) : 0;
        const int at = (!(f & PV) || (f & PE)) ? 1 : 0;
        B.tk[which - 4][g] = make_int2(B.hap_ptr[slot][g],
                                       int32_t(uint32_t(B.hap_seq[slot][g]) | (uint32_t(at) << 8) | (uint32_t(B.vs_hap[slot][gm1]) << 9)));
    }
}

// K0d: packed constants of the row-sweep walk (after k_prep_ins).  which 0,1: wk_q; 2,3: wk_r; 4,5: wk_t
__global__ void k_prep_wk(DevBatch B, int which, int64_t n_pos) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= n_pos) return;
    const int slot = which < 2 ? which : (which < 4 ? -1 : which - 2);
    const int lo = (slot >= 0 ? B.sc_hap[slot] : B.sc_ref)[g];
    const int64_t rbeg = B.ref_off[lo], rlen = B.ref_off[lo + 1] - rbeg;
    auto ins4 = [&](int32_t r) -> int {
        if (r < 0 || r >= rlen) return 0;
        const int64_t k = rbeg + r;
        return int(B.has_ins[0][k] != 0) | (int(B.has_ins[1][k] != 0) << 1) | (int(B.has_ins[2][k] != 0) << 2) |
               (int(B.has_ins[3][k] != 0) << 3);
    };
    if (slot >= 0) {
        const int32_t r = B.hap_ptr[slot][g];
        const int2 v = make_int2(r, int(B.hap_flag[slot][g]) | (ins4(r) << 8));
        if (which < 2) B.wk_q[which][g] = v; else B.wk_t[which - 4][g] = v;
    } else {
        B.wk_r[which - 2][g] = make_int2(B.ref_ptr[which - 2][g], ins4(int32_t(g - rbeg)) << 8);
    }
}

// value of lane `src` of this lane's 16-lane row (out-of-row sources give `fill`)
__device__ __forceinline__ int grp_get(int gbase, int src, int v, int fill) {
    const int r = __builtin_amdgcn_ds_bpermute((gbase | (src & 15)) << 2, v);
    return (unsigned(src) < 16u) ? r : fill;
}
__device__ __forceinline__ int row_shr1(int x, int fill) { return dpp_mov<0x111, 0xf>(fill, x); }   // lane i <- lane i-1 of the row

// two interleaved inclusive prefix-min scans inside every 16-lane row (see wave_prefix_min2)
__device__ __forceinline__ void row_prefix_min2(int &a, int &b) {
    asm volatile(
        "s_nop 1\n\t"
        "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_i32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_i32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_i32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b));
}

// two interleaved inclusive prefix compositions of max-plus maps inside every 16-lane row (see wave_prefix_mp2)
__device__ __forceinline__ void row_prefix_mp2(MP &a, MP &b) {
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
    asm volatile(
        "s_nop 1\n\t"
        MP_STEP("row_shr:1") MP_STEP("row_shr:2") MP_STEP("row_shr:4") MP_STEP("row_shr:8")
        "s_nop 0"
        : "+v"(Aq), "+v"(Bq), "+v"(Ar), "+v"(Br), "=&v"(tq), "=&v"(tr));
#undef MP_STEP
    a.A = (Aq >= MP_OFF) ? Aq - MP_OFF : S_NEG; a.B = (Bq < 0) ? -1 : Bq;
    b.A = (Ar >= MP_OFF) ? Ar - MP_OFF : S_NEG; b.B = (Br < 0) ? -1 : Br;
}

__device__ __forceinline__ int wave_max4(int v) {   // max of the four row leaders' values
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// ===========================================================================
// K1q: forward sweep, 16-cell window, four alignments per wave (calc_prec_recall_aln, dist.cpp:251-443)
// ===========================================================================
__global__ void __launch_bounds__(64) k_fwd_q16(DevBatch B, const AlnDesc *__restrict__ descs,
                                                const int32_t *__restrict__ work, int n_work,
                                                uint8_t *__restrict__ ws, int32_t *__restrict__ blo_all,
                                                AlnOut *__restrict__ outs, const int32_t *__restrict__ n_dev) {
    // n_dev: optional device-side length of a device-built work list (at most n_work entries of it are processed)
    if (n_dev) n_work = min(n_work, *n_dev);
    if (int(blockIdx.x) * 4 >= n_work) return;
    const int lane = threadIdx.x, gl = lane & 15, gbase = lane & 48;
    const int wi = int(blockIdx.x) * 4 + (lane >> 4);
    const int a_ = work[min(wi, n_work - 1)];
    const bool live = wi < n_work && a_ >= 0;      // (-1: padding of a device-built work list)
    const int a = max(a_, 0);
    const AlnDesc *dp = descs + a;
    const int Lq = dp->Lq, Lr = dp->Lr, Lt = live ? dp->Lt : 0;
    const int qs = dp->qs, ts = dp->ts;
    const int64_t q_off = dp->q_off, r_off = dp->r_off;
    const int Lp[2] = {Lq, Lr};
    const int2 *tk = (ts == 2 ? B.tk[0] : B.tk[1]) + dp->t_off;
    const uint16_t *tjp = (ts == 2 ? B.tj[0] : B.tj[1]) + dp->t_off;
    const int32_t *r2q = (qs == 0 ? B.ref_ptr[0] : B.ref_ptr[1]) + r_off;
    const int4 *fk[2] = {(qs == 0 ? B.fk4_q[0] : B.fk4_q[1]) + q_off, (qs == 0 ? B.fk4_r[0] : B.fk4_r[1]) + r_off};
    auto load_k = [&](int p, int idx) -> int4 { return fk[p][idx]; };
    const int4 *cand[2] = {(qs == 0 ? B.cand_q[0] : B.cand_q[1]) + q_off, (qs == 0 ? B.cand_r[0] : B.cand_r[1]) + r_off};
    const int4 *cand2[2] = {(qs == 0 ? B.cand2_q[0] : B.cand2_q[1]) + q_off, (qs == 0 ? B.cand2_r[0] : B.cand2_r[1]) + r_off};
    uint32_t *mat = reinterpret_cast<uint32_t *>(ws + dp->mat_off[0]);
    int2 *blo2 = reinterpret_cast<int2 *>(blo_all + dp->blo_off);
    const int nstr = (Lt + Q_K - 1) / Q_K;
    const int smax = wave_max4(nstr);

    // stripe origins: centred between the reference coordinates of the stripe's first and last truth row
    auto origin = [&](int s, int &oq, int &orr) {
        oq = 0; orr = 0;
        if (s > 0 && s < nstr) {
            const int ta = s * Q_K, tb = min(ta + Q_K - 1, Lt - 1);
            const int ra = tk[ta].x, rb = tk[tb].x;
            // (query_center, pr_band.hip: follow the query hap's insertion while the truth rows are inside one)
            const int ja = tjp[ta], jb = tjp[tb];
            const int rac = min(max(ra, 0), Lr - 1), rbc = min(max(rb, 0), Lr - 1);
            int qa = r2q[rac], qb = r2q[rbc];
            if (ja | jb) {      // rare
                if (ja) qa += min(ja, max((rac + 1 < Lr ? r2q[rac + 1] - qa - 1 : 0), 0));
                if (jb) qb += min(jb, max((rbc + 1 < Lr ? r2q[rbc + 1] - qb - 1 : 0), 0));
            }
            orr = max(0, min((ra + rb) / 2 - Q_W / 2, Lr - min(Q_W, Lr)));
            oq = max(0, min((qa + qb) / 2 - Q_W / 2, Lq - min(Q_W, Lq)));
        }
        if (s < nstr) blo2[s] = make_int2(oq, orr);    // read by K2 / K3
    };
    int cbQ, cbR, nbQ, nbR;                 // origins of stripes c0 + gl (this 16-stripe chunk / the next)
    origin(gl, cbQ, cbR);
    origin(16 + gl, nbQ, nbR);
    // truth constants of rows (t & ~15) + gl / the next 16 (clamped loads; rows past the end are masked)
    int2 tkc = tk[max(min(gl, Lt - 1), 0)], tkn = tk[max(min(16 + gl, Lt - 1), 0)];

    int exit_min = D_INF;
    int Dp[2] = {gl, gl};                   // row 0: D = x along the INS chain (origin 0)
    int lo[2] = {0, 0}, hi[2] = {min(Lq, Q_W) - 1, min(Lr, Q_W) - 1};
    int dlo[2] = {0, 0};                    // origins the D registers are aligned to
    int nlo[2] = {0, 0}, nhi[2] = {0, 0};
    int4 kc[2], kn[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        kc[p] = load_k(p, min(gl, Lp[p] - 1));
    }

    for (int s = 0; s < smax; s++) {
        const bool act = s < nstr;
        const bool has_next = s + 1 < nstr;
        // ---- next stripe's window, prefetch of its constants
        if (((s + 1) & 15) == 0) { nlo[0] = grp_get(gbase, 0, nbQ, 0); nlo[1] = grp_get(gbase, 0, nbR, 0); }
        else { nlo[0] = grp_get(gbase, (s + 1) & 15, cbQ, 0); nlo[1] = grp_get(gbase, (s + 1) & 15, cbR, 0); }
        if (!has_next) { nlo[0] = lo[0]; nlo[1] = lo[1]; }
        nhi[0] = min(Lq - 1, nlo[0] + Q_W - 1);
        nhi[1] = min(Lr - 1, nlo[1] + Q_W - 1);
        // (unconditional, clamped loads: lanes outside the window / rows past the end are masked where used)
#pragma unroll
        for (int p = 0; p < 2; p++) kn[p] = load_k(p, min(nlo[p] + gl, Lp[p] - 1));
        if ((s & 3) == 0 && s > 0) {
            tkc = tkn;
            tkn = tk[max(min(s * Q_K + 16 + gl, Lt - 1), 0)];
        }
        // ---- per-lane constants of this stripe
        int s0[2];
        uint32_t base[2];
        bool multi[2], ex_in[2], ex_last[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int o = 1 - p;
            const int x = lo[p] + gl;
            const bool valid = act & (x <= hi[p]);
            s0[p] = (kc[p].x < 0) ? -1 : (kc[p].x & (FK_MULTI - 1));
            multi[p] = (kc[p].x >= 0) & ((kc[p].x & FK_MULTI) != 0);
            base[p] = valid ? (uint32_t(kc[p].y) >> 24) : 0xffu;
            // branch-free edge tests (bitwise on purpose: && / || make hipcc emit exec-mask branches)
            const int z = kc[p].y & 0xffffff;                       // FK_NONE24 >= any string length
            const bool zok = valid & (z < Lp[o]);
            const bool ins_out = valid & (x == hi[p]) & (hi[p] < Lp[p] - 1);
            const bool z_out = unsigned(z - lo[o]) > unsigned(hi[o] - lo[o]);
            const bool z_out_n = unsigned(z - nlo[o]) > unsigned(nhi[o] - nlo[o]);
            const bool x_out_n = (x < nlo[p]) | ((x + 1 < Lp[p]) & (x + 1 > nhi[p]));
            ex_in[p] = ins_out | (zok & z_out);
            ex_last[p] = ins_out | (has_next & ((valid & x_out_n) | (zok & z_out_n)));
        }
        uint32_t facc[2] = {0, 0};
        const int trow = (gbase | ((s & 3) * Q_K)) << 2;     // bpermute address of this stripe's first row constants

#pragma unroll 1
        for (int r = 0; r < Q_K; r++) {
            const int t = s * Q_K + r;
            const bool ract = t < Lt;
            const bool last = (r == Q_K - 1) || (t == Lt - 1);
            const int tau = __builtin_amdgcn_ds_bpermute(trow + 4 * r, tkc.x);
            const int tky = __builtin_amdgcn_ds_bpermute(trow + 4 * r, tkc.y);
            const int vt = int(uint32_t(tky) >> 9);
            if (r == 0 && s == 0) {   // row 0, dist.cpp:300-305,397-405
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const bool valid = gl <= hi[p];
                    const bool ex = last ? ex_last[p] : ex_in[p];
                    facc[p] = valid ? ((gl == 0) ? F_MAT : F_INS) : 0;
                    const int off0_ = kc[p].z - tau;
                    const int lb0 = max((off0_ < 0 ? -off0_ : off0_) - kc[p].w - vt, 0);
                    exit_min = (ex && ract) ? min(exit_min, gl + lb0) : exit_min;
                }
                continue;
            }
            const uint32_t Tt = uint32_t(tky) & 0xff;
            const bool at = (tky >> 8) & 1;
            const bool first = (r == 0);   // the previous row belongs to the previous stripe (origins dlo)

            int v[2], up[2], dg[2], sw[2];
            uint32_t mk[2];
            bool match[2], need_multi = false;
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int o = 1 - p;
                if (first) {
                    const int sh = lo[p] - dlo[p];
                    up[p] = grp_get(gbase, gl + sh, Dp[p], D_INF);
                    dg[p] = grp_get(gbase, gl + sh - 1, Dp[p], D_INF);
                } else {
                    up[p] = Dp[p];
                    dg[p] = row_shr1(Dp[p], D_INF);
                }
                match[p] = base[p] == Tt;
                const bool on = match[p] && at && s0[p] >= 0;
                const int sv = grp_get(gbase, s0[p] - (first ? dlo[o] : lo[o]), Dp[o], D_INF);
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
                    if (need) cc = cand[p][lo[p] + gl];
                    const int olo = first ? dlo[o] : lo[o];
                    const int srcs[3] = {cc.y, cc.z, cc.w};
                    int choice = 0;
                    bool tie = false;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const int val0 = grp_get(gbase, srcs[k] - olo, Dp[o], D_INF);
                        const int val = (need && srcs[k] >= 0) ? val0 : D_INF;
                        if (need && srcs[k] >= 0 && val <= sw[p]) { tie = (val == sw[p]); sw[p] = val; choice = k + 1; }
                    }
                    if (__builtin_expect(__any(need && cc.w >= 0), 0)) {      // sources five to eight
                        int4 c2 = make_int4(-1, -1, -1, -1);
                        if (need && cc.w >= 0) c2 = cand2[p][lo[p] + gl];
                        const int more[4] = {c2.x, c2.y, c2.z, c2.w};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int val = grp_get(gbase, more[k] - olo, Dp[o], D_INF);
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
                v[p] = b - gl;
            }
            int iq = v[0], ir = v[1];
            row_prefix_min2(iq, ir);
            const int inc[2] = {iq, ir};
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int carry = row_shr1(inc[p], D_INF);
                const int Dn = inc[p] + gl;                          // inclusive prefix-min + x
                uint32_t f = (v[p] <= carry) ? mk[p] : 0;
                const int left = row_shr1(Dn, D_INF);
                f |= (left + 1 == Dn) ? F_INS : 0;
                const bool valid = lo[p] + gl <= hi[p];
                facc[p] |= (ract && valid) ? (f << (8 * r)) : 0;
                // exit test, see k_fwd_stripe
                const bool ex = last ? ex_last[p] : ex_in[p];
                const int doff = kc[p].z - tau;
                const int lb = max((doff < 0 ? -doff : doff) - kc[p].w - vt, 0);
                exit_min = (ex && ract) ? min(exit_min, Dn + lb) : exit_min;
                Dp[p] = ract ? Dn : Dp[p];
            }
            if (first) { dlo[0] = act ? lo[0] : dlo[0]; dlo[1] = act ? lo[1] : dlo[1]; }
        }
        // ---- the stripe's flag bytes: one dword per lane and plane, 64 contiguous bytes per group
        if (act) {
            mat[s * 32 + gl] = facc[0];
            mat[s * 32 + 16 + gl] = facc[1];
        }
        // ---- advance to the next stripe
        if (has_next) {
            lo[0] = nlo[0]; lo[1] = nlo[1]; hi[0] = nhi[0]; hi[1] = nhi[1];
            kc[0] = kn[0]; kc[1] = kn[1];
        }
        if (((s + 1) & 15) == 0) {
            cbQ = nbQ; cbR = nbR;
            origin(s + 1 + 16 + gl, nbQ, nbR);
        }
    }
    // end cells: the D registers are aligned to the last stripe's origins
    const int dq = grp_get(gbase, Lq - 1 - dlo[0], Dp[0], D_INF);
    const int dr = grp_get(gbase, Lr - 1 - dlo[1], Dp[1], D_INF);
    int em = exit_min, dummy = D_INF;
    row_prefix_min2(em, dummy);
    if (live && gl == 15) {
        outs[a].dist_q = dq;
        outs[a].dist_r = dr;
        outs[a].exit_min = em;
    }
}

// ===========================================================================
// K2q: backward max-TP sweep over the 16-cell layout, four alignments per wave (calc_prec_recall_path,
// dist.cpp:486-823).  Lanes are mirrored inside the row (lane gl owns column 15 - gl) so "x+1" is lane gl-1
// and the suffix composition of the max-plus maps is a prefix scan in lane order.  The forward flags of a
// stripe are one dword per lane and plane, replaced in place by the path_ptr bytes.
// ===========================================================================
// tag: the level tag the accept test stored in band_ok (see k_fwd_band_finish).
__global__ void __launch_bounds__(64) k_bwd_q16(DevBatch B, const AlnDesc *__restrict__ descs,
                                                const int32_t *__restrict__ work, int n_work,
                                                uint8_t *__restrict__ ws, const int32_t *__restrict__ blo_all,
                                                AlnOut *__restrict__ outs, int tag, int dtag,
                                                const int32_t *__restrict__ n_dev) {
    if (n_dev) n_work = min(n_work, *n_dev);
    if (int(blockIdx.x) * 4 >= n_work) return;
    const int lane = threadIdx.x, gl = lane & 15, gbase = lane & 48;
    const int wi = int(blockIdx.x) * 4 + (lane >> 4);
    const int a_ = work[min(wi, n_work - 1)];
    const bool live = wi < n_work && a_ >= 0;      // (-1: padding of a device-built work list)
    const int a = max(a_, 0);
    const AlnDesc *dp = descs + a;
    // tag: what this round's accept test stores in band_ok; dtag: level tag of the descriptors it runs on (they
    // differ when the general 16-cell round re-runs, in place, what the zero-distance round rejected)
    const bool on = live && outs[a].band_ok == tag && dp->band_pad == dtag;
    const int Lq = dp->Lq, Lr = dp->Lr, Lt = on ? dp->Lt : 0;
    const int qs = dp->qs;
    const int Lp[2] = {Lq, Lr};
    const int32_t *bk[2] = {(qs == 0 ? B.bk_q[0] : B.bk_q[1]) + dp->q_off, (qs == 0 ? B.bk_r[0] : B.bk_r[1]) + dp->r_off};
    uint32_t *mat = reinterpret_cast<uint32_t *>(ws + dp->mat_off[0]);
    const int2 *blo2 = reinterpret_cast<const int2 *>(blo_all + dp->blo_off);
    const int end_plane = outs[a].end_plane;
    const int nstr = (Lt + Q_K - 1) / Q_K;
    const int smax = wave_max4(nstr);
    if (smax == 0) return;
    const int col = 15 - gl;                       // window column of this lane

    auto load_chunk = [&](int c0, int &bq, int &br) {
        bq = 0; br = 0;
        const int s = c0 + gl;
        if (c0 >= 0 && s < nstr) { const int2 o = blo2[s]; bq = o.x; br = o.y; }
    };
    int cbQ, cbR, lbQ, lbR;                        // origins of stripes c0 + gl (this chunk / the one below)
    const int c_top = (smax - 1) & ~15;
    load_chunk(c_top, cbQ, cbR);
    load_chunk(c_top - 16, lbQ, lbR);

    int lo[2], plo[2] = {0, 0};
    int bkc[2], bkn[2];
    uint32_t fw[2], fwn[2];
    lo[0] = grp_get(gbase, (smax - 1) & 15, cbQ, 0);
    lo[1] = grp_get(gbase, (smax - 1) & 15, cbR, 0);
#pragma unroll
    for (int p = 0; p < 2; p++) {
        bkc[p] = int(FK_NONE24);
        fw[p] = 0;
        if (smax - 1 < nstr) {
            if (lo[p] + col <= min(Lp[p] - 1, lo[p] + Q_W - 1)) bkc[p] = bk[p][lo[p] + col];
            fw[p] = mat[(smax - 1) * 32 + p * 16 + col];
        }
    }
    int sc1[2] = {S_NEG, S_NEG};     // scores of row t+1 (own column)
    int f1[2] = {0, 0};              // forward flags of row t+1
    uint32_t tie_used = 0;

    for (int s = smax - 1; s >= 0; s--) {
        const bool act = s < nstr;
        // ---- the stripe below: origins, constants and forward flags (prefetched)
        int nlo[2] = {0, 0};
        if (s > 0) {
            if ((s & 15) == 0) { nlo[0] = grp_get(gbase, 15, lbQ, 0); nlo[1] = grp_get(gbase, 15, lbR, 0); }
            else { nlo[0] = grp_get(gbase, (s - 1) & 15, cbQ, 0); nlo[1] = grp_get(gbase, (s - 1) & 15, cbR, 0); }
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            bkn[p] = int(FK_NONE24);
            fwn[p] = 0;
            if (s > 0 && s - 1 < nstr) {
                const int xn = nlo[p] + col;
                if (xn <= min(Lp[p] - 1, nlo[p] + Q_W - 1)) bkn[p] = bk[p][xn];
                fwn[p] = mat[(s - 1) * 32 + p * 16 + col];
            }
        }
        // ---- per-lane constants of this stripe
        bool valid[2];
        int tp_own[2], tp_right[2], zl[2], sh[2];
        uint32_t zkey[2];
        uint32_t fwr[2], oacc[2] = {0, 0};
#pragma unroll
        for (int p = 0; p < 2; p++) {
            valid[p] = act && lo[p] + col <= min(Lp[p] - 1, lo[p] + Q_W - 1);
            tp_own[p] = (bkc[p] >> 24) & 1;
            tp_right[p] = row_shr1(tp_own[p], 0);
            zl[p] = bkc[p] & 0xffffff;   // swap target (absolute index in the other plane) or FK_NONE24
            zkey[p] = f_swp_key(rank_of(uint32_t(bkc[p]) >> 24));
            fw[p] = valid[p] ? fw[p] : 0;
            fwr[p] = uint32_t(row_shr1(int(fw[p]), 0));          // forward flags of column x+1
            sh[p] = (s == nstr - 1) ? 0 : plo[p] - lo[p];        // origin shift against the stripe above
        }

#pragma unroll
        for (int r = Q_K - 1; r >= 0; r--) {
            const int t = s * Q_K + r;
            const bool ract = t < Lt;
            const bool first = (r == Q_K - 1);   // row t+1 is aligned to the stripe above
            int best[2], lk[2], f0[2];
            uint32_t bm[2];
            MP g[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int o = 1 - p;
                int up_s, up_f, dn_s, dn_f;
                if (first) {
                    up_s = grp_get(gbase, gl + sh[p] - 1, sc1[p], S_NEG);
                    up_f = grp_get(gbase, gl + sh[p] - 1, f1[p], 0);
                    dn_s = grp_get(gbase, gl + sh[p], sc1[p], S_NEG);
                    dn_f = grp_get(gbase, gl + sh[p], f1[p], 0);
                } else {
                    up_s = row_shr1(sc1[p], S_NEG);
                    up_f = row_shr1(f1[p], 0);
                    dn_s = sc1[p];
                    dn_f = f1[p];
                }
                int b = S_NEG;
                uint32_t m = 0;
                const uint32_t dgm = f_diag(up_f);
                if (dgm) { b = up_s + tp_right[p]; m = dgm; }
                if (dn_f & F_DEL) {
                    if (dn_s > b) { b = dn_s; m = F_DEL; } else if (dn_s == b) m |= F_DEL;
                }
                // swap successor z = (other plane, zl, t+1): its lane in the alignment of row t+1
                const int olo = (first && s != nstr - 1) ? plo[o] : lo[o];
                const int zsrc = (zl[p] == int(FK_NONE24)) ? -1 : 15 - (zl[p] - olo);
                const int zf = grp_get(gbase, zsrc, f1[o], 0);
                const int zs = grp_get(gbase, zsrc, sc1[o], S_NEG);
                if ((uint32_t(zf) & F_SWP_KEY_MASK) == zkey[p]) {
                    const int v = zs + ((bkc[p] >> 27) & 1);
                    if (v >= 0 && (zf & F_TIE)) tie_used++;
                    if (v > b) { b = v; m = F_SWP; } else if (v == b) m |= F_SWP;
                }
                if (t == Lt - 1 && p == end_plane && lo[p] + col == Lp[p] - 1) { b = 0; m = F_MAT; }   // dist.cpp:538-546
                if (!valid[p] || !ract) { b = S_NEG; m = 0; }
                best[p] = b;
                bm[p] = m;
                f0[p] = int((fw[p] >> (8 * r)) & 0xff);           // forward flags of (x, t)
                const int f0r = int((fwr[p] >> (8 * r)) & 0xff);  // forward flags of (x+1, t)
                lk[p] = (f0r & F_INS) ? tp_right[p] : -1;
                g[p].A = b; g[p].B = lk[p];
            }
            MP hq = g[0], hr = g[1];
            row_prefix_mp2(hq, hr);
            const int inc[2] = {row_shr1(hq.A, S_NEG), row_shr1(hr.A, S_NEG)};
#pragma unroll
            for (int p = 0; p < 2; p++) {
                int v = best[p];
                uint32_t m = bm[p];
                if (lk[p] >= 0) {
                    const int w = inc[p] + lk[p];
                    if (w > v) { v = w; m = F_INS; } else if (w == v) m |= F_INS;
                }
                if (v < 0 || !valid[p]) { v = S_NEG; m = 0; }
                if (ract) {
                    sc1[p] = v;          // becomes the "row t+1" score of the next iteration
                    f1[p] = f0[p];       // ... and its forward flags
                    oacc[p] |= (m ? (m | (uint32_t(f0[p]) & F_KEEP)) : 0u) << (8 * r);
                }
            }
        }
        // ---- this stripe's path_ptr bytes replace the forward flags
        if (act) {
            mat[s * 32 + col] = oacc[0];
            mat[s * 32 + 16 + col] = oacc[1];
        }
        // ---- advance to the stripe below
        if (act) { plo[0] = lo[0]; plo[1] = lo[1]; }
        lo[0] = nlo[0]; lo[1] = nlo[1];
        bkc[0] = bkn[0]; bkc[1] = bkn[1];
        fw[0] = fwn[0]; fw[1] = fwn[1];
        if ((s & 15) == 0 && s > 0) {
            cbQ = lbQ; cbR = lbR;
            load_chunk(((s - 1) & ~15) - 16, lbQ, lbR);
        }
    }
    // (QUERY, 0, 0) is column 0 of row 0 = lane 15 of the row (stripe 0 starts at the origin)
    if (on && gl == 15) outs[a].beg_plane = (sc1[0] >= 0) ? VPR_PLANE_QUERY : VPR_PLANE_REF;   // dist.cpp:811-814
    if (on && tie_used) { atomicOr(&outs[a].status, VPR_ST_SWAP_TIE); outs[a].band_ok = TIE_MARK(tag); atomicAdd(&outs[a].n_sec, int(tie_used)); }
}

// ===========================================================================
// K3q: the forward walk of get_prec_recall_path_sync (dist.cpp:865-998) as a sweep over truth rows for the
// 16-cell layout, four alignments per wave (see k_walk_rows for the 64-cell version and the argument):
// inside one row the walk takes INS moves exactly while INS is the highest-priority move left in the cell's
// path_ptr byte, so the cells visited in row t are the entry cell plus the run of "INS-only" cells behind
// it - a ballot over the row, a count-trailing-ones, the lanes of the run store their path entries side by
// side, and one cross-lane read of the run's last cell decides the move into row t+1.
// ===========================================================================
__global__ void __launch_bounds__(64) k_walk_q16(DevBatch B, const AlnDesc *__restrict__ descs,
                                                 const int32_t *__restrict__ work, int n_work,
                                                 const uint8_t *__restrict__ ws, const int32_t *__restrict__ blo_all,
                                                 AlnOut *__restrict__ outs, PathEnt *__restrict__ paths, int tag, int dtag,
                                                 const int32_t *__restrict__ n_dev) {
    if (n_dev) n_work = min(n_work, *n_dev);
    if (int(blockIdx.x) * 4 >= n_work) return;
    const int lane = threadIdx.x, gl = lane & 15, gbase = lane & 48;
    const int wi = int(blockIdx.x) * 4 + (lane >> 4);
    const int a_ = work[min(wi, n_work - 1)];
    const bool live = wi < n_work && a_ >= 0;      // (-1: padding of a device-built work list)
    const int a = max(a_, 0);
    const AlnDesc *dp = descs + a;
    const bool on = live && outs[a].band_ok == tag && dp->band_pad == dtag;
    const int Lq = dp->Lq, Lr = dp->Lr, Lt = on ? dp->Lt : 0;
    const int qs = dp->qs, ts = dp->ts;
    const int path_cap = dp->path_cap;
    const int2 *wq = (qs == 0 ? B.wk_q[0] : B.wk_q[1]) + dp->q_off;
    const int2 *wr = (qs == 0 ? B.wk_r[0] : B.wk_r[1]) + dp->r_off;
    const int2 *wt = (ts == 2 ? B.wk_t[0] : B.wk_t[1]) + dp->t_off;
    const int insmask = ((1 << qs) | (1 << ts)) << 8;
    const uint32_t *mat = reinterpret_cast<const uint32_t *>(ws + dp->mat_off[0]);
    const int2 *blo2 = reinterpret_cast<const int2 *>(blo_all + dp->blo_off);
    PathEnt *path = paths + dp->path_off;
    const int nstr = (Lt + Q_K - 1) / Q_K;
    const int smax = wave_max4(nstr);
    if (smax == 0) return;

    auto load_chunk = [&](int c0, int &bq, int &br) {   // origins of stripes c0 + gl
        bq = 0; br = 0;
        if (c0 + gl < nstr) { const int2 o = blo2[c0 + gl]; bq = o.x; br = o.y; }
    };
    int cbQ, cbR, nbQ, nbR;
    load_chunk(0, cbQ, cbR);
    load_chunk(16, nbQ, nbR);
    int2 wtc = make_int2(0, 0), wtn = make_int2(0, 0);   // truth-row constants of rows (t & ~15) + gl / the next 16
    if (gl < Lt) wtc = wt[gl];
    if (16 + gl < Lt) wtn = wt[16 + gl];
    // per-stripe data of this lane's column (prefetched one stripe ahead): path_ptr dwords and column constants
    int lo[2] = {0, 0};
    uint32_t pp[2] = {0, 0}, ppn[2] = {0, 0};
    int2 cq = make_int2(0, 0), cr = make_int2(0, 0), cqn, crn;
    if (nstr > 0) {
        pp[0] = mat[gl]; pp[1] = mat[16 + gl];
        if (gl < Lq) cq = wq[gl];
        if (gl < Lr) cr = wr[gl];
    }

    int hi = outs[a].beg_plane, e = 0, n = 0, mv_in = 0;   // plane / column of the entry cell of the current row
    uint32_t edit_in = 0, status = 0;
    bool ok = on;

    for (int s = 0; s < smax; s++) {
        // ---- prefetch the next stripe
        int nlo[2];
        if (((s + 1) & 15) == 0) { nlo[0] = grp_get(gbase, 0, nbQ, 0); nlo[1] = grp_get(gbase, 0, nbR, 0); }
        else { nlo[0] = grp_get(gbase, (s + 1) & 15, cbQ, 0); nlo[1] = grp_get(gbase, (s + 1) & 15, cbR, 0); }
        ppn[0] = 0; ppn[1] = 0; cqn = make_int2(0, 0); crn = make_int2(0, 0);
        if (s + 1 < nstr) {
            ppn[0] = mat[(s + 1) * 32 + gl]; ppn[1] = mat[(s + 1) * 32 + 16 + gl];
            if (nlo[0] + gl < Lq) cqn = wq[nlo[0] + gl];
            if (nlo[1] + gl < Lr) crn = wr[nlo[1] + gl];
        }
        if ((s & 3) == 0 && s > 0) {
            wtc = wtn;
            wtn = make_int2(0, 0);
            const int tt = s * Q_K + 16 + gl;
            if (tt < Lt) wtn = wt[tt];
        }
        const int trow = (gbase | ((s & 3) * Q_K)) << 2;

#pragma unroll
        for (int r = 0; r < Q_K; r++) {
            const int t = s * Q_K + r;
            bool ract = ok && t < Lt;
            int trv, twy, ex, ey;
            const int lo_h = hi ? lo[1] : lo[0];
            const int Lh = hi ? Lr : Lq;
            const int el = e - lo_h;
            if (ract && (el < 0 || el > 15 || e >= Lh)) { status |= VPR_ST_ERR_NO_PTR; ok = false; ract = false; }
            const int colx = hi ? cr.x : cq.x;          // r2q / q2r of this lane's column in the walk's plane
            {
                trv = __builtin_amdgcn_ds_bpermute(trow + 4 * r, wtc.x);
                twy = __builtin_amdgcn_ds_bpermute(trow + 4 * r, wtc.y);
                const int coly = hi ? cr.y : cq.y;          // flags | ins4 << 8 (REF plane: no flags)
                ex = grp_get(gbase, el, colx, 0);
                ey = grp_get(gbase, el, coly, 0);
            }
            // sync flag of the entry cell, dist.cpp:949-968 (only a diagonal move can make a sync point)
            uint32_t sync_in = 1;
            if (mv_in != 0) {
                const int qr = hi ? e : ex;
                const bool in_t = (twy & PV) && !(twy & PB);
                const bool in_q = (ey & PV) && !(ey & PB);
                sync_in = (!in_t && !in_q && !((twy | ey) & insmask) && trv == qr && (mv_in & (F_MAT | F_SWP | F_SUB))) ? 1u : 0u;
            }
            // the run of INS-only cells that starts at the entry cell
            const int pc = int(((hi ? pp[1] : pp[0]) >> (8 * r)) & 0xff);
            int k = 0;
            {
                const bool ins_only = (pc & F_INS) && !(pc & (F_MAT | F_SUB)) && !(hi == 1 && (pc & F_SWP));
                const unsigned long long bal = __ballot(ins_only);
                const uint32_t m16 = (uint32_t(bal >> gbase) & 0xffffu) >> (el & 15);
                k = __builtin_ctz(~m16);                         // bits 16.. of ~m16 are ones: k <= 16
            }
            const int c = e + k;                                 // last cell visited in this row
            if (ract && (c - lo_h > 15 || c >= Lh)) { status |= VPR_ST_ERR_NO_PTR; ok = false; ract = false; }
            if (ract && n + k + 1 > path_cap) { status |= VPR_ST_ERR_LIMIT; ok = false; ract = false; }
            if (ract && gl >= el && gl <= el + k) {
                const int x = lo_h + gl;
                PathEnt pe;
                pe.a = uint32_t(x) | (uint32_t(hi) << 31);
                // bits 28 / 29 (credit_walk, pr_kernels.hip): "the reference base here equals the truth base" / "that bit is valid" --
                // a cell entered by MAT or SWP has the truth row's base, by SUB another one, and its base is the reference's on
                // the REF plane and outside variants on the QUERY plane
                const uint32_t eqb = ((mv_in & (F_MAT | F_SWP | F_SUB)) && (hi == 1 || !(ey & PV))) ? (2u | ((mv_in & F_SUB) ? 0u : 1u)) : 0u;
                pe.b = uint32_t(t) | ((gl == el) ? ((sync_in << 31) | (edit_in << 30) | (eqb << 28)) : (1u << 30));
                pe.qref = hi ? x : colx;
                pe.tref = trv;
                path[n + (gl - el)] = pe;
            }
            // move out of the row from cell c, by priority
            const int cl = c - lo_h;
            const int p = grp_get(gbase, cl, pc, 0) & 31;
            const int cxv = grp_get(gbase, cl, colx, 0);
            if (ract) {
                n += k + 1;
                if (t == Lt - 1) {                               // the walk ends at the end cell of its plane
                    if (c != Lh - 1) { status |= VPR_ST_ERR_NO_PTR; ok = false; }
                } else if (hi == 1 && (p & F_SWP)) { mv_in = F_SWP; e = cxv + 1; hi = 0; edit_in = 0; }
                else if (p & F_MAT) { mv_in = F_MAT; e = c + 1; edit_in = 0; }
                else if (p & F_SUB) { mv_in = F_SUB; e = c + 1; edit_in = 1; }
                else if (p & F_DEL) { mv_in = F_DEL; e = c; edit_in = 1; }
                else if (hi == 0 && (p & F_SWP)) { mv_in = F_SWP; e = cxv + 1; hi = 1; edit_in = 0; }
                else { status |= VPR_ST_ERR_NO_PTR; ok = false; }
            }
        }
        // ---- advance to the next stripe
        lo[0] = nlo[0]; lo[1] = nlo[1];
        pp[0] = ppn[0]; pp[1] = ppn[1];
        cq = cqn; cr = crn;
        if (((s + 1) & 15) == 0) {
            cbQ = nbQ; cbR = nbR;
            load_chunk(s + 1 + 16, nbQ, nbR);
        }
    }
    if (on && gl == 0) {
        outs[a].path_len = n;
        if (!ok) outs[a].n_sec = 0;
        if (status) atomicOr(&outs[a].status, status);
    }
}

#endif
