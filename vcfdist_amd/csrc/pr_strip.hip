// pr_strip.hip -- the dense level for WIDE alignments: one alignment spread over several workgroups (compute units).
//
// A dense sweep is sequential over the truth rows and, inside a workgroup, bound by one compute unit's issue rate: a
// 12 000 x 24 000 alignment with a one-sided SV (distance in the thousands, so no window level accepts it) took 0.2 s
// forward and 0.7 s backward in one 1024-thread workgroup whose per-thread state no longer fits the register file.  Here
// the columns are cut into STRIPS of at most ST_NT * C cells per plane, one workgroup each, and the strips of an
// alignment run as a pipeline over the rows:
//   * every edge of the two-plane graph leads to the same or a larger column (MAT / SUB / INS: +1, DEL: same, swap:
//     pointer + 1 in the other plane, and the pointers are monotone), so strip j only needs values from strip j - 1
//     (forward sweep) or strip j + 1 (backward sweep: the edges reversed);
//   * the cuts are placed where the planes map 1:1 onto each other (k_strip_plan), so exactly ONE cell per plane and row
//     crosses a cut: the neighbour's last (first) column.  It is handled as a GHOST cell of the consuming strip -- a cell
//     whose value is given -- and everything else (diagonal move, INS chain, swap sources, the backward max-plus chain)
//     is the single-workgroup code unchanged;
//   * a strip publishes its boundary column in blocks of 64 rows (values staged in LDS, one coalesced store, a release
//     of its progress counter); the consumer acquires the counter once per 64 rows.  A workgroup's strip comes with a
//     ticket drawn at entry, from a list that holds every alignment's strip j - 1 before its strip j (strip-major: the host,
//     strip_plan in pr_api.hip), and a strip only ever waits for the strip before its own, so the wait cannot deadlock.
// Flag matrices, results and everything downstream (walk, credit) are those of k_fwd / k_bwd (pr_kernels.hip): dist.cpp
// :251-443 (forward), :486-823 (backward) -- with one difference: the backward sweep rewrites the flag bytes of the cells it
// SCORES (the cells on optimal paths) and leaves the others as the forward sweep wrote them; the walk only ever reads scored
// cells, and the one reader of whole workspaces, an early tie replay, is kept away from strip workspaces (tie_round).
#ifndef PR_STRIP_HIP_
#define PR_STRIP_HIP_

#define ST_NT 1024
#define ST_C 4
#define ST_CAP (ST_NT * ST_C - 8)     // columns of one plane a strip may hold (ghost + alignment slack taken off)

struct StripTab { int32_t lo[2], hi[2]; };      // columns [lo[p], hi[p]) of plane p (0 QUERY, 1 REF)

// Cuts of every alignment of a launch.  One thread per alignment: from the last cut, the furthest column of the REF plane
// (and the QUERY column it maps to) within ST_CAP of the cut in both planes where both planes map 1:1 onto each other
// and the two cells behind the cut have the cells in front of it as their only swap source.
//   tab_base[i]..tab_base[i+1]: the slots of alignment i (an alignment that needs more, or has no such column, gets
//   n_strips[i] = 0: it is left to the one-workgroup kernels if they can hold it, fits[i])
__global__ void k_strip_plan(DevBatch B, const AlnDesc *__restrict__ descs, const int32_t *__restrict__ work, int n,
                             const int32_t *__restrict__ tab_base, StripTab *__restrict__ tab, int32_t *__restrict__ n_strips,
                             int32_t *__restrict__ ok, const uint8_t *__restrict__ fits, uint32_t *__restrict__ err,
                             AlnOut *__restrict__ outs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AlnDesc d = descs[work[i]];
    const int Lq = d.Lq, Lr = d.Lr;
    const int32_t *q2r = B.hap_ptr[d.qs] + d.q_off, *r2q = B.ref_ptr[d.qs] + d.r_off;
    const int4 *cq = B.cand_q[d.qs] + d.q_off, *cr = B.cand_r[d.qs] + d.r_off;
    const int max_n = tab_base[i + 1] - tab_base[i];
    StripTab *T = tab + tab_base[i];
    int r_lo = 0, q_lo = 0, j = 0;
    bool good = true;
    for (;;) {
        if (j >= max_n) { good = false; break; }
        if (Lr - r_lo <= ST_CAP && Lq - q_lo <= ST_CAP) {
            T[j].lo[0] = q_lo; T[j].lo[1] = r_lo; T[j].hi[0] = Lq; T[j].hi[1] = Lr;
            j++;
            break;
        }
        int fr = -1, fq = -1;
        for (int r = min(r_lo + ST_CAP, Lr - 1); r > r_lo; r--) {
            const int q = r2q[r];
            if (q <= q_lo) break;                       // (monotone: nothing further left qualifies)
            if (q - q_lo > ST_CAP || q >= Lq) continue;
            if (r2q[r - 1] != q - 1 || q2r[q] != r || q2r[q - 1] != r - 1) continue;
            const int4 a = cr[r], b = cq[q];
            if (!(a.x < 0 || (a.x == q - 1 && a.y < 0))) continue;
            if (!(b.x < 0 || (b.x == r - 1 && b.y < 0))) continue;
            fr = r; fq = q;
            break;
        }
        if (fr < 0) { good = false; break; }
        T[j].lo[0] = q_lo; T[j].lo[1] = r_lo; T[j].hi[0] = fq; T[j].hi[1] = fr;
        j++;
        r_lo = fr; q_lo = fq;
    }
    n_strips[i] = good ? j : 0;
    // ok: 1 = the strips take it; 0 = the one-workgroup kernels do; 2 = neither can (those kernels skip it; the alignment is
    // reported with VPR_ST_ERR_LIMIT)
    ok[i] = good ? 1 : (fits[i] ? 0 : 2);
    if (!good && !fits[i]) { atomicOr(err, VPR_ST_ERR_LIMIT); atomicOr(&outs[work[i]].status, VPR_ST_ERR_LIMIT); }
}

// cell roles of a thread's chunk
#define ST_REAL 1
#define ST_GHOST 2

// ---------------------------------------------------------------------------
// K1 over strips.  prog[slot]: rows a strip has published; bnd[bnd_off[i] + j * Lt + t] = {D(QUERY, last column of strip j,
// row t), D(REF, ...)}.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(ST_NT) k_fwd_strip(DevBatch B, const AlnDesc *__restrict__ descs, const int32_t *__restrict__ work,
                                                     int n, const int32_t *__restrict__ tab_base, const StripTab *__restrict__ tab,
                                                     const int32_t *__restrict__ n_strips, const int32_t *__restrict__ order,
                                                     const int64_t *__restrict__ bnd_off,
                                                     int2 *__restrict__ bnd, int32_t *__restrict__ prog, uint8_t *__restrict__ ws,
                                                     AlnOut *__restrict__ outs, int use_ub) {
    constexpr int NT = ST_NT, C = ST_C, NC = NT * C;
    // The strip slot comes with a TICKET drawn at entry (a counter behind the two progress arrays, zeroed with them), not with
    // blockIdx: order[ticket] lists every alignment's strip j - 1 before its strip j, a strip waits for the strip before its own,
    // and that one's ticket has been drawn by a workgroup that is already running -- the wait cannot deadlock whatever order
    // the workgroups are dispatched in and however many fit on the device at once.
    __shared__ int s_slot;
    if (threadIdx.x == 0) s_slot = order[atomicAdd(prog + 2 * int(gridDim.x), 1)];
    __syncthreads();
    const int slot = s_slot;
    int i;
    {
        int lo = 0, hi = n;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tab_base[mid] <= slot) lo = mid; else hi = mid; }
        i = lo;
    }
    const int j = slot - tab_base[i], ns = n_strips[i];
    if (j >= ns) return;
    const int a = work[i];
    const AlnDesc d = descs[a];
    const StripTab S = tab[tab_base[i] + j];
    const int tid = threadIdx.x, lane = tid & 63;
    const int Lq = d.Lq, Lr = d.Lr, Lt = d.Lt;
    const int Lp[2] = {Lq, Lr};
    const bool has_left = j > 0, has_right = j + 1 < ns;
    const int bal[2] = {max(S.lo[0] - 1, 0) & ~(C - 1), max(S.lo[1] - 1, 0) & ~(C - 1)};    // column of thread 0's first cell
    __shared__ __align__(16) int32_t rowD[2][NC + 4];      // (column NC: a spare that stays D_INF)
    __shared__ int32_t wsc[2 * (NT / 64)];
    __shared__ int2 bin[64], bout[64];
    __shared__ int32_t blk_min;
    // use_ub: the alignment comes from a window level whose exit test failed; the distance that level found (paths inside
    // its window only) bounds the true one from above.  A cell beyond the bound is on no optimal path, and neither is
    // anything reached through it: a block of 64 rows in which every cell of the strip (and what comes in from the left)
    // is beyond it is not swept -- its cells count as unreachable, its flags are never read (the backward sweep only
    // follows flags of cells on optimal paths).
    int s_ub = D_INF;
    if (use_ub) { const int sv = outs[a].s; if (sv >= 0 && sv < D_INF / 2) s_ub = sv; }

    const uint8_t *seq[2] = {B.hap_seq[d.qs] + d.q_off, B.ref_seq + d.r_off};
    const uint8_t *Ts = B.hap_seq[d.ts] + d.t_off;
    const uint8_t *Tf = B.hap_flag[d.ts] + d.t_off;
    const int4 *cand[2] = {B.cand_q[d.qs] + d.q_off, B.cand_r[d.qs] + d.r_off};
    const int4 *cand2[2] = {B.cand2_q[d.qs] + d.q_off, B.cand2_r[d.qs] + d.r_off};
    uint8_t *mat[2] = {ws + d.mat_off[0], ws + d.mat_off[1]};
    const int rel0 = tid * C;
    int2 *my_bnd = bnd + bnd_off[i] + int64_t(j) * Lt;                   // the column this strip publishes
    const int2 *left_bnd = bnd + bnd_off[i] + int64_t(j - 1) * Lt;       // the column it consumes
    int32_t *my_prog = prog + tab_base[i] + j;

    // per-cell constants and roles
    uint8_t sb[2][C];
    int32_t c0[2][C];
    uint32_t multi[2] = {0, 0}, real[2] = {0, 0}, ghost[2] = {0, 0};
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int q = bal[p] + rel0 + c;
            sb[p][c] = 0xff;
            c0[p][c] = -1;
            if (q >= S.lo[p] && q < S.hi[p]) {
                real[p] |= 1u << c;
                sb[p][c] = seq[p][q];
                const int4 cc = cand[p][q];
                c0[p][c] = cc.x;
                if (cc.y >= 0) multi[p] |= 1u << c;
            } else if (has_left && q == S.lo[p] - 1) {
                ghost[p] |= 1u << c;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int c = 0; c < C; c++) asm volatile("" ::"v"(uint32_t(sb[p][c])), "v"(c0[p][c]));

    // row 0: D = q (INS chain from the origin), dist.cpp:300-305,397-405; cells left of the ghost do not exist
    int32_t dp[2][C];
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int q = bal[p] + rel0 + c;
            const bool left_ph = has_left && q < S.lo[p] - 1;
            dp[p][c] = left_ph ? D_INF : q;
            rowD[p][rel0 + c] = dp[p][c];
            if (real[p] & (1u << c)) mat[p][q] = (q == 0) ? F_MAT : F_INS;
        }
    }
    if (tid == 0) { bout[0] = make_int2(S.hi[0] - 1, S.hi[1] - 1); rowD[0][NC] = D_INF; rowD[1][NC] = D_INF; }
    __syncthreads();

    auto publish = [&](int t0, int t_last) {       // rows [t0, t_last] of this strip's last column
        if (has_right) {
            if (tid < 64 && t0 + tid <= t_last) my_bnd[t0 + tid] = bout[tid];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(my_prog, t_last + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (Lt == 1) publish(0, 0);

    uint32_t tchunk = 0;
    if (lane < Lt) tchunk = uint32_t(Ts[lane]) | (uint32_t(Tf[lane]) << 8);
    uint32_t tlast = 0;
    for (int t = 1; t < Lt; t++) {
        const int t0 = t & ~63;
        if ((t & 63) == 0 || t == 1) {
            if (has_left) {     // the left strip's column for the rows of this block
                const int need = min(t0 + 64, Lt);
                if (tid == 0)
                    while (__hip_atomic_load(my_prog - 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(4);
                __syncthreads();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (tid < 64 && t0 + tid < Lt) bin[tid] = left_bnd[t0 + tid];
                __syncthreads();
            }
            if (s_ub < D_INF) {
                if (tid == 0) blk_min = D_INF;
                __syncthreads();
                int m = D_INF;
#pragma unroll
                for (int p = 0; p < 2; p++)
#pragma unroll
                    for (int c = 0; c < C; c++) if ((real[p] | ghost[p]) & (1u << c)) m = min(m, dp[p][c]);      // (the row above the block:
                        // also the ghost's cell there, from which a diagonal move enters the block's first row)
                if (has_left && tid < 64 && t0 + tid < Lt) m = min(m, min(bin[tid].x, bin[tid].y));
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) m = min(m, __shfl_xor(m, o));
                if (lane == 0) atomicMin(&blk_min, m);
                __syncthreads();
                if (blk_min > s_ub) {
                    const int t_end = min(t0 + 63, Lt - 1);
#pragma unroll
                    for (int p = 0; p < 2; p++)
#pragma unroll
                        for (int c = 0; c < C; c++) { dp[p][c] = D_INF; rowD[p][rel0 + c] = D_INF; }
                    if (tid < 64) bout[tid] = make_int2(D_INF, D_INF);
                    __syncthreads();
                    publish(t0, t_end);
                    t = t_end;
                    continue;
                }
            }
        }
        if ((t & 63) == 0) {
            tlast = uint32_t(Ts[t - 1]) | (uint32_t(Tf[t - 1]) << 8);
            const int tt = t + lane;
            tchunk = 0;
            if (tt < Lt) tchunk = uint32_t(Ts[tt]) | (uint32_t(Tf[tt]) << 8);
        }
        const uint32_t cur = __builtin_amdgcn_readlane(tchunk, t & 63);
        const uint32_t prv = ((t & 63) == 0) ? tlast : uint32_t(__builtin_amdgcn_readlane(tchunk, (t - 1) & 63));
        const uint8_t Tt = cur & 0xff;
        const bool at = fwd_allow(int((prv >> 8) & 0xff));   // truth flag of row t-1, dist.cpp:338-339
        int2 gin = make_int2(0, 0);
        if (has_left) gin = bin[t - t0];

        int32_t bv[2][C];     // base - q
        uint8_t mk[2][C];     // flags achieving base
        int cmin[2];
        // Nearly every row has no cell with several swap sources whose base matches the truth base: then a cell's one source is
        // read without asking (a cell without a source reads the spare column NC = D_INF), all eight reads go out side by side,
        // and the cell code is straight-line (round 6: the row step is issue-bound, 16 waves on four SIMDs).  A wave with such a
        // cell takes the general code for the row.
        bool hard = false;
        if (at && (multi[0] | multi[1])) {
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int c = 0; c < C; c++) hard = hard || (((multi[p] >> c) & 1u) && sb[p][c] == Tt);
        }
        if (!__any(hard)) {
            int swv[2][C];
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int c = 0; c < C; c++) swv[p][c] = rowD[1 - p][c0[p][c] >= 0 ? c0[p][c] - bal[1 - p] : NC];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                int diag = (rel0 > 0) ? rowD[p][rel0 - 1] : D_INF;
                int run = D_INF;
                const int gval = (p == 0) ? gin.x : gin.y;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const int q = bal[p] + rel0 + c;
                    const int up = dp[p][c];
                    const bool match = sb[p][c] == Tt;
                    const int cm = diag + (match ? 0 : 1);
                    const int sw = (match && at && c0[p][c] >= 0) ? swv[p][c] : D_INF;
                    const int b = min(min(cm, up + 1), sw);
                    uint32_t m = ((match && diag == b) ? uint32_t(F_MAT) : 0u) | ((diag + 1 == b) ? uint32_t(F_SUB) : 0u) |
                                 ((up + 1 == b) ? uint32_t(F_DEL) : 0u) | ((sw == b) ? uint32_t(F_SWP) : 0u);
                    int nb = b - q;
                    const bool is_ghost = (ghost[p] >> c) & 1u, off = has_left && q < S.lo[p];      // (off: left of the strip; the ghost is one of them)
                    nb = is_ghost ? gval - q : (off ? D_INF : nb);
                    m = off ? 0u : m;
                    mk[p][c] = uint8_t(m);
                    bv[p][c] = nb;
                    run = min(run, nb);
                    diag = up;
                }
                cmin[p] = run;
            }
        } else {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int32_t *other = rowD[1 - p];
            const int ob = bal[1 - p];
            int diag = (rel0 > 0) ? rowD[p][rel0 - 1] : D_INF;
            int run = D_INF;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const int q = bal[p] + rel0 + c;
                const int up = dp[p][c];
                uint8_t m = 0;
                int nb;
                if (ghost[p] & (1u << c)) {
                    nb = (p == 0 ? gin.x : gin.y) - q;
                } else if (has_left && q < S.lo[p]) {
                    nb = D_INF;
                } else {
                    const bool match = sb[p][c] == Tt;
                    const int cm = match ? diag : diag + 1;
                    int b = min(cm, up + 1);
                    int sw = D_INF;
                    int choice = 0;
                    bool tie = false;
                    if (match && at && c0[p][c] >= 0) {
                        sw = other[c0[p][c] - ob];
                        if (multi[p] & (1u << c)) {
                            const int4 cc = cand[p][q];
                            const int v1 = other[cc.y - ob];
                            if (v1 <= sw) { tie = (v1 == sw); sw = v1; choice = 1; }
                            if (cc.z >= 0) {
                                const int v2 = other[cc.z - ob];
                                if (v2 <= sw) { tie = (v2 == sw); sw = v2; choice = 2; }
                                if (cc.w >= 0) {
                                    const int v3 = other[cc.w - ob];
                                    if (v3 <= sw) { tie = (v3 == sw); sw = v3; choice = 3; }
                                    const int4 c2 = cand2[p][q];
                                    const int more[4] = {c2.x, c2.y, c2.z, c2.w};
                                    for (int k = 0; k < 4 && more[k] >= 0; k++) {
                                        const int v = other[more[k] - ob];
                                        if (v <= sw) { tie = (v == sw); sw = v; choice = 4 + k; }
                                    }
                                }
                            }
                        }
                        b = min(b, sw);
                    }
                    if (match && diag == b) m |= F_MAT;
                    if (diag + 1 == b) m |= F_SUB;
                    if (up + 1 == b) m |= F_DEL;
                    if (sw == b) m |= F_SWP | f_choice_bits(choice) | (tie ? F_TIE : 0);
                    nb = b - q;
                }
                mk[p][c] = m;
                bv[p][c] = nb;
                run = min(run, nb);
                diag = up;
            }
            cmin[p] = run;
        }
        }
        int carryQ = cmin[0], carryR = cmin[1];
        block_excl_prefix_min2<NT>(carryQ, carryR, wsc);   // barrier inside: all reads of rowD are done
        const int carry[2] = {carryQ, carryR};
#pragma unroll
        for (int p = 0; p < 2; p++) {
            uint8_t fl[C];
            int run = carry[p];
            int left = carry[p] + bal[p] + rel0 - 1;   // D of the cell in front of the chunk in this row
#pragma unroll
            for (int c = 0; c < C; c++) {
                const int q = bal[p] + rel0 + c;
                const int nb = bv[p][c];
                uint8_t f = 0;
                if (nb <= run) { run = nb; f = mk[p][c]; }
                const int Dn = min(run + q, D_INF);
                if (left + 1 == Dn && q > 0) f |= F_INS;
                fl[c] = f;
                dp[p][c] = Dn;
                left = Dn;
                rowD[p][rel0 + c] = Dn;
                if ((real[p] & (1u << c)) && q == S.hi[p] - 1) {
                    if (p == 0) bout[t - t0].x = Dn; else bout[t - t0].y = Dn;
                }
            }
            uint8_t *row = mat[p] + size_t(t) * d.pitch[p] + bal[p] + rel0;
            if (real[p] == (1u << C) - 1) {
                typename FlagVec<C>::T v;
                __builtin_memcpy(&v, fl, C);
                *reinterpret_cast<typename FlagVec<C>::T *>(row) = v;
            } else if (real[p]) {
#pragma unroll
                for (int c = 0; c < C; c++) if (real[p] & (1u << c)) row[c] = fl[c];
            }
        }
        lds_barrier<NT>();
        if ((t & 63) == 63 || t == Lt - 1) publish(t0, t);
    }
    // end cells
#pragma unroll
    for (int c = 0; c < C; c++) {
        if ((real[0] & (1u << c)) && bal[0] + rel0 + c == Lq - 1) outs[a].dist_q = dp[0][c];
        if ((real[1] & (1u << c)) && bal[1] + rel0 + c == Lr - 1) outs[a].dist_r = dp[1][c];
    }
    (void)Lp;
}

// ---------------------------------------------------------------------------
// K2 over strips, right to left.  A strip publishes, per row and plane, {score of its first column, that
// cell's FORWARD flag byte} (the consumer's successor cell, whose flags the producer has overwritten by then);
// bbnd[bnd_off[i] + j * Lt + t] is the record of strip j + 1's first column.  prog counts rows from the last one.
//
// The cells with a score -- the cells on optimal paths -- are a thin bundle: in a row of 2 x 4 096 cells a few dozen.  A row
// step therefore has two paths per wavefront (round 6; a row cost 6.8 us when every wave took the full one, the launch was as
// long as the longest alignment's rows x that):
//   * a wave none of whose cells can have a scored successor -- no score in its own columns of the row below or in the column
//     right of them, no swap target inside the range of scored columns of the other plane (rng: the range of the row below, two
//     LDS words per plane), not the ghost cell of a scored record -- only forms the max-plus maps of its INS links for the
//     row's suffix scan; a wave into which nothing flows from the right either only moves its flag rows along;
//   * the others run the cell code.
// The suffix scan is a DPP prefix scan (wave_prefix_mp2, pr_band.hip): the columns are mirrored inside a wave (lane 0 owns the
// wave's highest columns).  Flag rows and per-cell constants are packed four to a register: the kernel runs 16 waves per
// compute unit on 128 registers each and used to spill.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(ST_NT) k_bwd_strip(DevBatch B, const AlnDesc *__restrict__ descs, const int32_t *__restrict__ work,
                                                     int n, const int32_t *__restrict__ tab_base, const StripTab *__restrict__ tab,
                                                     const int32_t *__restrict__ n_strips, const int64_t *__restrict__ bnd_off,
                                                     int4 *__restrict__ bbnd, int32_t *__restrict__ prog, uint8_t *__restrict__ ws,
                                                     AlnOut *__restrict__ outs) {
    constexpr int NT = ST_NT, C = ST_C, NC = NT * C;
    static_assert(C == 4, "a thread's flag bytes of a row are one 32-bit word per plane");
    // One workgroup per alignment, its strips one after the other from the right: the optimal paths cross the strips one
    // after the other, so strips side by side would mostly wait for each other (and hold a compute unit while they do);
    // here a strip's rows off the paths are skipped in blocks, and the sweep costs what the paths' rows cost.
    const int i = blockIdx.x;
    if (i >= n) return;
    const int ns = n_strips[i];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rel0 = ((tid & ~63) | (63 - lane)) * C;      // mirrored inside the wave
    __shared__ __align__(16) int32_t srow[2][NC + 4];                  // scores of row t+1
    __shared__ __align__(16) uint8_t frow[2][2][NC + 16];              // [buffer][plane]: forward flags of rows t+1 / t
    __shared__ int32_t wsc[4 * (NT / 64)];
    __shared__ int4 rin[66], rout[64];                   // rin[x]: record of row (t0 - 1) + x
    __shared__ int32_t actf[2];                          // [row parity]: a cell of the row has a score
    __shared__ int32_t rng[2][2][2];                     // [row parity][plane]: lowest / highest LDS column with a score
    __shared__ int32_t blk_or;
    for (int j = ns - 1; j >= 0; j--) {
    const int a = work[i];
    const AlnDesc d = descs[a];
    const StripTab S = tab[tab_base[i] + j];
    const int Lq = d.Lq, Lr = d.Lr, Lt = d.Lt;
    const int Lp[2] = {Lq, Lr};
    const bool has_left = j > 0, has_right = j + 1 < ns;
    const int bal[2] = {max(S.lo[0] - 1, 0) & ~(C - 1), max(S.lo[1] - 1, 0) & ~(C - 1)};
    const int32_t *ptr[2] = {B.hap_ptr[d.qs] + d.q_off, B.ref_ptr[d.qs] + d.r_off};
    const uint8_t *pfl[2] = {B.hap_flag[d.qs] + d.q_off, B.ref_flag[d.qs] + d.r_off};
    const int4 *cand[2] = {B.cand_q[d.qs] + d.q_off, B.cand_r[d.qs] + d.r_off};
    const int4 *cand2[2] = {B.cand2_q[d.qs] + d.q_off, B.cand2_r[d.qs] + d.r_off};
    uint8_t *mat[2] = {ws + d.mat_off[0], ws + d.mat_off[1]};
    const int end_plane = outs[a].end_plane;
    const bool inb[2] = {bal[0] + rel0 < d.pitch[0], bal[1] + rel0 < d.pitch[1]};     // the chunk starts inside the flag row
    int4 *my_bnd = bbnd + bnd_off[i] + int64_t(j - 1) * Lt;              // what this strip publishes (boundary j-1 | j)
    const int4 *right_bnd = bbnd + bnd_off[i] + int64_t(j) * Lt;         // what it consumes (boundary j | j+1)
    int32_t *my_prog = prog + tab_base[i] + j;

    // per-cell constants (pr_kernels.hip: k_bwd) and roles; kcw: the cells' constant bytes (bit 0 tp, bit 3 tp(z))
    int32_t zq[2][C];             // LDS column of the cell's swap target in the other plane (NC, a spare column without flags or score: none)
    uint32_t kcw[2] = {0, 0};
    uint32_t keyw[2] = {0xffffffffu, 0xffffffffu};     // per cell: the forward flag bits of the target that name this cell as its swap source
    uint32_t real[2] = {0, 0}, ghost[2] = {0, 0}, rmask[2] = {0, 0};      // rmask: byte mask of the real cells
    int own[2] = {-1, -1};                                                // cell of the strip's first column, if it is this thread's
    int zlo[2] = {1 << 30, 1 << 30}, zhi[2] = {-1, -1};                   // range of the cells' swap targets (LDS columns of the other plane)
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int q = bal[p] + rel0 + c;
            zq[p][c] = NC;
            uint32_t kc = 0;
            const bool is_real = q >= S.lo[p] && q < S.hi[p];
            const bool is_ghost = has_right && q == S.hi[p];
            if (is_real) { real[p] |= 1u << c; rmask[p] |= 0xffu << (8 * c); }
            if (is_ghost) ghost[p] |= 1u << c;
            if (is_real && q == S.lo[p]) own[p] = c;
            if ((is_real || is_ghost) && q < Lp[p]) {
                const int pq = ptr[p][q];
                const int fq = pfl[p][q];
                if (p == 0 && q > 0 && ((pq != ptr[0][q - 1] + 1) || (fq & PB))) kc |= 1;  // dist.cpp:572-574
                const int z = pq + 1;
                if (is_real && fwd_allow(fq) && z > 0 && z < Lp[1 - p] && bwd_allow(pfl[1 - p][z])) {
                    const int4 cc = cand[1 - p][z];
                    int rank = -1;
                    if (cc.x == q) rank = 0; else if (cc.y == q) rank = 1; else if (cc.z == q) rank = 2; else if (cc.w == q) rank = 3;
                    else if (cc.w >= 0) {
                        const int4 c2 = cand2[1 - p][z];
                        if (c2.x == q) rank = 4; else if (c2.y == q) rank = 5; else if (c2.z == q) rank = 6; else if (c2.w == q) rank = 7;
                    }
                    if (rank >= 0) {
                        zq[p][c] = z - bal[1 - p];      // (index into the other plane's LDS rows)
                        zlo[p] = min(zlo[p], zq[p][c]); zhi[p] = max(zhi[p], zq[p][c]);
                        keyw[p] = (keyw[p] & ~(0xffu << (8 * c))) | (f_swp_key(rank) << (8 * c));
                        if (p == 1) {  // z on the QUERY plane: leaving it scores tp(z), dist.cpp:656-658
                            const int pz = ptr[0][z];
                            if ((pz != ptr[0][z - 1] + 1) || (pfl[0][z] & PB)) kc |= 8;
                        }
                    }
                }
            }
            kcw[p] |= kc << (8 * c);
        }
    }
    // tp of the QUERY-plane cell right of this chunk (constant over rows)
    int xtp_right = 0;
    {
        const int qn = bal[0] + rel0 + C;
        if (qn < Lq && qn <= S.hi[0]) xtp_right = ((ptr[0][qn] != ptr[0][qn - 1] + 1) || (pfl[0][qn] & PB)) ? 1 : 0;
    }
    auto byte_of = [](uint32_t w, int c) -> uint32_t { return (w >> (8 * c)) & 0xffu; };
    uint32_t gmask[2] = {0, 0};       // byte mask of the ghost cell
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int c = 0; c < C; c++) if (ghost[p] & (1u << c)) gmask[p] = 0xffu << (8 * c);
    auto ghost_word = [&](uint32_t w, int p, uint32_t gb) -> uint32_t {     // the real cells' bytes of w, the ghost cell's byte = gb
        return (w & rmask[p]) | (((gb & 0xffu) * 0x01010101u) & gmask[p]);
    };

    // the right strip's records for the rows of a block: rows [t0 - 1, t0 + 64] as far as they exist
    auto acquire_block = [&](int t0) {
        if (has_right) {
            const int lowest = max(t0 - 1, 0);                 // lowest row needed
            if (tid == 0)
                while (__hip_atomic_load(my_prog + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < Lt - lowest) __builtin_amdgcn_s_sleep(4);
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (tid < 66) {
                const int t = t0 - 1 + tid;
                if (t >= 0 && t < Lt) rin[tid] = right_bnd[t];
            }
            __syncthreads();
        }
    };
    auto publish = [&](int t0, int t_hi) {       // rows [t0, t_hi] of this strip's first column
        if (has_left) {
            if (tid < 64 && t0 + tid <= t_hi) my_bnd[t0 + tid] = rout[tid];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(my_prog, Lt - t0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    const uint8_t *cell0[2] = {mat[0] + bal[0] + rel0, mat[1] + bal[1] + rel0};      // the thread's first cell in row 0
    auto load_row = [&](int p, int t) -> uint32_t {      // a thread's four flag bytes of row t (0 outside the flag row)
        return (t >= 0 && inb[p]) ? *reinterpret_cast<const uint32_t *>(cell0[p] + size_t(t) * d.pitch[p]) : 0u;
    };
    auto store_row = [&](int p, int t, uint32_t w) {
        uint8_t *row = mat[p] + size_t(t) * d.pitch[p] + bal[p] + rel0;
        if (real[p] == (1u << C) - 1) {
            *reinterpret_cast<uint32_t *>(row) = w;
        } else if (real[p]) {
#pragma unroll
            for (int c = 0; c < C; c++) if (real[p] & (1u << c)) row[c] = uint8_t(w >> (8 * c));
        }
    };

    int32_t sc[2][C];   // scores of row t+1 (S_NEG = unreachable)
    uint32_t f1w[2];    // forward flags of row t+1
    uint32_t f0w[2];    // forward flags of row t
    uint32_t pf[2];     // forward flags of row t-1 (requested one row ahead)
    acquire_block((Lt - 1) & ~63);
    {
        const int t0 = (Lt - 1) & ~63;
        const int4 Rl = has_right ? rin[Lt - 1 - (t0 - 1)] : make_int4(S_NEG, 0, S_NEG, 0);
#pragma unroll
        for (int p = 0; p < 2; p++) {
            f0w[p] = ghost_word(load_row(p, Lt - 1), p, uint32_t(p == 0 ? Rl.y : Rl.w));
            f1w[p] = 0;
#pragma unroll
            for (int c = 0; c < C; c++) sc[p][c] = S_NEG;
            *reinterpret_cast<int4 *>(&srow[p][rel0]) = make_int4(S_NEG, S_NEG, S_NEG, S_NEG);
            *reinterpret_cast<uint32_t *>(&frow[0][p][rel0]) = 0u;
            *reinterpret_cast<uint32_t *>(&frow[1][p][rel0]) = f0w[p];
            pf[p] = load_row(p, Lt - 2);
        }
        if (tid == 0) {
            for (int p = 0; p < 2; p++) {
                srow[p][NC] = S_NEG; frow[0][p][NC] = 0; frow[1][p][NC] = 0;
                for (int b = 0; b < 2; b++) { rng[b][p][0] = 1 << 30; rng[b][p][1] = -1; }
            }
            actf[0] = actf[1] = 0;
        }
    }
    lds_barrier<NT>();
    uint32_t tie_used = 0;

    for (int t = Lt - 1; t >= 0; t--) {
        const int t0 = t & ~63;
        if ((t & 63) == 63 && t != Lt - 1) {
            acquire_block(t0);
            // A whole block of 64 rows off the optimal paths -- nothing in the row above it, nothing coming in from the right
            // in any of its rows -- is not swept: the records for the left neighbour say "no score", and the sweep resumes below it.
            blk_or = 0;
            __syncthreads();
            if (tid < 64 && has_right) { const int4 Rr = rin[tid + 1]; if (Rr.x >= 0 || Rr.z >= 0) blk_or = 1; }
            if (tid == 0 && actf[(t + 1) & 1]) blk_or = 1;
            __syncthreads();
            if (!blk_or) {
                if (tid < 64 && has_left)      // the forward flags of this strip's first column
                    rout[tid] = make_int4(S_NEG, mat[0][size_t(t0 + tid) * d.pitch[0] + S.lo[0]], S_NEG, mat[1][size_t(t0 + tid) * d.pitch[1] + S.lo[1]]);
                __syncthreads();
                publish(t0, t);
                if (t0 > 0) {       // state for row t0 - 1: its forward flags (the ghost's from the record), nothing scored above
                    const int curm = (Lt - 1 - (t0 - 1) + 1) & 1;
                    const int4 Rg = has_right ? rin[0] : make_int4(S_NEG, 0, S_NEG, 0);
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        f0w[p] = ghost_word(load_row(p, t0 - 1), p, uint32_t(p == 0 ? Rg.y : Rg.w));
                        f1w[p] = 0;
                        pf[p] = load_row(p, t0 - 2);
#pragma unroll
                        for (int c = 0; c < C; c++) sc[p][c] = S_NEG;
                        *reinterpret_cast<uint32_t *>(&frow[curm][p][rel0]) = f0w[p];
                    }
                    if (tid == 0) {
                        actf[0] = actf[1] = 0;
                        for (int p = 0; p < 2; p++) for (int b = 0; b < 2; b++) { rng[b][p][0] = 1 << 30; rng[b][p][1] = -1; }
                    }
                    lds_barrier<NT>();
                }
                t = t0;
                continue;
            }
        }
        const int cur = (Lt - 1 - t + 1) & 1;      // buffer holding row t's forward flags
        const int nxt = cur ^ 1;                   // buffer holding row t+1's forward flags
        int4 R = make_int4(S_NEG, 0, S_NEG, 0), Rm = make_int4(S_NEG, 0, S_NEG, 0);      // records of rows t and t-1
        if (has_right) { R = rin[t - (t0 - 1)]; if (t > 0) Rm = rin[t - 1 - (t0 - 1)]; }

        // A row in which no cell of the strip can be on an optimal path -- none was in the row below, and nothing comes in
        // from the strip to the right -- only moves the flag rows along: the optimal paths of an alignment are a thin bundle,
        // most strips are off it in most rows.
        const bool act = t == Lt - 1 || actf[(t + 1) & 1] != 0 || R.x >= 0 || R.z >= 0;
        if (tid == 0) actf[t & 1] = 0;
        int32_t base[2][C];
        uint32_t bmw[2] = {0, 0};       // moves reaching base
        uint32_t lkw[2] = {0xffffffffu, 0xffffffffu};      // INS link from cell c+1 into c: tp value (0/1) or 0xff broken
        int inc[2] = {S_NEG, S_NEG};    // what flows into the thread's cells from the right
        bool full2[2] = {false, false}; // (wave-uniform, per plane) the cells of the plane take the full second half
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int c = 0; c < C; c++) base[p][c] = S_NEG;
        if (act) {
            const int xs_r[2] = {srow[0][rel0 + C], srow[1][rel0 + C]};
            const uint32_t xf0_r[2] = {frow[cur][0][rel0 + C], frow[cur][1][rel0 + C]};
            // per plane: can a cell of this thread have a scored successor?  (superset: see the header)
            bool maybe[2], live1[2], need1[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                bool m = t == Lt - 1 || xs_r[p] >= 0;
#pragma unroll
                for (int c = 0; c < C; c++) m = m || sc[p][c] >= 0;
                m = m || (zlo[p] <= rng[(t + 1) & 1][1 - p][1] && zhi[p] >= rng[(t + 1) & 1][1 - p][0]);
                if (ghost[p]) m = m || (p == 0 ? R.x : R.z) >= 0;
                maybe[p] = m;
                live1[p] = __any(m);
                // the INS chain of the row runs through ALL cells of this wave?  (the link into cell c is the F_INS bit of cell
                // c + 1's forward flags of this row; real cells only).  Nearly never: else the wave's map is "broken" whatever its links
                const uint32_t w = (f0w[p] >> 8) | (xf0_r[p] << 24);
                need1[p] = live1[p] || __all(real[p] == (1u << C) - 1 && (w & 0x01010101u) == 0x01010101u);
            }
            MP g[2], ex[2];         // ex: everything right of this thread inside the wave
            bool have_maps[2] = {false, false};
            // the threads' max-plus maps of the planes in `todo` (with_base: and the cells' scores over their successors in the
            // row below), scanned inside the wave
            auto thread_maps = [&](const bool (&todo)[2], const bool (&with_base)[2]) {
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    g[p].A = S_NEG; g[p].B = -1;
                    if (!todo[p]) continue;
                    const int o = 1 - p;
                    const int xtp_r = (p == 0) ? xtp_right : 0;
                    // (what the cells read from LDS is requested side by side, before any of it is used: a cell without a swap
                    // target reads the spare column NC -- no flags, no score)
                    uint32_t xf_r = 0, zf[C];
                    int zs[C];
                    if (with_base[p]) {
                        xf_r = frow[nxt][p][rel0 + C];
#pragma unroll
                        for (int c = 0; c < C; c++) { zf[c] = frow[nxt][o][zq[p][c]]; zs[c] = srow[o][zq[p][c]]; }
                    }
                    const int Rv = (p == 0) ? R.x : R.z;
                    uint32_t bm = 0, lk = 0;
                    MP G; G.A = S_NEG; G.B = 0;
#pragma unroll
                    for (int c = C - 1; c >= 0; c--) {
                        const uint32_t kc = byte_of(kcw[p], c);
                        const int xtp = (c == C - 1) ? xtp_r : int(byte_of(kcw[p], c + 1) & 1u);
                        const bool is_real = (real[p] >> c) & 1u;
                        int best = S_NEG; uint32_t m = 0;
                        if (with_base[p]) {
                            const int xs = (c == C - 1) ? xs_r[p] : sc[p][c + 1];
                            const uint32_t xf = (c == C - 1) ? xf_r : byte_of(f1w[p], c + 1);
                            const uint32_t dg = f_diag(xf);
                            best = dg ? xs + xtp : S_NEG; m = dg;
                            {
                                const bool on = (byte_of(f1w[p], c) & F_DEL) != 0;
                                const int v = sc[p][c];
                                const bool gt = on && v > best, eqv = on && v == best;
                                m = gt ? uint32_t(F_DEL) : (eqv ? (m | F_DEL) : m);
                                best = gt ? v : best;
                            }
                            {
                                const bool on = (zf[c] & F_SWP_KEY_MASK) == byte_of(keyw[p], c);      // (a cell without a target has key 0xff)
                                const int v = zs[c] + int((kc >> 3) & 1u);
                                tie_used += (on && v >= 0 && (zf[c] & F_TIE) && is_real) ? 1u : 0u;
                                const bool gt = on && v > best, eqv = on && v == best;
                                m = gt ? uint32_t(F_SWP) : (eqv ? (m | F_SWP) : m);
                                best = gt ? v : best;
                            }
                            if (t == Lt - 1 && p == end_plane && bal[p] + rel0 + c == Lp[p] - 1) { best = 0; m = F_MAT; }  // dist.cpp:538-546
                        }
                        const uint32_t xf0 = (c == C - 1) ? xf0_r[p] : byte_of(f0w[p], c + 1);
                        const int l = (is_real && (xf0 & F_INS)) ? xtp : -1;
                        // (a cell that is not the strip's: no score, no moves -- but the ghost cell's score in this row is given and
                        // final: nothing flows into it)
                        best = is_real ? best : (((ghost[p] >> c) & 1u) ? Rv : S_NEG);
                        m = is_real ? m : 0u;
                        base[p][c] = best;
                        bm |= m << (8 * c);
                        lk |= (uint32_t(l) & 0xffu) << (8 * c);
                        MP F; F.A = best; F.B = l;
                        G = (c == C - 1) ? F : mp_compose(F, G);
                    }
                    bmw[p] = bm; lkw[p] = lk;
                    g[p] = G;
                    have_maps[p] = true;
                }
                // suffix composition inside the wave: a prefix scan in lane order (the columns are mirrored)
                wave_prefix_mp2(g[0], g[1]);
#pragma unroll
                for (int p = 0; p < 2; p++)
                    if (todo[p]) { ex[p].A = wave_shr1(g[p].A, S_NEG); ex[p].B = wave_shr1(g[p].B, 0); }
            };
            int4 sum = make_int4(S_NEG, -1, S_NEG, -1);      // the wave's maps, both planes
            if (need1[0] || need1[1]) {
                thread_maps(need1, live1);
                if (need1[0]) { sum.x = g[0].A; sum.y = g[0].B; }
                if (need1[1]) { sum.z = g[1].A; sum.w = g[1].B; }
            }
            if (lane == 63) *reinterpret_cast<int4 *>(&wsc[wave * 4]) = sum;
            lds_barrier<NT>();          // (all reads of the score rows / flag buffer nxt / rng of the row below are done)
            // what flows into this wave from the waves right of it
            int xin[2] = {S_NEG, S_NEG};
            if (wave + 1 < NT / 64) {
                const int4 nb = *reinterpret_cast<const int4 *>(&wsc[(wave + 1) * 4]);
                if (nb.y < 0 && nb.w < 0) { xin[0] = nb.x; xin[1] = nb.z; }      // both chains break inside the neighbour: its own scores
                else {
                    for (int w = NT / 64 - 1; w > wave; w--) {
                        const int4 e = *reinterpret_cast<const int4 *>(&wsc[w * 4]);
                        xin[0] = (e.y < 0) ? e.x : max(e.x, xin[0] + e.y);
                        xin[1] = (e.w < 0) ? e.z : max(e.z, xin[1] + e.w);
                    }
                }
            }
            const bool late[2] = {!have_maps[0] && xin[0] >= 0, !have_maps[1] && xin[1] >= 0};      // (wave-uniform)
            if (late[0] || late[1]) { const bool nob[2] = {false, false}; thread_maps(late, nob); }
#pragma unroll
            for (int p = 0; p < 2; p++) {
                if (have_maps[p]) {
                    inc[p] = (ex[p].B < 0) ? ex[p].A : max(ex[p].A, xin[p] + ex[p].B);
                    full2[p] = __any(maybe[p] || inc[p] >= 0);
                }
            }
            if (tid == 0) { rng[(t + 1) & 1][0][0] = rng[(t + 1) & 1][1][0] = 1 << 30; rng[(t + 1) & 1][0][1] = rng[(t + 1) & 1][1][1] = -1; }
        }
        bool anyp = false;
        // (a plane of a wave off the live path: its scores were and stay S_NEG -- in the registers and in srow; the flag bytes of
        // its cells stay what the forward sweep left: nothing reads the flags of a cell without a score, the walk follows scored
        // cells only)
#pragma unroll
        for (int p = 0; p < 2; p++) {
            if (full2[p]) {
                int prev = inc[p];
                int lo = 1 << 30, hi = -1;
                uint32_t outw = 0;
#pragma unroll
                for (int c = C - 1; c >= 0; c--) {
                    int v = base[p][c];
                    uint32_t m = byte_of(bmw[p], c);
                    const int l = int(int8_t(byte_of(lkw[p], c)));
                    const int w = prev + l;
                    const bool gt = l >= 0 && w > v, eqv = l >= 0 && w == v;
                    m = gt ? uint32_t(F_INS) : (eqv ? (m | F_INS) : m);
                    v = gt ? w : v;
                    m = v < 0 ? 0u : m;
                    v = v < 0 ? S_NEG : v;
                    sc[p][c] = v;
                    lo = v >= 0 ? min(lo, rel0 + c) : lo;
                    hi = v >= 0 ? max(hi, rel0 + c) : hi;
                    outw |= (m ? (m | (byte_of(f0w[p], c) & F_KEEP)) : 0u) << (8 * c);
                    prev = v;
                }
                *reinterpret_cast<int4 *>(&srow[p][rel0]) = make_int4(sc[p][0], sc[p][1], sc[p][2], sc[p][3]);
                if (hi >= 0) { anyp = true; atomicMin(&rng[t & 1][p][0], lo); atomicMax(&rng[t & 1][p][1], hi); }
                store_row(p, t, outw);
            }
            if (own[p] >= 0) {       // this strip's first column: published
#pragma unroll
                for (int c = 0; c < C; c++) {
                    if (own[p] == c) {
                        const int v = sc[p][c];
                        const int fb = int(byte_of(f0w[p], c));
                        if (p == 0) { rout[t - t0].x = v; rout[t - t0].y = fb; } else { rout[t - t0].z = v; rout[t - t0].w = fb; }
                    }
                }
            }
            f1w[p] = f0w[p];
            if (t > 0) {
                f0w[p] = ghost_word(pf[p], p, uint32_t(p == 0 ? Rm.y : Rm.w));
                *reinterpret_cast<uint32_t *>(&frow[nxt][p][rel0]) = f0w[p];   // nxt becomes "cur" of row t-1
            }
            pf[p] = load_row(p, t - 2);
        }
        if (anyp) actf[t & 1] = 1;
        lds_barrier<NT>();
        if ((t & 63) == 0) publish(t0, min(t0 + 63, Lt - 1));
    }
    if (j == 0) {
#pragma unroll
        for (int c = 0; c < C; c++)
            if (bal[0] + rel0 + c == 0) outs[a].beg_plane = (sc[0][c] >= 0) ? VPR_PLANE_QUERY : VPR_PLANE_REF;   // dist.cpp:811-814
    }
    if (tie_used) { atomicOr(&outs[a].status, VPR_ST_SWAP_TIE); outs[a].band_ok = TIE_MARK(0); atomicAdd(&outs[a].n_sec, int(tie_used)); }
    __threadfence();
    __syncthreads();
    }
}

#endif
