// pr_tie.hip -- replay of the reference's container order for swap-predecessor ties.
//
// calc_prec_recall_aln keeps ONE swap predecessor per cell, "(*swap_pred_maps[i])[z] = x" (dist.cpp:347,376): when
// several sources x reach z at the optimal cost, the last one popped from the FIFO wins.  The FIFO of wave s is seeded
// by iterating std::unordered_set<idx1> prev_wave (dist.cpp:395; hash dist.h:42-50), so the pop order is a function
// of libstdc++'s bucket counts, its insert-at-bucket-begin rule and its rehash history.  The window / dense forward
// kernels cannot know that order; they keep the highest source index, set F_TIE on the cell, and the backward kernels
// mark an alignment (AlnOut::band_ok = -(tag + 1), VPR_ST_SWAP_TIE) when such a cell is on the optimal DAG.  For those
// alignments only, the host re-runs the forward sweep and then this kernel, which replays the reference's expansion
// cell by cell and rewrites the choice field of every tied cell with the predecessor the reference keeps; the
// backward sweep, walk and credit then run on the corrected flags.
//
// One workgroup (TIE_NW waves: four or eight, the template parameter) per alignment.  What is replayed, exactly:
//   * BFS of a wave (dist.cpp:317-381): the FIFO is a log in HBM; up to TIE_U x 64 TIE_NW entries are popped per step, thread e
//     expands entry e (MAT child, then SWP child, as the reference pushes them).  "not done and not in curr_wave"
//     == "never pushed before": every candidate push carries a running candidate id, an atomicMin on the cell's
//     stamp keeps the first one, and the lanes whose id survived append their cell in id order (ballot prefix).
//     A chunk never overtakes the FIFO: children go behind everything already queued.
//   * pop order of two cells of one wave == order of their stamps (pushes are appended in candidate-id order), so
//     "last writer of z" = the allowed source of z with the largest stamp among those popped in z's wave.
//   * prev_wave iteration order (dist.cpp:395): libstdc++'s _Hashtable keeps one singly linked list; a node whose
//     bucket is empty goes to the list head, otherwise right behind its bucket's head node; a rehash re-inserts the
//     list in order into the new bucket array; clear() keeps the bucket count.  Inserting a_0..a_{m-1} into an empty
//     table of B buckets therefore yields: buckets ordered by DEscending first insertion, inside a bucket
//     DEscending insertion -- position(a_i) = #{j : first(bkt_j) > first(bkt_i)} + #{j > i : bkt_j == bkt_i} -- which
//     is computed with two atomics passes and a suffix sum instead of chasing a list.  A wave that outgrows the
//     bucket count B (max_load_factor 1) re-inserts the current list followed by the remaining elements into
//     next_bkt(2 B) buckets: one more pass.  (tests/test_tie_order.py checks this model against std::unordered_set.)
//   * seeding of the next wave (dist.cpp:395-424): INS, DEL, SUB targets of every prev_wave element in iteration order,
//     deduplicated the same way.
// Scratch per alignment (planned by the host): stamps, 4 B per cell of the two planes' diagonal-major grids
// ((Lq + Lr + 2 Lt - 2) x Lt words), preset to
// 0xffffffff; two FIFO logs, two order buffers and three work arrays of `cap` entries; `bcap` 8-byte bucket words.
#ifndef PR_TIE_HIP_
#define PR_TIE_HIP_

#define TIE_NEVER 0xffffffffu
#define TIE_N_BUCKETS 26
// bucket counts libstdc++'s _Prime_rehash_policy walks through from its first allocation when it doubles
// (_M_next_bkt(2 * n)): measured on the image's libstdc++ (tests/test_tie_order.py re-derives it)
#define TIE_BUCKET_LIST {13u, 29u, 59u, 127u, 257u, 541u, 1109u, 2357u, 5087u, 10273u, 20753u, 42043u, 85229u, 172933u, \
                         351061u, 712697u, 1447153u, 2938679u, 5967347u, 12117689u, 24607243u, 49969847u, 101473717u,    \
                         206062531u, 418451333u, 849749479u}
__constant__ uint32_t TIE_BUCKETS[TIE_N_BUCKETS] = TIE_BUCKET_LIST;
static const uint32_t TIE_BUCKETS_HOST[TIE_N_BUCKETS] = TIE_BUCKET_LIST;

struct TieJob {
    int32_t a;            // alignment id
    int32_t cap;          // entries per FIFO log / order buffer (largest wave the job can hold)
    int32_t bcap;         // bucket words available
    int32_t pad;
    int64_t stamp_off;    // uint32 index into the scratch: stamps[(Lq + Lr) * Lt]
    int64_t buf_off;      // uint32 index (even): 10 * cap words
    int64_t bkt_off;      // uint64 index: bcap words
    // mode 1 ("early"): the job runs before the forward sweep is repeated, on the bytes the backward sweep of the round that
    // marked the alignment left in that round's workspace (old_*): it decides only the n_used tied cells that sweep
    // consulted (F_KEEP), appends {alignment, cell, row, choice} to the launch's decision list instead of patching, and
    // stops behind the wave that decides the last of them.  mode 2 ("speculative"): launched right behind the forward
    // sweep of a long alignment that has a tied cell within its distance, before anyone knows whether the backward sweep
    // consults it; reads no flag bytes (a tie is a cell with two allowed sources popped in its wave) and lists every tied
    // cell.  mode 0: behind the repeated forward sweep, patches its flags in place.
    int32_t mode, n_used;
    // stamp grids: per plane the diagonals q - t in [dlo, dlo + dn), dn x Lt words each (the host bounds them from the
    // haps' pointer ranges and the alignment's distance; a push outside fails the job, which then runs again with all)
    int32_t dlo[2], dn[2];
    int32_t old_band_w, old_pitch[2], pad2;
    int64_t old_mat_off[2], old_blo_off;
    uint8_t *old_arena;
    int32_t dbg_us, dbg_steps, dbg_cells, dbg_waves, dbg_nres, dbg_lastw;
    int32_t dbg_oob[4];   // (debug) first cell outside the stamp grids: plane, position, row, wave
    int32_t dbg_la[6];    // (debug) narrow steps: all, not in one row, frontier not reproduced (count / slots), look-ahead steps, levels they committed
    int32_t dbg_t[8];     // us in: BFS, patch, order A, B, suffix, C, seeding, setup   // written by the kernel (the jobs live in host-pinned memory): VPR_DEBUG
};
#define TIE_BUF_WORDS 10
#define TIE_BLIST 8          // members a bucket's list holds (load factor <= 1: more is rare and takes the slow pass)
#define TIE_BKT_WORDS (2 + 1 + TIE_BLIST)   // uint32 words per bucket: first-insertion word, member count, member list

// coherent load (bypasses the CU's L1): for words other lanes modify with atomics, which execute in L2.  Arrays that are
// only written with plain stores (queue logs, order buffers, F / K) are read with plain loads behind a tie_wait().
__device__ __forceinline__ uint32_t tie_ld(const uint32_t *p) {
    return __hip_atomic_load(const_cast<uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tie_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// address of the forward flag byte of cell (plane p, column x, truth row t) in the alignment's current layout,
// nullptr outside the window
__device__ __forceinline__ uint8_t *tie_flag_ptr(const AlnDesc &d, uint8_t *ws, const int32_t *blo_all, int p, int x, int t) {
    if (d.band_w == 16) {   // stripe-transposed records (pr_q16.hip)
        const int2 o = reinterpret_cast<const int2 *>(blo_all + d.blo_off)[t >> 2];
        const int col = x - (p ? o.y : o.x);
        if (col < 0 || col >= 16) return nullptr;
        return ws + d.mat_off[0] + size_t(t >> 2) * 128 + p * 64 + col * 4 + (t & 3);
    }
    if (d.band_w > 0) {     // window rows with per-row origins (pr_band.hip, pr_wide.hip)
        const int col = x - blo_all[d.blo_off + int64_t(p) * d.Lt + t];
        if (col < 0 || col >= d.band_w) return nullptr;
        return ws + d.mat_off[p] + size_t(t) * d.pitch[p] + col;
    }
    return ws + d.mat_off[p] + size_t(t) * d.pitch[p] + x;   // dense
}

#define TIE_U 4    // chunks a wide pass keeps in flight per wave (more does not help: a wave sustains about one scattered access per 10 cycles)
// One workgroup of TIE_NW waves per alignment.  Every pass over a wave's cells (expansion of a wide frontier, the order
// passes, the seeding of the next wave) is strided over the workgroup: entry e of a pass belongs to thread e mod (64 TIE_NW) of
// chunk e / (64 TIE_NW), candidate ids and append positions are functions of e alone (appends: ballot prefix inside a wave,
// per-wave counts through LDS across waves), so the result does not depend on the number of waves.  A narrow frontier
// (<= 64 entries: a chain of dependent steps) is expanded by wave 0 alone, which publishes the queue state to the others.
#define TIE_LA 4   // levels of a narrow stretch a lane looks at per step

template <int TIE_NW>
__global__ void __launch_bounds__(64 * TIE_NW) k_tie_replay(DevBatch B, const AlnDesc *__restrict__ descs,
                                                   TieJob *__restrict__ jobs, int n_jobs, uint8_t *ws,
                                                   const int32_t *__restrict__ blo_all, AlnOut *__restrict__ outs,
                                                   uint32_t *scratch, int32_t *__restrict__ n_overflow,
                                                   int4 *__restrict__ dec, int32_t *__restrict__ n_dec, int dec_cap) {
    constexpr int NT_ = 64 * TIE_NW;
    // positions with more than one allowed swap source (the only cells that can be tied), one bit per position and plane
    __shared__ uint32_t mmask[2][1024];
    __shared__ int lds_ntie, lds_nres;
    __shared__ int lds_cnt[3 * TIE_U][TIE_NW];      // per-wave append counts of a pass (one row per in-flight chunk)
    __shared__ int lds_or[2][TIE_NW];               // per-wave flags of a workgroup-wide "any"
    __shared__ int lds_state[2][6];                 // queue state after a narrow step of wave 0 (double buffered)
    __shared__ uint32_t lds_tot[TIE_NW];
    const int j = blockIdx.x;
    if (j >= n_jobs) return;
    // a replay is one long chain of dependent steps that shares its SIMD with the bulk kernels' issue-bound waves
    __builtin_amdgcn_s_setprio(3);
    const TieJob J = jobs[j];
    const int a = J.a;
    const AlnDesc d = descs[a];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    int or_par = 0, st_par = 0;
    // workgroup-wide OR of a per-thread flag (two barriers; callers are at points every wave reaches)
    auto wg_any = [&](bool x) -> bool {
        const bool w_any = __any(x);
        if (lane == 0) lds_or[or_par][wv] = w_any ? 1 : 0;
        __syncthreads();
        bool r = false;
#pragma unroll
        for (int k = 0; k < TIE_NW; k++) r = r || lds_or[or_par][k];
        or_par ^= 1;
        return r;
    };
    const int Lq = d.Lq, Lr = d.Lr, Lt = d.Lt;
    const int s_fin = outs[a].s;
    const unsigned long long clk0 = wall_clock64();
    int dbg_steps = 0, dbg_cells = 0, dbg_waves = 0, dbg_lastw = -1, dbg_nres0 = 0;
    int dbg_la[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk = clk0;
    auto lap = [&](int k) { const unsigned long long now = wall_clock64(); dbg_t[k] += now - tk; tk = now; };
    const uint8_t *seq0 = B.hap_seq[d.qs] + d.q_off, *seq1 = B.ref_seq + d.r_off;
    const int32_t *ptr0 = B.hap_ptr[d.qs] + d.q_off, *ptr1 = B.ref_ptr[d.qs] + d.r_off;
    const uint8_t *flg0 = B.hap_flag[d.qs] + d.q_off, *flg1 = B.ref_flag[d.qs] + d.r_off;
    const uint8_t *Ts = B.hap_seq[d.ts] + d.t_off, *Tf = B.hap_flag[d.ts] + d.t_off;
    const int4 *cand0 = B.cand_q[d.qs] + d.q_off, *cand1 = B.cand_r[d.qs] + d.r_off;
    const int4 *cand20 = B.cand2_q[d.qs] + d.q_off, *cand21 = B.cand2_r[d.qs] + d.r_off;

    uint32_t *stamp = scratch + J.stamp_off;
    uint32_t *buf = scratch + J.buf_off;
    const int cap = J.cap;
    uint2 *qc = reinterpret_cast<uint2 *>(buf), *qn = reinterpret_cast<uint2 *>(buf + 2 * size_t(cap));
    uint32_t *oc = buf + 4 * size_t(cap), *on = buf + 5 * size_t(cap);
    uint32_t *Fa = buf + 6 * size_t(cap), *Ha = buf + 7 * size_t(cap), *Ka = buf + 8 * size_t(cap);
    uint2 *Ta = reinterpret_cast<uint2 *>(buf + 9 * size_t(cap));    // cells of the current wave that may be tied
    const int tcap = cap / 2;
    unsigned long long *bfirst = reinterpret_cast<unsigned long long *>(scratch) + J.bkt_off;     // [bcap]
    uint32_t *bcount = reinterpret_cast<uint32_t *>(bfirst + J.bcap);                                // [bcap] members per bucket
    uint32_t *blist = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(bcount + J.bcap) + 15) & ~uintptr_t(15));   // [bcap][TIE_BLIST]
    // stamps are stored diagonal-major, [plane][q - t + Lt - 1][t]: a run of matches walks consecutive words (the
    // dense [q][t] order made every step of a run touch another page: the replay was bound by address translation)
    const uint32_t sbase1 = uint32_t(J.dn[0]) * uint32_t(Lt);
    const uint32_t s_dummy = sbase1 + uint32_t(J.dn[1]) * uint32_t(Lt);     // a word behind the grids, for cells outside them
    bool oob = false;
    int oob_p = 0, oob_q = 0, oob_t = 0;
    auto sidx = [&](int p, int q, int t) -> uint32_t {
        const uint32_t dg = uint32_t(q - t - J.dlo[p]);
        if (dg >= uint32_t(J.dn[p])) { if (!oob) { oob_p = p; oob_q = q; oob_t = t; } oob = true; return s_dummy; }
        return (p ? sbase1 : 0u) + dg * uint32_t(Lt) + uint32_t(t);
    };
    const unsigned long long hi_q = (unsigned long long)(2 * d.aln) * 73856093ull + 0x517cc1b727220a95ull;       // dist.h:45
    const unsigned long long hi_r = (unsigned long long)(2 * d.aln + 1) * 73856093ull + 0x517cc1b727220a95ull;

    bool fail = (cap < 2) || Lq > 32768 || Lr > 32768 || J.dn[0] <= 0 || J.dn[1] <= 0;
    for (int p = 0; p < 2 && !fail; p++) {
        const int L = p ? Lr : Lq;
        const int4 *cd = p ? cand1 : cand0;
        for (int x0 = wv * 64; x0 < L; x0 += NT_) {
            const int x = x0 + lane;
            const bool multi = x < L && cd[x].y >= 0;
            const unsigned long long bal = __ballot(multi);
            if (lane < 2) mmask[p][(x0 >> 5) + lane] = uint32_t(bal >> (32 * lane));
        }
    }
    if (tid == 0) { lds_ntie = 0; lds_nres = 0; }
    __syncthreads();
    // the layout the tied cells' bytes are read from: the alignment's current descriptor, or (mode 1) the marking round's
    AlnDesc dl = d;
    uint8_t *wsl = ws;
    if (J.mode) {
        dl.band_w = J.old_band_w; dl.pitch[0] = J.old_pitch[0]; dl.pitch[1] = J.old_pitch[1];
        dl.mat_off[0] = J.old_mat_off[0]; dl.mat_off[1] = J.old_mat_off[1]; dl.blo_off = J.old_blo_off;
        wsl = J.old_arena;
    }
    const int32_t *blo_l = J.mode ? reinterpret_cast<const int32_t *>(J.old_arena) : blo_all;
    auto is_multi = [&](int p, int x) -> bool { return (mmask[p][x >> 5] >> (x & 31)) & 1u; };
    // remember a pushed cell that may be tied (order irrelevant); on overflow the whole wave is scanned instead
    auto note_tie = [&](int p, int x, int t) {
        const int k = atomicAdd(&lds_ntie, 1);
        if (k < tcap) Ta[k] = make_uint2((uint32_t(p) << 31) | uint32_t(x), uint32_t(t));
    };

    // wave 0: the two start cells (dist.cpp:300-305), candidate ids 0 and 1
    if (tid == 0 && !fail) {
        qc[0] = make_uint2(0u, 0u);
        qc[1] = make_uint2(0x80000000u, 0u);
        stamp[sidx(0, 0, 0)] = 0u;
        stamp[sidx(1, 0, 0)] = 1u;
    }
    tie_wait();
    fail = wg_any(fail || oob);
    uint32_t cid = 2;               // next candidate id
    uint32_t n_bkt = TIE_BUCKETS[0];   // prev_wave's bucket count (the first insert allocates 13)
    int bi = 0;
    uint32_t tag = 0;               // order() invocation counter (tags the bucket words)
    int n_cur = 2;

    uint32_t wave_lo = 0;           // candidate ids of the current wave start here
    lap(7);
    for (int w = 0; !fail; w++) {
        // ---- BFS: pop entries, expand, append (dist.cpp:317-381)
        int head = 0;
        while (head < n_cur) {
            const int navail = n_cur - head;
            if (navail > 64) {
                // ---- wide: up to TIE_U chunks of NT_ entries in flight, appended in entry order
                const int n = min(NT_ * TIE_U, navail);
                uint2 x[TIE_U];
                bool ty[TIE_U], tz[TIE_U];
                uint32_t iy[TIE_U], iz[TIE_U];
                int zq[TIE_U];
#pragma unroll
                for (int u = 0; u < TIE_U; u++) {
                    const int e = u * NT_ + tid;
                    x[u] = make_uint2(0u, 0u);
                    if (e < n) x[u] = qc[head + e];
                }
                uint8_t tb[TIE_U], sq[TIE_U];
                int fx[TIE_U], ft[TIE_U];
#pragma unroll
                for (int u = 0; u < TIE_U; u++) {
                    const int p = int(x[u].x >> 31), q = int(x[u].x & 0x7fffffffu), t = int(x[u].y);
                    const bool in = (u * NT_ + tid < n) && t + 1 < Lt;
                    tb[u] = 0; sq[u] = 1; zq[u] = 0; fx[u] = PV; ft[u] = PV;
                    if (in) {
                        tb[u] = Ts[t + 1];
                        ft[u] = Tf[t];
                        if (q + 1 < (p ? Lr : Lq)) sq[u] = (p ? seq1 : seq0)[q + 1]; else sq[u] = 0xff;
                        zq[u] = (p ? ptr1 : ptr0)[q] + 1;
                        fx[u] = (p ? flg1 : flg0)[q];
                    }
                }
                uint8_t so[TIE_U];
#pragma unroll
                for (int u = 0; u < TIE_U; u++) {     // (clamped, unconditional: the four loads go out together)
                    const int p = int(x[u].x >> 31);
                    so[u] = (p ? seq0 : seq1)[min(max(zq[u], 0), (p ? Lq : Lr) - 1)];
                }
#pragma unroll
                for (int u = 0; u < TIE_U; u++) {
                    const int p = int(x[u].x >> 31), q = int(x[u].x & 0x7fffffffu), t = int(x[u].y);
                    const bool in = (u * NT_ + tid < n) && t + 1 < Lt;
                    ty[u] = in && sq[u] == tb[u] && q + 1 < (p ? Lr : Lq);
                    tz[u] = in && fwd_allow(fx[u]) && fwd_allow(ft[u]) && zq[u] >= 0 && zq[u] < (p ? Lq : Lr) && so[u] == tb[u];
                    iy[u] = ty[u] ? sidx(p, q + 1, t + 1) : 0u;
                    iz[u] = tz[u] ? sidx(1 - p, zq[u], t + 1) : 0u;
                }
#pragma unroll
                for (int u = 0; u < TIE_U; u++) {
                    const uint32_t cy = cid + 2u * uint32_t(u * NT_ + tid);
                    if (ty[u]) (void)atomicMin(stamp + iy[u], cy);
                    if (tz[u]) (void)atomicMin(stamp + iz[u], cy + 1u);
                }
                tie_wait();
                __syncthreads();            // every wave's candidate ids are in the stamps
                bool wy[TIE_U], wz[TIE_U];
#pragma unroll
                for (int u = 0; u < TIE_U; u++) {
                    const uint32_t cy = cid + 2u * uint32_t(u * NT_ + tid);
                    wy[u] = ty[u] && tie_ld(stamp + iy[u]) == cy;
                    wz[u] = tz[u] && tie_ld(stamp + iz[u]) == cy + 1u;
                }
                unsigned long long by_[TIE_U], bz_[TIE_U];
#pragma unroll
                for (int u = 0; u < TIE_U; u++) {
                    by_[u] = __ballot(wy[u]); bz_[u] = __ballot(wz[u]);
                    if (lane == 0) lds_cnt[u][wv] = __popcll(by_[u]) + __popcll(bz_[u]);
                }
                if (lane == 0) lds_or[or_par][wv] = __any(oob) ? 1 : 0;
                __syncthreads();
                bool any_oob_ = false;
#pragma unroll
                for (int k = 0; k < TIE_NW; k++) any_oob_ = any_oob_ || lds_or[or_par][k];
                or_par ^= 1;
                int base = n_cur;
#pragma unroll
                for (int u = 0; u < TIE_U; u++) {
                    const unsigned long long by = by_[u], bz = bz_[u];
                    int tot = 0, below = 0;
#pragma unroll
                    for (int k = 0; k < TIE_NW; k++) { const int c_ = lds_cnt[u][k]; tot += c_; below += (k < wv) ? c_ : 0; }
                    if (base + tot > cap) { fail = true; break; }
                    const int p = int(x[u].x >> 31), q = int(x[u].x & 0x7fffffffu), t = int(x[u].y);
                    const int py = base + below + __popcll(by & lt_mask) + __popcll(bz & lt_mask);
                    if (wy[u]) { qc[py] = make_uint2((uint32_t(p) << 31) | uint32_t(q + 1), uint32_t(t + 1)); if (is_multi(p, q + 1)) note_tie(p, q + 1, t + 1); }
                    if (wz[u]) { qc[py + (wy[u] ? 1 : 0)] = make_uint2((uint32_t(1 - p) << 31) | uint32_t(zq[u]), uint32_t(t + 1)); if (is_multi(1 - p, zq[u])) note_tie(1 - p, zq[u], t + 1); }
                    base += tot;
                }
                if (fail || any_oob_) { fail = true; break; }
                tie_wait();
                __syncthreads();            // the appended entries are visible to every wave; lds_cnt may be reused
                n_cur = base;
                head += n;
                dbg_steps++;
                cid += 2u * uint32_t(n);
                if (cid > 0xf0000000u) { fail = true; break; }
                continue;
            }
            const int n = navail;
            // (wave 0 alone; the others wait for the queue state it publishes)
            if (wv == 0) {
                // Narrow stretches (runs of matches behind the last edits; all of wave 0) are chains of levels of a few cells
                // each.  When the chunk is the whole queue, lanes j * n + i look ahead at slot i of level j (the slot's cell
                // moved j steps down its diagonal).  If level 0 reproduces the frontier slot by slot one step down the
                // diagonals, the slots sit on distinct diagonals and every later level's candidates meet each other exactly
                // as level 0's do; what can differ is validity (a mismatch, a variant's edge, the last row) and cells pushed
                // before the batch -- a slot that trails another one down the same diagonal first reaches that slot's own
                // cell, which has its stamp.  So if every slot's candidate pushes of level j are valid / invalid / pushed before
                // the batch exactly as on level 0 and its swap target moves along, level j repeats level 0's outcome; the
                // levels up to the first one that differs are committed at once.  The slots need not share a row (after a
                // wave's seeds, a handful of runs at different rows is the common frontier: 721 of 814 narrow steps of the
                // whole-genome batch's slowest replay).
                // Every lane looks at TIE_LA levels of its slot: levels k * nk + jl, k = 0 .. TIE_LA - 1 (a step costs four or five
                // trips to the L2 whatever it commits, and the loads of the further levels travel with those of the first).
                const int nk = 64 / n;
                const int jl = lane / n, il = lane - jl * n;
                const bool act = lane < nk * n;
                uint2 x = make_uint2(0u, 0u);
                if (act) x = qc[head + il];
                const int p = int(x.x >> 31), q0 = int(x.x & 0x7fffffffu), t0 = int(x.y);
                const int Lme = p ? Lr : Lq, Loth = p ? Lq : Lr;
                const uint8_t *sme = p ? seq1 : seq0, *soth = p ? seq0 : seq1;
                const int32_t *pme = p ? ptr1 : ptr0;
                const uint8_t *fme = p ? flg1 : flg0;
                const bool l0 = lane < n;                 // the lanes of the level actually popped now
                bool ty[TIE_LA], tz[TIE_LA], out[TIE_LA];
                uint32_t iy[TIE_LA], iz[TIE_LA];
                int zq[TIE_LA];
                uint8_t tb[TIE_LA], sq[TIE_LA], so[TIE_LA];
                int fx[TIE_LA], ft[TIE_LA];
#pragma unroll
                for (int k = 0; k < TIE_LA; k++) {       // (clamped, unconditional: the loads go out together)
                    const int q = q0 + k * nk + jl, t = t0 + k * nk + jl;
                    tb[k] = Ts[min(t + 1, Lt - 1)];
                    ft[k] = Tf[min(t, Lt - 1)];
                    sq[k] = sme[min(q + 1, Lme - 1)];
                    zq[k] = pme[min(q, Lme - 1)] + 1;
                    fx[k] = fme[min(q, Lme - 1)];
                }
#pragma unroll
                for (int k = 0; k < TIE_LA; k++) so[k] = soth[min(max(zq[k], 0), Loth - 1)];
#pragma unroll
                for (int k = 0; k < TIE_LA; k++) {
                    const int q = q0 + k * nk + jl, t = t0 + k * nk + jl;
                    const bool in = act && t + 1 < Lt && q < Lme;
                    ty[k] = in && q + 1 < Lme && sq[k] == tb[k];
                    tz[k] = in && fwd_allow(fx[k]) && fwd_allow(ft[k]) && zq[k] >= 0 && zq[k] < Loth && so[k] == tb[k];
                    out[k] = false;
                    iy[k] = 0u; iz[k] = 0u;
                    if (k == 0 && l0) {       // (a cell outside the stamp grids fails the job)
                        if (ty[k]) iy[k] = sidx(p, q + 1, t + 1);
                        if (tz[k]) iz[k] = sidx(1 - p, zq[k], t + 1);
                    } else {                  // (a look-ahead lane whose hypothetical target lies outside only ends the batch of levels)
                        if (ty[k]) { const uint32_t dg = uint32_t(q - t - J.dlo[p]); if (dg >= uint32_t(J.dn[p])) { out[k] = true; iy[k] = s_dummy; } else iy[k] = (p ? sbase1 : 0u) + dg * uint32_t(Lt) + uint32_t(t + 1); }
                        if (tz[k]) { const uint32_t dg = uint32_t(zq[k] - (t + 1) - J.dlo[1 - p]); if (dg >= uint32_t(J.dn[1 - p])) { out[k] = true; iz[k] = s_dummy; } else iz[k] = (p ? 0u : sbase1) + dg * uint32_t(Lt) + uint32_t(t + 1); }
                    }
                }
                const uint32_t cy = cid + 2u * uint32_t(lane), cz = cy + 1u;
                uint32_t oy = TIE_NEVER, oz = TIE_NEVER;
                if (l0 && ty[0]) oy = atomicMin(stamp + iy[0], cy);
                if (l0 && tz[0]) oz = atomicMin(stamp + iz[0], cz);
                // (the look-ahead lanes' stamps are read beside the atomics: a cell both touch is a pushed cell either way, and the
                // level that reaches the slot it belongs to ends the batch before)
                uint32_t vy[TIE_LA], vz[TIE_LA];
#pragma unroll
                for (int k = 0; k < TIE_LA; k++) {
                    vy[k] = 0u; vz[k] = 0u;
                    if (k == 0 && l0) continue;
                    if (ty[k]) vy[k] = tie_ld(stamp + iy[k]);
                    if (tz[k]) vz[k] = tie_ld(stamp + iz[k]);
                }
                tie_wait();
                if (l0) { vy[0] = ty[0] ? tie_ld(stamp + iy[0]) : 0u; vz[0] = tz[0] ? tie_ld(stamp + iz[0]) : 0u; }
                const bool wy = l0 && ty[0] && vy[0] == cy;
                const bool wz = l0 && tz[0] && vz[0] == cz;
                const unsigned long long by = __ballot(wy), bz = __ballot(wz);
                const int tot = __popcll(by) + __popcll(bz);
                int nlev = 1;
                const bool over = n_cur + tot > cap;
                if (over) fail = true;
                const int q = q0, t = t0;                 // (level 0's cell, for the lanes that pop it)
                if (!over) {
                const int ry = __popcll(by & lt_mask) + __popcll(bz & lt_mask);    // rank of this lane's first winner
                const bool one_row = !__any(l0 && int(x.y) != __shfl(int(x.y), 0));
                dbg_la[0]++; if (!one_row) dbg_la[1]++; if (tot != n) dbg_la[2]++;
                if (tot == n && n_cur + nk * n <= cap) {
                    // level 0 must reproduce the frontier, slot by slot, one step down the diagonals (same plane, next
                    // position, next row: the slots then sit on distinct diagonals, whatever their rows)
                    const uint32_t ycode = x.x + 1u, zcode = (uint32_t(1 - p) << 31) | uint32_t(zq[0]);
                    const uint32_t sy = uint32_t(__shfl(int(x.x), ry)) + 1u, sz = uint32_t(__shfl(int(x.x), ry + (wy ? 1 : 0))) + 1u;
                    const int ty_row = __shfl(int(x.y), ry), tz_row = __shfl(int(x.y), ry + (wy ? 1 : 0));
                    const bool stable = !__any((wy && (ycode != sy || t0 != ty_row)) || (wz && (zcode != sz || t0 != tz_row)));
                    if (!stable) dbg_la[3]++;
                    if (stable) {
                        // outcome class of a candidate push: 0 invalid, 1 pushed, 2 same cell as an earlier candidate of the
                        // level, 3 cell pushed before this level
                        const int ky = !ty[0] ? 0 : (wy ? 1 : (oy < cid ? 3 : 2)), kz = !tz[0] ? 0 : (wz ? 1 : (oz < cid ? 3 : 2));
                        const int ky0 = __shfl(ky, il), kz0 = __shfl(kz, il), zq0 = __shfl(zq[0], il);
                        const int ry0 = __shfl(ry, il);
                        const int k_max = (n_cur + TIE_LA * nk * n <= cap) ? TIE_LA : 1;
                        // a look-ahead lane has only read its targets' stamps: untouched (classes 1, 2) or not (class 3)
                        nlev = 0;
                        bool open = true;
#pragma unroll
                        for (int k = 0; k < TIE_LA; k++) {
                            const int L = k * nk + jl;
                            const bool same = !out[k] && (ty[k] == (ky0 != 0)) && (tz[k] == (kz0 != 0)) && (!tz[k] || zq[k] == zq0 + L) &&
                                              (!ty[k] || ((vy[k] == TIE_NEVER) == (ky0 != 3))) && (!tz[k] || ((vz[k] == TIE_NEVER) == (kz0 != 3)));
                            const unsigned long long bad = __ballot(act && !(k == 0 && l0) && !same);
                            if (open && k < k_max) {
                                if (bad) { nlev = k * nk + int(__builtin_ctzll(bad)) / n; open = false; }
                                else nlev = (k + 1) * nk;
                            } else open = false;
                        }
                        if (nlev < 1) nlev = 1;       // (k = 0, jl = 0 is the popped level itself)
                        dbg_la[4]++; dbg_la[5] += nlev;
#pragma unroll
                        for (int k = 0; k < TIE_LA; k++) {
                            const int L = k * nk + jl;
                            if (!act || L < 1 || L >= nlev) continue;
                            const uint32_t c0 = cid + 2u * uint32_t(n) * uint32_t(L) + 2u * uint32_t(il);
                            const int pos = n_cur + L * n + ry0;
                            const int ql = q0 + L, tl = t0 + L;
                            if (ky0 == 1) {
                                stamp[iy[k]] = c0; qc[pos] = make_uint2((uint32_t(p) << 31) | uint32_t(ql + 1), uint32_t(tl + 1));
                                if (is_multi(p, ql + 1)) note_tie(p, ql + 1, tl + 1);
                            }
                            if (kz0 == 1) {
                                stamp[iz[k]] = c0 + 1u; qc[pos + (ky0 == 1 ? 1 : 0)] = make_uint2((uint32_t(1 - p) << 31) | uint32_t(zq[k]), uint32_t(tl + 1));
                                if (is_multi(1 - p, zq[k])) note_tie(1 - p, zq[k], tl + 1);
                            }
                        }
                    }
                }
                if (wy) { qc[n_cur + ry] = make_uint2((uint32_t(p) << 31) | uint32_t(q + 1), uint32_t(t + 1)); if (is_multi(p, q + 1)) note_tie(p, q + 1, t + 1); }
                if (wz) { qc[n_cur + ry + (wy ? 1 : 0)] = make_uint2((uint32_t(1 - p) << 31) | uint32_t(zq[0]), uint32_t(t + 1)); if (is_multi(1 - p, zq[0])) note_tie(1 - p, zq[0], t + 1); }
                }
                tie_wait();
                if (__any(oob)) fail = true;
                if (!fail) {
                    n_cur += tot * nlev;
                    head += n * nlev;
                    cid += 2u * uint32_t(n) * uint32_t(nlev);
                    if (cid > 0xf0000000u) fail = true;
                }
                if (lane == 0) { lds_state[st_par][0] = n_cur; lds_state[st_par][1] = head; lds_state[st_par][2] = int(cid); lds_state[st_par][3] = fail ? 1 : 0; }
            }
            __syncthreads();
            n_cur = lds_state[st_par][0]; head = lds_state[st_par][1]; cid = uint32_t(lds_state[st_par][2]); fail = lds_state[st_par][3] != 0;
            st_par ^= 1;
            dbg_steps++;
            if (fail) break;
        }
        if (fail) break;
        lap(0);
        // ---- tied cells popped in this wave: the allowed source popped last wins (dist.cpp:347,376)
        {
            __syncthreads();
            const int ntie = lds_ntie;
            __syncthreads();
            if (tid == 0) lds_ntie = 0;
            const bool scan_all = ntie > tcap;         // the list overflowed: look at every cell of the wave
            const int nn = scan_all ? n_cur : ntie;
            for (int i0 = 0; i0 < nn; i0 += NT_) {
                const int i = i0 + tid;
                if (i >= nn) continue;
                const uint2 z = scan_all ? qc[i] : Ta[i];
                const int p = int(z.x >> 31), xq = int(z.x & 0x7fffffffu), t = int(z.y);
                if (t == 0) continue;
                uint8_t *fp = nullptr;
                uint32_t f = 0;
                if (J.mode == 2) {  // speculative: no flag bytes yet; the swap edge into z must exist (dist.cpp:338-341)
                    if (!fwd_allow(Tf[t - 1]) || (p ? seq1 : seq0)[xq] != Ts[t]) continue;
                } else {
                    fp = tie_flag_ptr(dl, wsl, blo_l, p, xq, t);
                    if (!fp) continue;
                    f = *fp;
                    if (J.mode) {   // consulted by the backward sweep: on an optimal path, tied, swap edge allowed (dist.cpp:600)
                        if (!(f & F_TIE) || !(f & 31) || !bwd_allow((p ? flg1 : flg0)[xq])) continue;
                    } else if ((f & (F_SWP | F_TIE)) != (F_SWP | F_TIE)) continue;
                }
                const int4 cc = (p ? cand1 : cand0)[xq];
                int4 c2 = make_int4(-1, -1, -1, -1);
                if (cc.w >= 0) c2 = (p ? cand21 : cand20)[xq];
                const int srcs[SWAP_SOURCES_MAX] = {cc.x, cc.y, cc.z, cc.w, c2.x, c2.y, c2.z, c2.w};
                int best = -1, nsrc = 0;
                uint32_t best_st = 0;
#pragma unroll
                for (int k = 0; k < SWAP_SOURCES_MAX; k++) {
                    if (srcs[k] < 0) continue;
                    const uint32_t st = tie_ld(stamp + sidx(1 - p, srcs[k], t - 1));
                    if (st == TIE_NEVER || st < wave_lo) continue;       // not popped in z's wave
                    nsrc++;
                    if (best < 0 || st > best_st) { best = k; best_st = st; }
                }
                if (best < 0 || (J.mode == 2 && nsrc < 2)) continue;     // (speculative: a tie is two writers in one wave)
                if (J.mode) {
                    const int k = atomicAdd(n_dec, 1);
                    if (k < dec_cap) dec[k] = make_int4(a, int(z.x), t, best);
                    atomicAdd(&lds_nres, 1);
                } else {
                    *fp = uint8_t((f & ~uint32_t(F_CHOICE_MASK | F_TIE)) | f_choice_bits(best));
                }
            }
            if (wg_any(oob)) { fail = true; break; }
            if (lds_nres != dbg_nres0) { dbg_nres0 = lds_nres; dbg_lastw = w; }
            if (J.mode == 1 && lds_nres >= J.n_used) { dbg_cells += n_cur; dbg_waves++; break; }   // every consulted tie is decided
        }
        dbg_cells += n_cur; dbg_waves++;
        lap(1);
        if (w >= s_fin) break;

        // ---- iteration order of prev_wave = the n_cur cells of qc in pop order (see the header)
        const int n = n_cur;
        int done = 0;
        bool have_lam = false;      // oc holds the list order of the first `done` elements
        while (true) {
            const int room = int(n_bkt) - done;
            const int take = min(room, n - done);
            const int m = done + take;
            tag++;
            const unsigned long long tagw = (unsigned long long)(~tag) << 32;
            // pass 0: bucket member counts cleared
            for (int b0 = tid; b0 < int(n_bkt); b0 += NT_) bcount[b0] = 0u;
            tie_wait();
            __syncthreads();
            // pass A: bucket of every element, first insertion per bucket, bucket member lists; H cleared
            bool blist_full = false;
            for (int i0 = 0; i0 < m; i0 += NT_ * TIE_U) {
                uint32_t e[TIE_U], bk[TIE_U], slot[TIE_U];
                uint2 c[TIE_U];
#pragma unroll
                for (int u = 0; u < TIE_U; u++) {
                    const int i = i0 + u * NT_ + tid;
                    e[u] = (i < m) ? ((have_lam && i < done) ? oc[i] : uint32_t(i)) : 0u;
                }
#pragma unroll
                for (int u = 0; u < TIE_U; u++) c[u] = (i0 + u * NT_ + tid < m) ? qc[e[u]] : make_uint2(0u, 0u);
#pragma unroll
                for (int u = 0; u < TIE_U; u++) {
                    const int i = i0 + u * NT_ + tid;
                    bk[u] = 0u; slot[u] = 0u;
                    if (i >= m) continue;
                    const unsigned long long hv = ((c[u].x >> 31) ? hi_r : hi_q) ^
                                                  ((unsigned long long)(c[u].x & 0x7fffffffu) * 19349669ull + 0xd15f392b3d4704a2ull) ^
                                                  ((unsigned long long)(c[u].y) * 83492791ull);
                    bk[u] = uint32_t(hv % (unsigned long long)n_bkt);
                    Ka[i] = bk[u];
                    Ha[i] = 0u;
                    (void)atomicMin(bfirst + bk[u], tagw | (unsigned long long)uint32_t(i));
                    slot[u] = atomicAdd(bcount + bk[u], 1u);
                }
#pragma unroll
                for (int u = 0; u < TIE_U; u++) {
                    const int i = i0 + u * NT_ + tid;
                    if (i >= m) continue;
                    if (slot[u] < TIE_BLIST) blist[size_t(bk[u]) * TIE_BLIST + slot[u]] = uint32_t(i); else blist_full = true;
                }
            }
            tie_wait();
            blist_full = wg_any(blist_full);     // (its barrier also ends pass A for every wave)
            lap(2);
            // pass B: F_i = first insertion index of the element's bucket; histogram of F
            for (int i0 = 0; i0 < m; i0 += NT_ * TIE_U) {
                uint32_t b[TIE_U];
                unsigned long long fw[TIE_U];
#pragma unroll
                for (int u = 0; u < TIE_U; u++) b[u] = (i0 + u * NT_ + tid < m) ? Ka[i0 + u * NT_ + tid] : 0u;
#pragma unroll
                for (int u = 0; u < TIE_U; u++)
                    fw[u] = (i0 + u * NT_ + tid < m) ? __hip_atomic_load(bfirst + b[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
                for (int u = 0; u < TIE_U; u++) {
                    const int i = i0 + u * NT_ + tid;
                    if (i >= m) continue;
                    const uint32_t f = uint32_t(fw[u] & 0xffffffffull);
                    Fa[i] = f;
                    (void)atomicAdd(Ha + f, 1u);
                }
            }
            tie_wait();
            __syncthreads();
            lap(3);
            // exclusive suffix sum over H: G[f] = elements in buckets created after f.  Each wave owns a contiguous segment
            // (a multiple of the chunk size): its total first, then the running sums from the segment's top with the totals
            // of the segments above as the start value
            {
                const int CH = 64 * TIE_U;
                const int seg = ((m + TIE_NW * CH - 1) / (TIE_NW * CH)) * CH;       // elements per wave
                const int s_lo = wv * seg, s_hi = min(m, s_lo + seg);
                uint32_t mine = 0;
                for (int i0 = s_lo; i0 < s_hi; i0 += 64) { const int i = i0 + lane; mine += (i < s_hi) ? tie_ld(Ha + i) : 0u; }
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) mine += uint32_t(__shfl_xor(int(mine), o));
                if (lane == 0) lds_tot[wv] = mine;
                __syncthreads();
                uint32_t run = 0;
#pragma unroll
                for (int k = 0; k < TIE_NW; k++) run += (k > wv) ? lds_tot[k] : 0u;
                if (s_hi > s_lo) {
                    const int top = s_lo + ((s_hi - s_lo - 1) / CH) * CH;
                    for (int i0 = top; i0 >= s_lo; i0 -= CH) {
                        uint32_t hcnt[TIE_U];
#pragma unroll
                        for (int u = 0; u < TIE_U; u++) hcnt[u] = (i0 + u * 64 + lane < s_hi) ? tie_ld(Ha + i0 + u * 64 + lane) : 0u;
#pragma unroll
                        for (int u = TIE_U - 1; u >= 0; u--) {
                            const int i = i0 + u * 64 + lane;
                            uint32_t suf = hcnt[u];
#pragma unroll
                            for (int o = 1; o < 64; o <<= 1) {
                                const uint32_t tv = uint32_t(__shfl_down(int(suf), o));
                                if (lane + o < 64) suf += tv;
                            }
                            if (i < s_hi) Ha[i] = run + suf - hcnt[u];
                            run += uint32_t(__shfl(int(suf), 0));
                        }
                    }
                }
            }
            tie_wait();
            __syncthreads();
            lap(4);
            if (!blist_full) {
                // pass C: position = G[F_i] + members of the element's bucket inserted later (from the bucket's member list)
                for (int i0 = 0; i0 < m; i0 += NT_ * TIE_U) {
                    uint32_t f[TIE_U], bk[TIE_U], e[TIE_U], g[TIE_U], cn[TIE_U];
                    uint4 ml[TIE_U], mh[TIE_U];
#pragma unroll
                    for (int u = 0; u < TIE_U; u++) {
                        const int i = i0 + u * NT_ + tid;
                        const bool act = i < m;
                        f[u] = act ? Fa[i] : 0u;
                        bk[u] = act ? Ka[i] : 0u;
                        e[u] = act ? ((have_lam && i < done) ? oc[i] : uint32_t(i)) : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < TIE_U; u++) {
                        const bool act = i0 + u * NT_ + tid < m;
                        g[u] = act ? tie_ld(Ha + f[u]) : 0u;
                        cn[u] = act ? tie_ld(bcount + bk[u]) : 0u;
                        const uint4 *mp = reinterpret_cast<const uint4 *>(blist + size_t(bk[u]) * TIE_BLIST);
                        ml[u] = act ? mp[0] : make_uint4(0u, 0u, 0u, 0u);
                        mh[u] = act ? mp[1] : make_uint4(0u, 0u, 0u, 0u);
                    }
#pragma unroll
                    for (int u = 0; u < TIE_U; u++) {
                        const int i = i0 + u * NT_ + tid;
                        if (i >= m) continue;
                        const uint32_t mem[8] = {ml[u].x, ml[u].y, ml[u].z, ml[u].w, mh[u].x, mh[u].y, mh[u].z, mh[u].w};
                        uint32_t rank = 0;
#pragma unroll
                        for (int k = 0; k < 8; k++) rank += (uint32_t(k) < cn[u] && mem[k] > uint32_t(i)) ? 1u : 0u;
                        on[g[u] + rank] = e[u];
                    }
                }
                tie_wait();
            } else {
                // (a bucket with more than TIE_BLIST members: the same positions from per-bucket counters, chunk by chunk
                // from the back)
                for (int i0 = tid; i0 < m; i0 += NT_) Ka[i0] = 0u;
                tie_wait();
                __syncthreads();
                const int top = (m - 1) & ~63;
                for (int i0 = top; i0 >= 0 && wv == 0; i0 -= 64) {
                    const int i = i0 + lane;
                    const bool act = i < m;
                    const uint32_t f = act ? tie_ld(Fa + i) : 0xffffffffu;
                    const uint32_t e = act ? ((have_lam && i < done) ? oc[i] : uint32_t(i)) : 0u;
                    const uint32_t base = act ? tie_ld(Ka + f) : 0u;
                    const uint32_t g = act ? tie_ld(Ha + f) : 0u;
                    uint32_t intra = 0;
                    for (int l2 = 1; l2 < 64; l2++) {
                        const uint32_t v = uint32_t(__builtin_amdgcn_readlane(int(f), l2));
                        intra += (l2 > lane && v == f) ? 1u : 0u;
                    }
                    if (act) {
                        on[g + base + intra] = e;
                        (void)atomicAdd(Ka + f, 1u);
                    }
                    tie_wait();
                }
            }
            __syncthreads();     // `on` is complete (pass C ended with a wait of every wave's stores)
            lap(5);
            { uint32_t *tmp = oc; oc = on; on = tmp; }
            have_lam = true;
            done = m;
            if (done == n) break;
            if (bi + 1 >= TIE_N_BUCKETS || TIE_BUCKETS[bi + 1] > uint32_t(J.bcap)) { fail = true; break; }
            n_bkt = TIE_BUCKETS[++bi];
        }
        if (fail) break;

        // ---- next wave: INS, DEL, SUB targets of every popped cell, in iteration order (dist.cpp:395-424)
        int n_next = 0;
        wave_lo = cid;
        for (int k0 = 0; k0 < n && !fail; k0 += NT_ * TIE_U) {
            uint32_t e[TIE_U];
            uint2 x[TIE_U];
            bool t0[TIE_U], t1[TIE_U], t2[TIE_U];
            uint32_t j0[TIE_U], j1[TIE_U], j2[TIE_U];
#pragma unroll
            for (int u = 0; u < TIE_U; u++) e[u] = (k0 + u * NT_ + tid < n) ? oc[k0 + u * NT_ + tid] : 0u;
#pragma unroll
            for (int u = 0; u < TIE_U; u++) x[u] = (k0 + u * NT_ + tid < n) ? qc[e[u]] : make_uint2(0u, 0u);
#pragma unroll
            for (int u = 0; u < TIE_U; u++) {
                const bool act = k0 + u * NT_ + tid < n;
                const int p = int(x[u].x >> 31), q = int(x[u].x & 0x7fffffffu), t = int(x[u].y);
                t0[u] = act && q + 1 < (p ? Lr : Lq); t1[u] = act && t + 1 < Lt; t2[u] = t0[u] && t1[u];
                const uint32_t c0 = cid + 3u * uint32_t(u * NT_ + tid);
                j0[u] = t0[u] ? sidx(p, q + 1, t) : 0u; j1[u] = t1[u] ? sidx(p, q, t + 1) : 0u; j2[u] = t2[u] ? sidx(p, q + 1, t + 1) : 0u;
                if (t0[u]) (void)atomicMin(stamp + j0[u], c0);
                if (t1[u]) (void)atomicMin(stamp + j1[u], c0 + 1u);
                if (t2[u]) (void)atomicMin(stamp + j2[u], c0 + 2u);
            }
            tie_wait();
            __syncthreads();
            bool w0[TIE_U], w1[TIE_U], w2[TIE_U];
#pragma unroll
            for (int u = 0; u < TIE_U; u++) {
                const uint32_t c0 = cid + 3u * uint32_t(u * NT_ + tid);
                w0[u] = t0[u] && tie_ld(stamp + j0[u]) == c0;
                w1[u] = t1[u] && tie_ld(stamp + j1[u]) == c0 + 1u;
                w2[u] = t2[u] && tie_ld(stamp + j2[u]) == c0 + 2u;
            }
            unsigned long long b0_[TIE_U], b1_[TIE_U], b2_[TIE_U];
#pragma unroll
            for (int u = 0; u < TIE_U; u++) {
                b0_[u] = __ballot(w0[u]); b1_[u] = __ballot(w1[u]); b2_[u] = __ballot(w2[u]);
                if (lane == 0) lds_cnt[u][wv] = __popcll(b0_[u]) + __popcll(b1_[u]) + __popcll(b2_[u]);
            }
            if (lane == 0) lds_or[or_par][wv] = __any(oob) ? 1 : 0;
            __syncthreads();
            bool any_oob_ = false;
#pragma unroll
            for (int k = 0; k < TIE_NW; k++) any_oob_ = any_oob_ || lds_or[or_par][k];
            or_par ^= 1;
#pragma unroll
            for (int u = 0; u < TIE_U; u++) {
                const unsigned long long b0 = b0_[u], b1 = b1_[u], b2 = b2_[u];
                int tot = 0, below = 0;
#pragma unroll
                for (int k = 0; k < TIE_NW; k++) { const int c_ = lds_cnt[u][k]; tot += c_; below += (k < wv) ? c_ : 0; }
                if (n_next + tot > cap) { fail = true; break; }
                const int p = int(x[u].x >> 31), q = int(x[u].x & 0x7fffffffu), t = int(x[u].y);
                int pos = n_next + below + __popcll(b0 & lt_mask) + __popcll(b1 & lt_mask) + __popcll(b2 & lt_mask);
                if (w0[u]) { qn[pos++] = make_uint2((uint32_t(p) << 31) | uint32_t(q + 1), uint32_t(t)); if (is_multi(p, q + 1)) note_tie(p, q + 1, t); }
                if (w1[u]) { qn[pos++] = make_uint2((uint32_t(p) << 31) | uint32_t(q), uint32_t(t + 1)); if (is_multi(p, q)) note_tie(p, q, t + 1); }
                if (w2[u]) { qn[pos++] = make_uint2((uint32_t(p) << 31) | uint32_t(q + 1), uint32_t(t + 1)); if (is_multi(p, q + 1)) note_tie(p, q + 1, t + 1); }
                n_next += tot;
            }
            cid += 3u * uint32_t(min(NT_ * TIE_U, n - k0));
            if (cid > 0xf0000000u || any_oob_) fail = true;
            tie_wait();
            __syncthreads();            // (lds_cnt is reused by the next chunk)
        }
        if (fail) break;
        lap(6);
        { uint2 *tmp = qc; qc = qn; qn = tmp; }
        n_cur = n_next;
        if (n_cur == 0) { fail = true; break; }   // "Empty queue" (dist.cpp:314): cannot happen for an accepted alignment
    }
    if (fail && tid == 0) atomicAdd(n_overflow, 1);
    const bool any_oob = wg_any(oob);
    if (oob && lane == int(__builtin_ctzll(__ballot(oob)))) { jobs[j].dbg_oob[0] = oob_p; jobs[j].dbg_oob[1] = oob_q; jobs[j].dbg_oob[2] = oob_t; jobs[j].dbg_oob[3] = dbg_waves; }
    if (tid == 0) jobs[j].pad = fail ? (any_oob ? 2 : 1) : 0;       // (debug) why the job gave up: 1 logs / buckets, 2 stamp grid
    if (tid == 0) {
        jobs[j].dbg_us = int32_t((wall_clock64() - clk0) / 100);   // 100 MHz counter
        jobs[j].dbg_steps = dbg_steps; jobs[j].dbg_cells = dbg_cells; jobs[j].dbg_waves = dbg_waves;
        jobs[j].dbg_nres = lds_nres; jobs[j].dbg_lastw = dbg_lastw;
        for (int k = 0; k < 8; k++) jobs[j].dbg_t[k] = int32_t(dbg_t[k] / 100);
        for (int k = 0; k < 6; k++) jobs[j].dbg_la[k] = dbg_la[k];
    }
}

// apply a launch's decision list to the flags the repeated forward sweep has just written (tie-round descriptors)
// (tag: the level tag of the part's tie-round descriptors; decisions of alignments that are not in it are left alone)
__global__ void k_tie_patch(const AlnDesc *__restrict__ descs, const int4 *__restrict__ dec, const int32_t *__restrict__ n_dec,
                            int dec_cap, uint8_t *ws, int tag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(*n_dec, dec_cap)) return;
    const int4 e = dec[i];
    const AlnDesc d = descs[e.x];
    if (d.band_pad != tag) return;
    uint8_t *fp = tie_flag_ptr(d, ws, reinterpret_cast<const int32_t *>(ws), int(uint32_t(e.y) >> 31), e.y & 0x7fffffff, e.z);
    if (!fp) return;
    const uint32_t f = *fp;
    if (!(f & F_SWP)) return;      // (the choice's third bit shares F_SUB's place: only a swap cell has a choice field)
    *fp = uint8_t((f & ~uint32_t(F_CHOICE_MASK | F_TIE)) | f_choice_bits(e.w));
}

#endif
