// pr_kernels.hip -- hand-written gfx950 kernels of the precision/recall path.
//
//   k_prep_cand / k_prep_ins   position attributes derived from the pointer arrays
//   k_fwd<NT,C>   K1: forward two-plane edit-distance sweep      (calc_prec_recall_aln,  dist.cpp:251-443)
//   k_bwd<NT,C>   K2: backward max-TP sweep over the optimal DAG (calc_prec_recall_path, dist.cpp:486-823)
//   k_walk        K3: path walk, sync points, credit sections    (get_prec_recall_path_sync :842-999,
//                                                                 calc_prec_recall :1005-1401)
//   k_ed          K4: edit distance of deferred sync sections    (wf_ed, dist.cpp:1406-1506)
//
// Formulation (DESIGN.md "Kernels"): every edge of the two-plane graph except INS
// consumes one truth base, so both planes are swept row by row over the truth
// index t.  One workgroup owns one (supercluster, alignment); thread `tid` owns C
// consecutive cells of each plane, keeps their distances in registers, mirrors the
// row in LDS for neighbour / swap-edge reads, and the in-row INS chain
//   D[q] = min_k (base[q-k] + k)
// is a prefix-min scan of (base[q]-q) done with wave shuffles plus one LDS hop
// across waves.  Integer work only: no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcfdist_pr.h"
#include "pr_device.h"
#include "pr_scan.h"

__device__ __forceinline__ bool fwd_allow(int f) { return !(f & PV) || (f & PE); }  // dist.cpp:336-339
__device__ __forceinline__ bool bwd_allow(int f) { return !(f & PV) || (f & PB); }  // dist.cpp:600-601

// ---------------------------------------------------------------------------
// K0: position attributes
// ---------------------------------------------------------------------------
// supercluster of every position: a workgroup takes 256 superclusters, whose positions are one contiguous range of the
// array, and searches their 257 offsets in LDS.  blockIdx.y: hap slot 0..3, 4 = ref
__global__ void __launch_bounds__(256) k_prep_scof(DevBatch B) {
    __shared__ int64_t s_off[257];
    const int which = blockIdx.y;
    const int64_t *off = which < 4 ? B.hap_off[which] : B.ref_off;
    int32_t *out = which < 4 ? B.sc_hap[which] : B.sc_ref;
    const int sc0 = blockIdx.x * 256, m = min(256, B.n_sc - sc0);
    for (int i = threadIdx.x; i <= m; i += 256) s_off[i] = off[sc0 + i];
    __syncthreads();
    const int64_t beg = s_off[0], end = s_off[m];
    for (int64_t g = beg + threadIdx.x; g < end; g += 256) {
        int lo = 0, hi = m;  // largest sc with off[sc] <= g
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_off[mid] <= g) lo = mid; else hi = mid;
        }
        out[g] = sc0 + lo;
    }
}

// dir 0: sources = positions of query hap h, destinations in the REF plane   -> cand_r[h]
// dir 1: sources = ref positions (ref->query hap h ptrs), destinations in hap h -> cand_q[h]
__global__ void k_prep_cand(DevBatch B, int h, int dir, int64_t n_src, uint32_t *err) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= n_src) return;
    const int64_t *src_off = dir == 0 ? B.hap_off[h] : B.ref_off;
    const int64_t *dst_off = dir == 0 ? B.ref_off : B.hap_off[h];
    const int32_t *ptr = dir == 0 ? B.hap_ptr[h] : B.ref_ptr[h];
    const uint8_t *flg = dir == 0 ? B.hap_flag[h] : B.ref_flag[h];
    int4 *cand = dir == 0 ? B.cand_r[h] : B.cand_q[h];
    int4 *cand2 = dir == 0 ? B.cand2_r[h] : B.cand2_q[h];
    if (!fwd_allow(flg[g])) return;
    const int sc = (dir == 0 ? B.sc_hap[h] : B.sc_ref)[g];
    const int64_t s0 = src_off[sc];
    const int32_t x = int32_t(g - s0);
    const int32_t p = ptr[g];
    const int64_t dlen = dst_off[sc + 1] - dst_off[sc];
    const int64_t d = int64_t(p) + 1;
    if (d < 1 || d >= dlen) return;
    int rank = 0;
    for (int64_t y = g - 1; y >= s0 && ptr[y] == p; y--)
        if (fwd_allow(flg[y])) rank++;
    if (rank >= SWAP_SOURCES_MAX) { atomicOr(err, VPR_ST_ERR_LIMIT); B.sc_limit[sc] = 1; return; }
    int32_t *slot = reinterpret_cast<int32_t *>((rank < 4 ? cand : cand2) + dst_off[sc] + d);
    slot[rank & 3] = x;
}

__global__ void k_prep_ins(DevBatch B, int slot, int64_t n_src) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= n_src) return;
    if (!(B.hap_flag[slot][g] & PI)) return;
    const int sc = B.sc_hap[slot][g];
    const int32_t p = B.hap_ptr[slot][g];
    if (p >= 0) B.has_ins[slot][B.ref_off[sc] + p] = 1;
}

// ---------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for
// every outstanding HBM flag-row store at each row of the sweep; the rows of one alignment never read
// each other's global stores inside a kernel, so only lgkmcnt has to be drained.  A single-wave
// workgroup needs no s_barrier at all: LDS operations of one wave execute in program order.
// ---------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void lds_barrier() {
    if (NT > 64) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------
// block-wide exclusive prefix-min of two ints (one per plane), tid order.
// wsc: 2*(NT/64) ints of LDS.  Contains one __syncthreads().
// ---------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void block_excl_prefix_min2(int &a, int &b, int32_t *wsc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int ia = a, ib = b;  // inclusive within the wave: a DPP scan (twelve v_min_i32_dpp, no LDS round trips)
    wave_prefix_min2(ia, ib);
    int ea = wave_shr1(ia, D_INF), eb = wave_shr1(ib, D_INF);
    if (NT > 64) {
        if (lane == 63) { wsc[wave * 2] = ia; wsc[wave * 2 + 1] = ib; }
        lds_barrier<NT>();
        for (int w = 0; w < wave; w++) { ea = min(ea, wsc[w * 2]); eb = min(eb, wsc[w * 2 + 1]); }
    } else {
        lds_barrier<NT>();
    }
    a = ea; b = eb;
}

// ---------------------------------------------------------------------------
// K1: forward sweep.  Flag matrices are [plane][t][pitch] bytes in HBM.
// ---------------------------------------------------------------------------
template <int C> struct FlagVec;
template <> struct FlagVec<1> { typedef uint8_t T; };
template <> struct FlagVec<4> { typedef uint32_t T; };
template <> struct FlagVec<8> { typedef uint2 T; };
template <> struct FlagVec<16> { typedef uint4 T; };
template <> struct FlagVec<32> { struct __align__(16) T { uint4 a, b; }; };

// C flag bytes of one thread's chunk held in whole VGPRs (a lone uint8_t would be packed with its
// neighbour into one register, which forces a wait right behind the load)
template <int C> struct FlagReg { typename FlagVec<C>::T v; };
template <> struct FlagReg<1> { uint32_t v; };
template <int C> __device__ __forceinline__ FlagReg<C> load_flags(const uint8_t *p) {
    FlagReg<C> r;
    r.v = *reinterpret_cast<const typename FlagVec<C>::T *>(p);
    return r;
}
template <> __device__ __forceinline__ FlagReg<1> load_flags<1>(const uint8_t *p) {
    FlagReg<1> r;
    r.v = *p;
    return r;
}
template <int C> __device__ __forceinline__ void unpack_flags(const FlagReg<C> &r, uint8_t *out) {
    __builtin_memcpy(out, &r.v, C);
}

template <int NT, int C>
__global__ void __launch_bounds__(NT) k_fwd(DevBatch B, const AlnDesc *__restrict__ descs,
                                            const int32_t *__restrict__ work, uint8_t *__restrict__ ws,
                                            AlnOut *__restrict__ outs, const int32_t *__restrict__ skip) {
    if (skip && skip[blockIdx.x]) return;      // (done by the strip kernels, pr_strip.hip)
    const int a = work[blockIdx.x];
    const AlnDesc d = descs[a];
    const int tid = threadIdx.x, lane = tid & 63;
    const int Lq = d.Lq, Lr = d.Lr, Lt = d.Lt;
    const int PQ = (Lq + C - 1) / C * C, PR = (Lr + C - 1) / C * C;
    extern __shared__ __align__(16) int32_t lds[];
    int32_t *rowD[2] = {lds, lds + PQ + 4};         // +4: keep 16-B alignment, 1 pad
    int32_t *wsc = lds + PQ + PR + 8;

    const uint8_t *seq[2] = {B.hap_seq[d.qs] + d.q_off, B.ref_seq + d.r_off};
    const uint8_t *Ts = B.hap_seq[d.ts] + d.t_off;
    const uint8_t *Tf = B.hap_flag[d.ts] + d.t_off;
    const int4 *cand[2] = {B.cand_q[d.qs] + d.q_off, B.cand_r[d.qs] + d.r_off};
    const int4 *cand2[2] = {B.cand2_q[d.qs] + d.q_off, B.cand2_r[d.qs] + d.r_off};
    const int Lp[2] = {Lq, Lr};
    const int Pp[2] = {PQ, PR};
    uint8_t *mat[2] = {ws + d.mat_off[0], ws + d.mat_off[1]};
    const int q0 = tid * C;

    // per-cell constants
    uint8_t sb[2][C];
    int32_t c0[2][C];
    uint32_t multi[2] = {0, 0};
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int q = q0 + c;
            sb[p][c] = 0xff;
            c0[p][c] = -1;
            if (q < Lp[p]) {
                sb[p][c] = seq[p][q];
                const int4 cc = cand[p][q];
                c0[p][c] = cc.x;
                if (cc.y >= 0) multi[p] |= 1u << c;
            }
        }
    }

    // Force the waits for the loads above to happen here: the compiler otherwise parks its
    // s_waitcnt vmcnt(0) at the first use inside the row loop, where it would also drain the previous
    // row's flag stores every iteration.
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int c = 0; c < C; c++) asm volatile("" ::"v"(uint32_t(sb[p][c])), "v"(c0[p][c]));
    asm volatile("" ::"v"(multi[0]), "v"(multi[1]));

    // row 0: D = q (INS chain from the origin), dist.cpp:300-305,397-405
    int32_t dp[2][C];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        uint8_t fl[C];
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int q = q0 + c;
            dp[p][c] = q;   // phantom cells (q >= L) continue the chain harmlessly
            fl[c] = (q == 0) ? F_MAT : F_INS;
            if (q0 < Pp[p]) rowD[p][q] = q;
        }
        if (q0 < Lp[p]) {
            typename FlagVec<C>::T v;
            __builtin_memcpy(&v, fl, C);
            *reinterpret_cast<typename FlagVec<C>::T *>(mat[p] + q0) = v;
        }
    }
    lds_barrier<NT>();

    // lane l holds truth base and truth flag of row (t & ~63) + l; refilled every 64 rows so the hot loop
    // has no global load (a vmcnt wait there would also wait for the previous rows' flag stores)
    uint32_t tchunk = 0;
    if (lane < Lt) tchunk = uint32_t(Ts[lane]) | (uint32_t(Tf[lane]) << 8);
    uint32_t tlast = 0;   // entry of row (t & ~63) - 1
    for (int t = 1; t < Lt; t++) {
        if ((t & 63) == 0) {
            tlast = __builtin_amdgcn_readlane(tchunk, 63);
            const int tt = t + lane;
            tchunk = 0;
            if (tt < Lt) tchunk = uint32_t(Ts[tt]) | (uint32_t(Tf[tt]) << 8);
        }
        const uint32_t cur = __builtin_amdgcn_readlane(tchunk, t & 63);
        const uint32_t prv = ((t & 63) == 0) ? tlast : uint32_t(__builtin_amdgcn_readlane(tchunk, (t - 1) & 63));
        const uint8_t Tt = cur & 0xff;
        const bool at = fwd_allow(int((prv >> 8) & 0xff));   // truth flag of row t-1, dist.cpp:338-339

        int32_t bv[2][C];     // base - q
        uint8_t mk[2][C];     // flags achieving base
        int cmin[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int32_t *other = rowD[1 - p];
            int diag = (q0 > 0 && q0 < Pp[p]) ? rowD[p][q0 - 1] : D_INF;
            int run = D_INF;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const int q = q0 + c;
                const int up = dp[p][c];
                const bool match = sb[p][c] == Tt;
                const int cm = match ? diag : diag + 1;
                int b = min(cm, up + 1);
                int sw = D_INF;
                int choice = 0;
                bool tie = false;
                if (match && at && c0[p][c] >= 0) {
                    sw = other[c0[p][c]];
                    if (multi[p] & (1u << c)) {
                        const int4 cc = cand[p][q];
                        const int v1 = other[cc.y];
                        if (v1 <= sw) { tie = (v1 == sw); sw = v1; choice = 1; }
                        if (cc.z >= 0) {
                            const int v2 = other[cc.z];
                            if (v2 <= sw) { tie = (v2 == sw); sw = v2; choice = 2; }
                            if (cc.w >= 0) {
                                const int v3 = other[cc.w];
                                if (v3 <= sw) { tie = (v3 == sw); sw = v3; choice = 3; }
                                const int4 c2 = cand2[p][q];
                                const int more[4] = {c2.x, c2.y, c2.z, c2.w};
                                for (int k = 0; k < 4 && more[k] >= 0; k++) {
                                    const int v = other[more[k]];
                                    if (v <= sw) { tie = (v == sw); sw = v; choice = 4 + k; }
                                }
                            }
                        }
                    }
                    b = min(b, sw);
                }
                uint8_t m = 0;
                if (match && diag == b) m |= F_MAT;
                if (diag + 1 == b) m |= F_SUB;
                if (up + 1 == b) m |= F_DEL;
                if (sw == b) m |= F_SWP | f_choice_bits(choice) | (tie ? F_TIE : 0);
                mk[p][c] = m;
                bv[p][c] = b - q;
                run = min(run, b - q);
                diag = up;
            }
            cmin[p] = run;
        }
        int carryQ = cmin[0], carryR = cmin[1];
        block_excl_prefix_min2<NT>(carryQ, carryR, wsc);   // barrier inside: all reads of rowD are done
        const int carry[2] = {carryQ, carryR};
#pragma unroll
        for (int p = 0; p < 2; p++) {
            uint8_t fl[C];
            int run = carry[p];
            int left = carry[p] + q0 - 1;   // D of cell q0-1 in this row
#pragma unroll
            for (int c = 0; c < C; c++) {
                const int q = q0 + c;
                const int nb = bv[p][c];
                uint8_t f = 0;
                if (nb <= run) { run = nb; f = mk[p][c]; }
                const int Dn = run + q;
                if (left + 1 == Dn && q > 0) f |= F_INS;
                fl[c] = f;
                dp[p][c] = Dn;
                left = Dn;
            }
            if (q0 < Lp[p]) {
#pragma unroll
                for (int c = 0; c < C; c++) rowD[p][q0 + c] = dp[p][c];
                typename FlagVec<C>::T v;
                __builtin_memcpy(&v, fl, C);
                *reinterpret_cast<typename FlagVec<C>::T *>(mat[p] + size_t(t) * d.pitch[p] + q0) = v;
            }
        }
        lds_barrier<NT>();
    }
    // end cells
#pragma unroll
    for (int c = 0; c < C; c++) {
        if (q0 + c == Lq - 1) outs[a].dist_q = dp[0][c];
        if (q0 + c == Lr - 1) outs[a].dist_r = dp[1][c];
    }
}

// one thread per alignment: s, end plane (prefer QUERY, dist.cpp:436-439)
__global__ void k_fwd_finish(const int32_t *__restrict__ work, int n, AlnOut *__restrict__ outs, int tag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    AlnOut &o = outs[work[i]];
    o.s = min(o.dist_q, o.dist_r);
    o.end_plane = (o.dist_q <= o.dist_r) ? VPR_PLANE_QUERY : VPR_PLANE_REF;
    o.band_ok = tag;   // 0: dense level; TIE_TAG_BIT: dense level, re-run by a tie round (clears the tie mark)
}

// ---------------------------------------------------------------------------
// K2: backward max-TP sweep.  Reads the forward flags row by row (t descending),
// overwrites them in place with path_ptrs (low 5 bits; 0 = cell not on an optimal path).
// A suffix scan of max-plus maps  x -> max(A, x + B)  carries the in-row INS chain.
// ---------------------------------------------------------------------------
struct MP { int A, B; };   // B < 0  <=> link broken
__device__ __forceinline__ MP mp_compose(MP l, MP r) {   // l applied after r
    MP o;
    if (l.B < 0) return l;
    o.A = max(l.A, r.A + l.B);
    o.B = (r.B < 0) ? -1 : r.B + l.B;
    return o;
}

// Inclusive suffix composition across the block; returns for each thread the value flowing in from
// its right neighbour, i.e. H_{tid+1}(NEG).A.  wsc: 4*(NT/64) ints.  One __syncthreads().
template <int NT>
__device__ __forceinline__ void block_suffix_mp2(MP gq, MP gr, int &inq, int &inr, int32_t *wsc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    MP hq = gq, hr = gr;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        MP tq, tr;
        tq.A = __shfl_down(hq.A, o); tq.B = __shfl_down(hq.B, o);
        tr.A = __shfl_down(hr.A, o); tr.B = __shfl_down(hr.B, o);
        if (lane + o < 64) { hq = mp_compose(hq, tq); hr = mp_compose(hr, tr); }
    }
    // exclusive: function of everything to the right within the wave
    MP eq, er;
    eq.A = __shfl_down(hq.A, 1); eq.B = __shfl_down(hq.B, 1);
    er.A = __shfl_down(hr.A, 1); er.B = __shfl_down(hr.B, 1);
    if (lane == 63) { eq.A = S_NEG; eq.B = 0; er.A = S_NEG; er.B = 0; }   // identity
    if (NT > 64) {
        if (lane == 0) { wsc[wave * 4] = hq.A; wsc[wave * 4 + 1] = hq.B; wsc[wave * 4 + 2] = hr.A; wsc[wave * 4 + 3] = hr.B; }
        lds_barrier<NT>();
        // value entering this wave from the right = fold of waves NT/64-1 .. wave+1 applied to NEG
        int xq = S_NEG, xr = S_NEG;
        for (int w = NT / 64 - 1; w > wave; w--) {
            const int Aq = wsc[w * 4], Bq = wsc[w * 4 + 1], Ar = wsc[w * 4 + 2], Br = wsc[w * 4 + 3];
            xq = (Bq < 0) ? Aq : max(Aq, xq + Bq);
            xr = (Br < 0) ? Ar : max(Ar, xr + Br);
        }
        inq = (eq.B < 0) ? eq.A : max(eq.A, xq + eq.B);
        inr = (er.B < 0) ? er.A : max(er.A, xr + er.B);
    } else {
        lds_barrier<NT>();
        inq = eq.A; inr = er.A;
    }
}

// S16: the score rows in LDS are int16 (a score counts query-variant entries on a path, so it is bounded by the
// supercluster's query variants; the host checks that bound).  4 B instead of 6 B of LDS per cell: the variant the
// planner picks when the int32 rows of an alignment do not fit.  Registers, arithmetic and results are unchanged.
template <int NT, int C, bool S16>
__global__ void __launch_bounds__(NT) k_bwd(DevBatch B, const AlnDesc *__restrict__ descs,
                                            const int32_t *__restrict__ work, uint8_t *__restrict__ ws,
                                            AlnOut *__restrict__ outs, const int32_t *__restrict__ skip) {
    if (skip && skip[blockIdx.x]) return;
    const int a = work[blockIdx.x];
    const AlnDesc d = descs[a];
    const int tid = threadIdx.x;
    const int Lq = d.Lq, Lr = d.Lr, Lt = d.Lt;
    const int PQ = (Lq + C - 1) / C * C, PR = (Lr + C - 1) / C * C;
    const int Lp[2] = {Lq, Lr};
    const int Pp[2] = {PQ, PR};
    extern __shared__ __align__(16) int32_t lds[];
    // score rows (int32) of row t+1, flag rows (bytes) of rows t+1 / t (double buffered)
    // All LDS accesses go through integer offsets from the one extern array so they stay ds_* instructions
    // (runtime-selected pointers degrade to flat_* loads, which wait on vmcnt as well).
    const int SO[2] = {0, PQ + 4};                                    // score rows (element index into lds)
    uint8_t *fbase = reinterpret_cast<uint8_t *>(lds) + ((size_t(PQ + PR + 8) * (S16 ? 2 : 4) + 15) & ~size_t(15));
    int16_t *lds16 = reinterpret_cast<int16_t *>(lds);
    // stored scores are S_NEG or >= 0 (the row loop cleans negatives before storing)
    auto s_ld = [&](int p, int i) -> int {
        if (S16) { const int x = lds16[SO[p] + i]; return x < 0 ? S_NEG : x; }
        return lds[SO[p] + i];
    };
    auto s_st = [&](int p, int i, int v) {
        if (S16) lds16[SO[p] + i] = int16_t(v < 0 ? -1 : v); else lds[SO[p] + i] = v;
    };
    const int FQ = (PQ + 16 + 15) & ~15, FR = (PR + 16 + 15) & ~15;   // bytes per flag row incl. guard
    const int FP[2] = {0, FQ};                                        // plane offset inside one flag buffer
    const int FB = FQ + FR;                                           // bytes per flag buffer (two planes)
    int32_t *wsc = reinterpret_cast<int32_t *>(fbase + 2 * FB);
#define FROW(b, p, i) fbase[(b) * FB + FP[p] + (i)]
    const int32_t *ptr[2] = {B.hap_ptr[d.qs] + d.q_off, B.ref_ptr[d.qs] + d.r_off};
    const uint8_t *pfl[2] = {B.hap_flag[d.qs] + d.q_off, B.ref_flag[d.qs] + d.r_off};
    const int4 *cand[2] = {B.cand_q[d.qs] + d.q_off, B.cand_r[d.qs] + d.r_off};
    const int4 *cand2[2] = {B.cand2_q[d.qs] + d.q_off, B.cand2_r[d.qs] + d.r_off};
    uint8_t *mat[2] = {ws + d.mat_off[0], ws + d.mat_off[1]};
    const int q0 = tid * C;
    const int end_plane = outs[a].end_plane;

    // per-cell constants
    int32_t zq[2][C];       // swap target in the other plane (or -1 if this cell can never be a swap source)
    uint8_t kc[2][C];       // bit0: tp of this cell (QUERY plane only); bits 1-2 and 4: my rank in z's candidate list
                            // (rank_bits); bit3: tp of z (only when z is on the QUERY plane)
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int q = q0 + c;
            zq[p][c] = -1;
            kc[p][c] = 0;
            if (q < Lp[p]) {
                const int pq = ptr[p][q];
                const int fq = pfl[p][q];
                if (p == 0 && q > 0 && ((pq != ptr[0][q - 1] + 1) || (fq & PB))) kc[p][c] |= 1;  // dist.cpp:572-574
                const int z = pq + 1;
                if (fwd_allow(fq) && z > 0 && z < Lp[1 - p] && bwd_allow(pfl[1 - p][z])) {
                    const int4 cc = cand[1 - p][z];
                    int rank = -1;
                    if (cc.x == q) rank = 0; else if (cc.y == q) rank = 1; else if (cc.z == q) rank = 2; else if (cc.w == q) rank = 3;
                    else if (cc.w >= 0) {
                        const int4 c2 = cand2[1 - p][z];
                        if (c2.x == q) rank = 4; else if (c2.y == q) rank = 5; else if (c2.z == q) rank = 6; else if (c2.w == q) rank = 7;
                    }
                    if (rank >= 0) {
                        zq[p][c] = z;
                        kc[p][c] |= uint8_t(rank_bits(rank));
                        if (p == 1) {  // z on the QUERY plane: leaving it scores tp(z), dist.cpp:656-658
                            const int pz = ptr[0][z];
                            if ((pz != ptr[0][z - 1] + 1) || (pfl[0][z] & PB)) kc[p][c] |= 8;
                        }
                    }
                }
            }
        }
    }

    // tp of the QUERY-plane cell right of this chunk (constant over rows)
    int xtp_right = 0;
    if (q0 + C < Lq) {
        const int qn = q0 + C;
        xtp_right = ((ptr[0][qn] != ptr[0][qn - 1] + 1) || (pfl[0][qn] & PB)) ? 1 : 0;
    }

    int32_t sc[2][C];   // scores of row t+1 (S_NEG = unreachable)
    uint8_t f1[2][C];   // forward flags of row t+1
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int c = 0; c < C; c++) { sc[p][c] = S_NEG; f1[p][c] = 0; }
    // LDS rows for "row Lt": nothing reachable
    for (int p = 0; p < 2; p++)
        if (q0 < Pp[p])
#pragma unroll
            for (int c = 0; c < C; c++) { s_st(p, q0 + c, S_NEG); FROW(0, p, q0 + c) = 0; }
    // stage flags of row Lt-1 into flag buffer 1
    uint8_t f0[2][C];
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int c = 0; c < C; c++) f0[p][c] = 0;
        if (q0 < Lp[p]) {
            typename FlagVec<C>::T v = *reinterpret_cast<const typename FlagVec<C>::T *>(mat[p] + size_t(Lt - 1) * d.pitch[p] + q0);
            __builtin_memcpy(f0[p], &v, C);
#pragma unroll
            for (int c = 0; c < C; c++) FROW(1, p, q0 + c) = f0[p][c];
        } else if (q0 < Pp[p]) {
#pragma unroll
            for (int c = 0; c < C; c++) FROW(1, p, q0 + c) = 0;
        }
    }
    if (tid == 0) {   // guard cells right of each row (read by the last thread as "q+1")
        for (int p = 0; p < 2; p++) { s_st(p, Pp[p], S_NEG); FROW(0, p, Pp[p]) = 0; FROW(1, p, Pp[p]) = 0; }
    }
    lds_barrier<NT>();
    uint32_t tie_used = 0;
    // Software pipeline of the flag-row loads: pf holds row t-1, requested at the end of iteration t+1,
    // *before* that iteration's path_ptr stores were issued.  vmcnt retires in order, so waiting for pf
    // (with the younger stores still allowed in flight) never waits for a store younger than one row.
    FlagReg<C> pf[2];
#pragma unroll
    for (int p = 0; p < 2; p++)
        if (Lt >= 2 && q0 < Lp[p]) pf[p] = load_flags<C>(mat[p] + size_t(Lt - 2) * d.pitch[p] + q0);

    for (int t = Lt - 1; t >= 0; t--) {
        const int cur = (Lt - 1 - t + 1) & 1;      // buffer holding row t's forward flags
        const int nxt = cur ^ 1;                   // buffer holding row t+1's forward flags

        int32_t base[2][C];
        uint8_t bm[2][C];
        int8_t lk[2][C];      // INS link from cell c+1 into c: tp value (0/1) or -1 broken
        MP g[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int o = 1 - p;
            // values of the cell to the right of this chunk
            int xs_r = S_NEG; int xf_r = 0, xf0_r = 0, xtp_r = 0;
            if (q0 + C <= Pp[p]) {
                xs_r = s_ld(p, q0 + C);
                xf_r = FROW(nxt, p, q0 + C);
                xf0_r = FROW(cur, p, q0 + C);
            }
            if (p == 0) xtp_r = xtp_right;
            MP G; G.A = S_NEG; G.B = 0;   // identity; composed from the right end of the chunk leftwards
            bool first = true;
#pragma unroll
            for (int c = C - 1; c >= 0; c--) {
                const int q = q0 + c;
                // successor (q+1, t+1): MAT / SUB
                const int xs = (c == C - 1) ? xs_r : sc[p][c + 1];
                const int xf = (c == C - 1) ? xf_r : f1[p][c + 1];
                const int xtp = (c == C - 1) ? xtp_r : (kc[p][c + 1] & 1);
                int best = S_NEG; uint8_t m = 0;
                if (f_diag(xf)) {
                    const int v = xs + xtp;
                    best = v; m = uint8_t(f_diag(xf));
                }
                // successor (q, t+1): DEL
                if (f1[p][c] & F_DEL) {
                    const int v = sc[p][c];
                    if (v > best) { best = v; m = F_DEL; } else if (v == best) m |= F_DEL;
                }
                // swap successor z = (other plane, zq, t+1)
                if (zq[p][c] >= 0) {
                    const int zf = FROW(nxt, o, zq[p][c]);
                    if ((zf & F_SWP) && f_choice_of(zf) == rank_of(kc[p][c])) {
                        const int v = s_ld(o, zq[p][c]) + ((kc[p][c] >> 3) & 1);
                        if (v >= 0 && (zf & F_TIE)) tie_used++;
                        if (v > best) { best = v; m = F_SWP; } else if (v == best) m |= F_SWP;
                    }
                }
                if (t == Lt - 1 && p == end_plane && q == Lp[p] - 1) { best = 0; m = F_MAT; }  // dist.cpp:538-546
                if (q >= Lp[p]) { best = S_NEG; m = 0; }
                base[p][c] = best;
                bm[p][c] = m;
                // INS link from (q+1, t) into (q, t): needs F_INS on row t's forward flags of q+1
                const int xf0 = (c == C - 1) ? xf0_r : f0[p][c + 1];
                const int l = (xf0 & F_INS) ? xtp : -1;
                lk[p][c] = l;
                MP F; F.A = best; F.B = l;
                if (first) { G = F; first = false; } else G = mp_compose(F, G);
            }
            g[p] = G;
        }
        int inq, inr;
        block_suffix_mp2<NT>(g[0], g[1], inq, inr, wsc);   // barrier: all reads of the score rows / flag buffer nxt done
        const int inc[2] = {inq, inr};
        uint8_t out[2][C];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            int prev = inc[p];
#pragma unroll
            for (int c = C - 1; c >= 0; c--) {
                int v = base[p][c];
                uint8_t m = bm[p][c];
                if (lk[p][c] >= 0) {
                    const int w = prev + lk[p][c];
                    if (w > v) { v = w; m = F_INS; } else if (w == v) m |= F_INS;
                }
                if (v < 0) { v = S_NEG; m = 0; }
                sc[p][c] = v;
                out[p][c] = m ? uint8_t(m | (f0[p][c] & F_KEEP)) : uint8_t(0);
                prev = v;
                f1[p][c] = f0[p][c];
            }
            if (q0 < Lp[p]) {
#pragma unroll
                for (int c = 0; c < C; c++) s_st(p, q0 + c, sc[p][c]);
                if (t > 0) {
                    unpack_flags<C>(pf[p], f0[p]);   // row t-1 flags (requested one row ago)
#pragma unroll
                    for (int c = 0; c < C; c++) FROW(nxt, p, q0 + c) = f0[p][c];   // nxt becomes "cur" of row t-1
                }
            }
        }
        // request row t-2 for both planes, then store this row's path_ptrs (loads first, see above)
#pragma unroll
        for (int p = 0; p < 2; p++)
            if (t > 1 && q0 < Lp[p]) pf[p] = load_flags<C>(mat[p] + size_t(t - 2) * d.pitch[p] + q0);
#pragma unroll
        for (int p = 0; p < 2; p++)
            if (q0 < Lp[p]) {
                typename FlagVec<C>::T v;
                __builtin_memcpy(&v, out[p], C);
                *reinterpret_cast<typename FlagVec<C>::T *>(mat[p] + size_t(t) * d.pitch[p] + q0) = v;
            }
        lds_barrier<NT>();
    }
    if (tid == 0) outs[a].beg_plane = (sc[0][0] >= 0) ? VPR_PLANE_QUERY : VPR_PLANE_REF;   // dist.cpp:811-814
    if (tie_used) { atomicOr(&outs[a].status, VPR_ST_SWAP_TIE); outs[a].band_ok = TIE_MARK(0); atomicAdd(&outs[a].n_sec, int(tie_used)); }
}

// ---------------------------------------------------------------------------
#undef FROW

// ---------------------------------------------------------------------------
// K3: walk + sync points + credit sections; one lane per alignment.
// ---------------------------------------------------------------------------
#define ED_INLINE_LONG 256      // longest segment whose edit distance against one of <= 32 bases is computed inside the credit walk
__device__ int small_ed(const uint8_t *a, int m, const uint8_t *b, int n) {
    // plain two-row Levenshtein for short segments (n <= 32); equals wf_ed (dist.cpp:1406-1506)
    int row[33];
    for (int j = 0; j <= n; j++) row[j] = j;
    for (int i = 1; i <= m; i++) {
        int diag = row[0];
        row[0] = i;
        const uint8_t ai = a[i - 1];
        for (int j = 1; j <= n; j++) {
            const int up = row[j];
            const int v = min(min(up + 1, row[j - 1] + 1), diag + (ai != b[j - 1]));
            diag = up;
            row[j] = v;
        }
    }
    return row[n];
}

// Phase B of K3: the backward credit walk of calc_prec_recall (dist.cpp:1035-1400) over a stored path.
// WAVE: every lane runs the same walk (one wavefront per alignment), lane 0 stores.
struct NoFetch { __device__ PathEnt operator()(int64_t) const { return PathEnt{0, 0, 0, 0}; } };
// XF: where the path entries come from when they are not 16-byte records at `path` (k_zero_credit, pr_zl.hip: the
// wave-interleaved walk log of the zero-distance lane kernel); XF()(i) returns entry i
template <bool WAVE, class XF = NoFetch, bool EXT = false>
__device__ __forceinline__ void credit_walk(const DevBatch &B, const AlnDesc &d, AlnOut &O, const int a,
                                            const PathEnt *__restrict__ path, const int64_t n, uint32_t status,
                                            Section *__restrict__ secs, int32_t *const *__restrict__ fp_group,
                                            EdJob *__restrict__ jobs, int32_t *__restrict__ n_jobs, int32_t jobs_cap,
                                            const bool lead, XF xf = XF()) {
    const int qi = 0, ri = 1;
    const int32_t *q2r = B.hap_ptr[d.qs] + d.q_off;
    const int32_t *t2r = B.hap_ptr[d.ts] + d.t_off;
    const int q_size = d.Lq, r_size = d.Lr, t_size = d.Lt;
    (void)qi;
    // The reference keeps three parallel vectors: path (P entries), sync and edits (P+1 entries, the last
    // being the forced final sync with edit=false).  Here entry k carries sync[k], edits[k] for k < n = P;
    // the virtual entry n is (sync=1, edit=0).
    // ---- backward credit walk, dist.cpp:1035-1400
    const uint8_t *Rs = B.ref_seq + d.r_off;
    const uint8_t *Ts = B.hap_seq[d.ts] + d.t_off;
    const int swap = (d.aln == 1 || d.aln == 2);
    const int32_t *qv_pos = B.var_pos[d.qs];
    const int32_t *tv_pos = B.var_pos[d.ts];
    int32_t *fpg = fp_group[d.qs * 2 + swap];   // qs is 0 or 1
    Section *sec = secs + d.sec_off;
    int n_sec = 0;

    int sync_group = 0;
    const int end_hi = O.end_plane;
    int cur_hi = end_hi;
    const int qri_size = (end_hi == qi) ? q_size : r_size;
    int prev_hi = cur_hi, prev_qri = qri_size - 1, prev_ti = t_size - 1;
    int prev_qref = (end_hi == ri) ? r_size - 1 : q2r[q_size - 1];   // reference coordinates of the end cell
    int prev_tref = t2r[t_size - 1];
    int prev_sync_ref_idx = r_size, prev_sync_truth_idx = t_size;
    int query_ed = 0;
    int64_t query_var_ptr = d.qv_end - 1, truth_var_ptr = d.tv_end - 1;
    int query_var_pos = (query_var_ptr >= d.qv_beg) ? qv_pos[query_var_ptr] : 0;
    int truth_var_pos = (truth_var_ptr >= d.tv_beg) ? tv_pos[truth_var_ptr] : 0;
    int64_t prev_query_var_ptr = query_var_ptr, prev_truth_var_ptr = truth_var_ptr;
    int64_t sync_idx = n;   // index into the (P+1)-long sync/edit arrays; entry n is the virtual final one

    // Path entries are consumed in descending order, one per step.  WAVE: the wavefront keeps 64 entries in
    // registers (lane l <-> entry cbase + l, next-lower chunk prefetched) and broadcasts one per step, so the
    // sequential walk of a long alignment pays no HBM round trip per step.  Lane mode: one entry ahead.
    const int lane_ = threadIdx.x & 63;
    int64_t cbase = 0;
    PathEnt ccur = PathEnt{0, 0, 0, 0}, cpre = PathEnt{0, 0, 0, 0};
    if (WAVE) {
        cbase = (n > 0) ? ((n - 1) & ~int64_t(63)) : 0;
        if (cbase + lane_ < n) ccur = path[cbase + lane_];
        if (cbase >= 64) cpre = path[cbase - 64 + lane_];
    }
    // lane mode: eight entries (one 128-byte line of the lane's own stream: path blocks are 128-byte aligned, make_plan) per
    // refill, so a line is fetched once instead of once per entry or per half (thousands of waves in flight thrash L2: PMC
    // showed 5x the path bytes with single entries, 3x with 64-byte refills)
    PathEnt q0 = ccur, q1 = ccur, q2 = ccur, q3 = ccur, q4 = ccur, q5 = ccur, q6 = ccur, q7 = ccur;
    int64_t qbase = int64_t(1) << 60;
    auto fetch = [&](int64_t i) -> PathEnt {
        if constexpr (EXT) {
            return xf(i);
        } else if (WAVE) {
            if (i < cbase) {   // uniform
                cbase -= 64;
                ccur = cpre;
                if (cbase >= 64) cpre = path[cbase - 64 + lane_];
            }
            const int l = int(i - cbase);
            PathEnt e;
            e.a = uint32_t(__builtin_amdgcn_readlane(int(ccur.a), l));
            e.b = uint32_t(__builtin_amdgcn_readlane(int(ccur.b), l));
            e.qref = __builtin_amdgcn_readlane(ccur.qref, l);
            e.tref = __builtin_amdgcn_readlane(ccur.tref, l);
            return e;
        } else {
            if (i < qbase) {
                qbase = i & ~int64_t(7);
                const uint4 *src = reinterpret_cast<const uint4 *>(path + qbase);
                const uint4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4], v5 = src[5], v6 = src[6], v7 = src[7];
                q0 = PathEnt{v0.x, v0.y, int(v0.z), int(v0.w)};
                q1 = PathEnt{v1.x, v1.y, int(v1.z), int(v1.w)};
                q2 = PathEnt{v2.x, v2.y, int(v2.z), int(v2.w)};
                q3 = PathEnt{v3.x, v3.y, int(v3.z), int(v3.w)};
                q4 = PathEnt{v4.x, v4.y, int(v4.z), int(v4.w)};
                q5 = PathEnt{v5.x, v5.y, int(v5.z), int(v5.w)};
                q6 = PathEnt{v6.x, v6.y, int(v6.z), int(v6.w)};
                q7 = PathEnt{v7.x, v7.y, int(v7.z), int(v7.w)};
            }
            const int k = int(i - qbase);
            PathEnt e = q0;
            if (k == 1) e = q1;
            if (k == 2) e = q2;
            if (k == 3) e = q3;
            if (k == 4) e = q4;
            if (k == 5) e = q5;
            if (k == 6) e = q6;
            if (k == 7) e = q7;
            return e;
        }
    };
    uint32_t cur_b = 0;     // .b of path[sync_idx]
    // PathEnt::b bits 28 / 29 (set by the zero-distance walk only, pr_zl.hip): "the reference base at the entry's reference
    // coordinate equals its truth base" / "that bit is valid".  A one-base section below sync entry i is exactly entry i + 1's
    // cell when every step is a diagonal one, so the comparison needs no loads.  eq_next: the bits of entry sync_idx + 1.
    uint32_t eq_next = 0, eq_cur = 0;
    int nx_qref = -2, nx_ti = -2;      // reference coordinate and truth row of entry sync_idx + 1
    // WAVE: the reference / truth bases the sections compare come from 64-base register chunks (lane l <-> base + l), refilled
    // when the walk leaves them -- it moves towards the front, so a chunk ends at the base asked for.  Nearly every step of a
    // long alignment is a sync point with a one-base section: two dependent global loads per step were most of this walk.
    int rs_base = 1 << 30, ts_base = 1 << 30;
    int rs_cur = 0, ts_cur = 0;
    auto rs_at = [&](int i) -> int {
        if constexpr (WAVE) {
            if (i < rs_base || i >= rs_base + 64) {   // uniform
                rs_base = max(0, i - 63);
                const int x = rs_base + lane_;
                rs_cur = x < r_size ? int(Rs[x]) : 0;
            }
            return __builtin_amdgcn_readlane(rs_cur, i - rs_base);
        } else {
            return int(Rs[i]);
        }
    };
    auto ts_at = [&](int i) -> int {
        if constexpr (WAVE) {
            if (i < ts_base || i >= ts_base + 64) {   // uniform
                ts_base = max(0, i - 63);
                const int x = ts_base + lane_;
                ts_cur = x < t_size ? int(Ts[x]) : 0;
            }
            return __builtin_amdgcn_readlane(ts_cur, i - ts_base);
        } else {
            return int(Ts[i]);
        }
    };

    // WAVE: runs of PLAIN entries are passed in one step.  An entry is plain when it is a sync point without an edit, lies one
    // diagonal step below its upper neighbour, the reference base and the truth base of the one-base section between the two
    // are equal, and neither variant pointer moves at it: its iteration then changes nothing but the last sync point (no
    // section, no warning, query_ed stays 0) -- provided the entry above was itself a sync point without an edit (`clean`).
    // Nearly every row of a long alignment is such an entry; the chain of dependent scalar steps per row was this walk.
    // pl_static: per lane (entry cbase + lane of the register chunk), everything but the pointer tests; the top entry of a
    // chunk and the path's last entry (their upper neighbour is not in the chunk) always take the scalar iteration.
    bool clean = false;
    int64_t pl_base = -1;
    bool pl_static = false;
    auto plain_static = [&]() {
        const int qr = ccur.qref, ti = int(ccur.b & 0x0fffffffu);
        const bool have = cbase + lane_ < n;
        const bool eqb = have && qr >= 0 && qr < r_size && ti < t_size && Rs[qr] == Ts[ti];
        const int qr_up = __shfl_down(qr, 1), ti_up = __shfl_down(ti, 1);
        const bool eq_up = __shfl_down(int(eqb), 1) != 0;
        pl_static = have && lane_ < 63 && cbase + lane_ + 1 < n && (ccur.b >> 31) && !((ccur.b >> 30) & 1u) &&
                    qr_up == qr + 1 && ti_up == ti + 1 && eq_up;
        pl_base = cbase;
    };
    while (sync_idx >= 0) {
        if constexpr (WAVE && !EXT) {
            if (clean && sync_idx < n) {      // (uniform)
                if (pl_base != cbase) plain_static();
                const int l = int(sync_idx - cbase);
                const bool plain = pl_static && ccur.qref >= query_var_pos && ccur.tref >= truth_var_pos;
                // the run of plain entries from lane l downwards
                const unsigned long long below = (l == 63) ? ~0ull : ((2ull << l) - 1ull);
                const unsigned long long stop = ~__ballot(plain) & below;
                const int run = stop ? l - (63 - __builtin_clzll(stop)) : l + 1;
                if (run > 0) {
                    const int lj = l - run + 1;      // the run's last entry: afterwards as if it had just been processed
                    prev_qref = __builtin_amdgcn_readlane(ccur.qref, lj);
                    prev_tref = __builtin_amdgcn_readlane(ccur.tref, lj);
                    cur_b = uint32_t(__builtin_amdgcn_readlane(int(ccur.b), lj));
                    prev_hi = int(uint32_t(__builtin_amdgcn_readlane(int(ccur.a), lj)) >> 31);
                    prev_ti = int(cur_b & 0x0fffffffu);
                    prev_sync_ref_idx = prev_qref + 1;
                    prev_sync_truth_idx = prev_ti + 1;
                    eq_cur = (cur_b >> 28) & 3u;
                    sync_idx -= run;
                    if (sync_idx < 0) break;
                    cur_hi = prev_hi;
                    const PathEnt e = fetch(sync_idx);
                    cur_b = e.b;
                    eq_next = eq_cur; eq_cur = (e.b >> 28) & 3u;
                    nx_qref = prev_qref; nx_ti = prev_ti;
                    prev_qri = int(e.a & 0x7fffffffu);
                    prev_hi = int(e.a >> 31);
                    prev_ti = int(e.b & 0x0fffffffu);
                    prev_qref = e.qref;
                    prev_tref = e.tref;
                    continue;
                }
            }
        }
        if constexpr (WAVE && !EXT) {
            // ... and so are runs of entries that are NOT sync points (round 6: an SV that one call set lacks is thousands of edit
            // steps in a row): where neither variant pointer moves, such an entry's iteration only counts its edit.
            if (sync_idx < n && sync_idx >= cbase) {      // (uniform; the entry's chunk is in registers: fetch() put it there)
                const int l = int(sync_idx - cbase);
                const bool ns = cbase + lane_ < n && !(ccur.b >> 31) && ccur.qref >= query_var_pos && ccur.tref >= truth_var_pos;
                const unsigned long long below = (l == 63) ? ~0ull : ((2ull << l) - 1ull);
                const unsigned long long stop = ~__ballot(ns) & below;
                const int run = stop ? l - (63 - __builtin_clzll(stop)) : l + 1;
                if (run >= 2) {
                    const int lj = l - run + 1;
                    const unsigned long long in_run = below & ~((1ull << lj) - 1ull);
                    query_ed += __popcll(__ballot(ns && ((ccur.b >> 30) & 1u)) & in_run);
                    clean = false;
                    prev_qref = __builtin_amdgcn_readlane(ccur.qref, lj);
                    prev_tref = __builtin_amdgcn_readlane(ccur.tref, lj);
                    cur_b = uint32_t(__builtin_amdgcn_readlane(int(ccur.b), lj));
                    prev_hi = int(uint32_t(__builtin_amdgcn_readlane(int(ccur.a), lj)) >> 31);
                    prev_ti = int(cur_b & 0x0fffffffu);
                    eq_cur = (cur_b >> 28) & 3u;
                    sync_idx -= run;
                    if (sync_idx < 0) break;
                    cur_hi = prev_hi;
                    const PathEnt e = fetch(sync_idx);
                    cur_b = e.b;
                    eq_next = eq_cur; eq_cur = (e.b >> 28) & 3u;
                    nx_qref = prev_qref; nx_ti = prev_ti;
                    prev_qri = int(e.a & 0x7fffffffu);
                    prev_hi = int(e.a >> 31);
                    prev_ti = int(e.b & 0x0fffffffu);
                    prev_qref = e.qref;
                    prev_tref = e.tref;
                    continue;
                }
            }
        }
        const int query_ref_pos = prev_qref;
        while (query_ref_pos < query_var_pos && query_var_ptr >= d.qv_beg) {
            if (cur_hi == ri) {   // FP: passed on the REF plane, dist.cpp:1157-1168
                if (lead) fpg[query_var_ptr] = sync_group;
                sync_group++;
            }
            query_var_ptr--;
            query_var_pos = (query_var_ptr < d.qv_beg) ? -1 : qv_pos[query_var_ptr];
        }
        const int truth_ref_pos = prev_tref;
        while (truth_ref_pos < truth_var_pos && truth_var_ptr >= d.tv_beg) {
            truth_var_ptr--;
            truth_var_pos = (truth_var_ptr < d.tv_beg) ? -1 : tv_pos[truth_var_ptr];
        }
        const bool is_sync = (sync_idx == n) ? true : bool(cur_b >> 31);
        const int is_edit = (sync_idx == n) ? 0 : int((cur_b >> 30) & 1);
        if (is_sync) {
            const int sync_ref_idx = prev_qref + 1;
            const int sync_truth_idx = prev_ti + 1;
            int rl = prev_sync_ref_idx - sync_ref_idx, tl = prev_sync_truth_idx - sync_truth_idx;
            if (rl < 0 || rl > r_size - sync_ref_idx) rl = r_size - sync_ref_idx;   // std::string::substr clamp
            if (tl < 0 || tl > t_size - sync_truth_idx) tl = t_size - sync_truth_idx;
            const bool has_q = query_var_ptr != prev_query_var_ptr;
            const bool has_t = truth_var_ptr != prev_truth_var_ptr;
            int ref_ed = -1;
            bool deferred = false;
            if (rl == 0) ref_ed = tl;
            else if (tl == 0) ref_ed = rl;
            else if (rl == 1 && tl == 1)
                ref_ed = ((eq_next & 2u) && nx_qref == sync_ref_idx && nx_ti == sync_truth_idx) ? int(!(eq_next & 1u))
                                                                                                : int(rs_at(sync_ref_idx) != ts_at(sync_truth_idx));
            else if (!has_q && !has_t && rl == tl) {
                bool same = true;
                for (int k = rl - 1; k >= 0 && same; k--) same = (rs_at(sync_ref_idx + k) == ts_at(sync_truth_idx + k));
                if (same) ref_ed = 0;
            }
            if (ref_ed < 0) {
                // (inline while it is cheap: the short side up to 32, the long side up to ED_INLINE_LONG.  The inline loop is
                // long x short sequential steps of one lane: a section of 9 980 reference bases against 23 truth bases -- an SV the
                // truth lacks, joint_synth -- held its wavefront for 8 ms; deferred, the bit-parallel kernel takes microseconds)
                if (tl <= 32 && rl <= ED_INLINE_LONG) ref_ed = small_ed(Rs + sync_ref_idx, rl, Ts + sync_truth_idx, tl);
                else if (rl <= 32 && tl <= ED_INLINE_LONG) ref_ed = small_ed(Ts + sync_truth_idx, tl, Rs + sync_ref_idx, rl);
                else deferred = true;
            }
            if (has_q || has_t || deferred) {
                if (n_sec < d.sec_cap) {
                    Section S;
                    S.q_lo = int32_t(query_var_ptr); S.q_hi = int32_t(prev_query_var_ptr);
                    S.t_lo = int32_t(truth_var_ptr); S.t_hi = int32_t(prev_truth_var_ptr);
                    S.sync_group = sync_group; S.query_ed = query_ed; S.ref_ed = ref_ed;
                    S.flags = (deferred ? SEC_DEFERRED : 0) | ((has_q || has_t) ? 0 : SEC_NOVAR);
                    if (deferred && lead) {
                        const int32_t j = atomicAdd(n_jobs, 1);
                        if (j < jobs_cap) jobs[j] = EdJob{a, n_sec, sync_ref_idx, rl, sync_truth_idx, tl};
                        else status |= VPR_ST_ERR_LIMIT;
                    }
                    if (lead) sec[n_sec] = S;
                    n_sec++;
                } else status |= VPR_ST_ERR_LIMIT;
            } else {
                // no variants, distance known: only the "should never happen" checks, dist.cpp:1203-1214
                if (ref_ed != 0) status |= VPR_ST_WARN_REF_ED;
                if (query_ed != ref_ed) status |= VPR_ST_WARN_QUERY_ED;
                if (query_ed > ref_ed) status |= VPR_ST_WARN_EXCEEDS;
            }
            if (has_q || has_t) sync_group++;
            prev_query_var_ptr = query_var_ptr;
            prev_truth_var_ptr = truth_var_ptr;
            prev_sync_ref_idx = sync_ref_idx;
            prev_sync_truth_idx = sync_truth_idx;
            query_ed = 0;
        }
        query_ed += is_edit;
        clean = is_sync && !is_edit;
        sync_idx--;
        if (sync_idx < 0) break;
        cur_hi = prev_hi;
        const PathEnt e = fetch(sync_idx);
        cur_b = e.b;
        eq_next = eq_cur; eq_cur = (e.b >> 28) & 3u;
        nx_qref = prev_qref; nx_ti = prev_ti;
        prev_qri = int(e.a & 0x7fffffffu);
        prev_hi = int(e.a >> 31);
        prev_ti = int(e.b & 0x0fffffffu);
        prev_qref = e.qref;
        prev_tref = e.tref;
        // (dist.cpp re-reads the two variant positions here; they only change in the loops above)
    }
    (void)prev_qri; (void)prev_ti;
    if (lead) {
        O.n_sec = n_sec;
        if (status) atomicOr(&O.status, status);
    }
}

// WAVE = false: one lane per alignment (64 alignments per wave), flags read straight from HBM.
// WAVE = true : one wavefront per alignment for the long ones; every lane runs the same (uniform) walk, the
//               window rows of the path_ptr matrices are staged through an LDS tile with coalesced 16-byte
//               loads, and lane 0 does the stores.  The walk is a chain of dependent steps, so the long
//               alignments are latency-bound: an LDS hit per step instead of an HBM round trip.
#define WALK_TR 32          // rows per LDS tile
#define WALK_TW 256         // widest window staged (band C = 1 or 4)
template <bool WAVE>
__global__ void __launch_bounds__(64) k_walk(DevBatch B, const AlnDesc *__restrict__ descs,
                       const int32_t *__restrict__ work, int n_work,
                       const uint8_t *__restrict__ ws, const int32_t *__restrict__ blo_all,
                       AlnOut *__restrict__ outs,
                       PathEnt *__restrict__ paths, Section *__restrict__ secs,
                       int32_t *const *__restrict__ fp_group /* [2 query haps * 2 swaps] */,
                       EdJob *__restrict__ jobs, int32_t *__restrict__ n_jobs, int32_t jobs_cap, int my_w, int ok_tag) {
    __shared__ __align__(16) uint8_t tile[WAVE ? 2 * WALK_TR * WALK_TW : 16];
    __shared__ int32_t tblo[WAVE ? 2 * WALK_TR : 2];
    const int wi = WAVE ? int(blockIdx.x) : int(blockIdx.x * blockDim.x + threadIdx.x);
    if (wi >= n_work) return;
    if (WAVE) __builtin_amdgcn_s_setprio(2);   // one wave per long alignment: a latency chain
    const int lane = threadIdx.x & 63;
    const bool lead = !WAVE || lane == 0;
    const int a = work[wi];
    const AlnDesc d = descs[a];
    AlnOut &O = outs[a];
    // my_w: level tag of the descriptors this launch may touch (0: dense), ok_tag: the tag its round's accept
    // test stores in band_ok (they differ only when a round re-runs alignments in place); see k_fwd_band_finish
    if (d.band_pad != my_w || (my_w > 0 ? O.band_ok != ok_tag : O.band_ok < 0)) return;   // (< 0: left to the tie pass)
    const int qi = 0, ri = 1;   // planes
    const int32_t *q2r = B.hap_ptr[d.qs] + d.q_off;
    const uint8_t *qfl = B.hap_flag[d.qs] + d.q_off;
    const int32_t *t2r = B.hap_ptr[d.ts] + d.t_off;
    const uint8_t *tfl = B.hap_flag[d.ts] + d.t_off;
    const int32_t *r2q = B.ref_ptr[d.qs] + d.r_off;
    const uint8_t *insQ = B.has_ins[d.qs] + d.r_off;
    const uint8_t *insT = B.has_ins[d.ts] + d.r_off;
    const uint8_t *mat[2] = {ws + d.mat_off[0], ws + d.mat_off[1]};
    const int q_size = d.Lq, r_size = d.Lr, t_size = d.Lt;
    const bool banded = d.band_w > 0;   // rows hold cells [blo[t], blo[t]+band_w) of each plane
    const bool wide = WAVE && (!banded || d.band_w > WALK_TW);
    const int32_t *blo = blo_all + d.blo_off;
    PathEnt *path = paths + d.path_off;
    uint32_t status = 0;
    int tile_t0 = -(1 << 30);   // first truth row held by the LDS tile (WAVE only)

    // WAVE: the packed walk constants (k_prep_wk) of 64 truth rows and of 64 columns of each plane are kept in
    // registers (lane l <-> element base + l) and broadcast per step, so a step of the sequential walk reads no
    // HBM: rows advance monotonically (next chunk prefetched), a column chunk is re-centred on a miss.
    const int2 *wq_ = B.wk_q[d.qs] + d.q_off, *wr_ = B.wk_r[d.qs] + d.r_off, *wt_ = B.wk_t[d.ts - 2] + d.t_off;
    const int insmask = ((1 << d.qs) | (1 << d.ts)) << 8;
    int rbase = 0, cbase_[2] = {-(1 << 30), -(1 << 30)};
    int2 rcur = make_int2(0, 0), rnxt = make_int2(0, 0), ccol[2] = {make_int2(0, 0), make_int2(0, 0)};
    if (WAVE) {
        if (lane < t_size) rcur = wt_[lane];
        if (64 + lane < t_size) rnxt = wt_[64 + lane];
    }
    auto row_get = [&](int t) -> int2 {
        if (t >= rbase + 64) {   // uniform
            if (t < rbase + 128) { rbase += 64; rcur = rnxt; }
            else {               // (a diagonal run skipped a chunk)
                rbase = t & ~63;
                rcur = make_int2(0, 0);
                if (rbase + lane < t_size) rcur = wt_[rbase + lane];
            }
            rnxt = make_int2(0, 0);
            if (rbase + 64 + lane < t_size) rnxt = wt_[rbase + 64 + lane];
        }
        return make_int2(__builtin_amdgcn_readlane(rcur.x, t - rbase), __builtin_amdgcn_readlane(rcur.y, t - rbase));
    };
    auto col_get = [&](int pl, int x) -> int2 {
        if (x < cbase_[pl] || x >= cbase_[pl] + 64) {   // uniform
            cbase_[pl] = max(0, x - 8);
            const int xx = cbase_[pl] + lane;
            ccol[pl] = make_int2(0, 0);
            if (pl == 0) { if (xx < q_size) ccol[0] = wq_[xx]; }
            else { if (xx < r_size) ccol[1] = wr_[xx]; }
        }
        const int l = x - cbase_[pl];
        return pl == 0 ? make_int2(__builtin_amdgcn_readlane(ccol[0].x, l), __builtin_amdgcn_readlane(ccol[0].y, l))
                       : make_int2(__builtin_amdgcn_readlane(ccol[1].x, l), __builtin_amdgcn_readlane(ccol[1].y, l));
    };

    // ---- forward walk, dist.cpp:865-998
    int hi = O.beg_plane, qri = 0, ti = 0;
    int64_t n = 0;
    if (lead) path[0] = PathEnt{uint32_t(qri) | (uint32_t(hi) << 31), uint32_t(ti) | (1u << 31),
                                (hi == ri) ? 0 : q2r[0], t2r[0]};   // sync, no edit
    n = 1;
    bool ok = true;
    int skip_run = 0;       // WAVE: steps until the next attempt at a run
    int last_mv = 0;        // WAVE: move of the last single step
    int singles = 0;        // WAVE: single steps since the last run
    while ((hi == ri && qri < r_size - 1) || (hi == qi && qri < q_size - 1) || ti < t_size - 1) {
        if (WAVE) {
            // DIAGONAL RUNS.  Nearly all of a long alignment's walk is MAT steps down one diagonal of one plane.  Lane l looks at
            // the cell l steps down that diagonal (one scattered byte per lane: ONE round trip for up to 64 steps, where the
            // step-by-step walk pays a dependent lookup per step); the leading lanes whose cell takes MAT by the walk's priority
            // (dist.cpp:907-935: on the REF plane only without a swap) are a run, and the run's path entries -- the cells
            // ENTERED by its steps, with their sync flags (dist.cpp:949-968) -- are written side by side.
            // INS / DEL RUNS (round 6).  An alignment of a call set WITHOUT an SV against a haplotype with it walks thousands of INS
            // steps along one row, or of DEL steps down one column (configs[2] / configs[3]: 10 000 edits in a row, 3.6 us each
            // step by step).  Same idea: lane l looks at the cell l steps along the row (one coalesced load) or down the column,
            // the leading lanes whose cell takes that move by the walk's priority are a run.  Which kind of run is tried follows the
            // last single step's move.
            const int lim = min(min((hi == ri ? r_size : q_size) - 1 - qri, t_size - 1 - ti), 64);
            const int lim_i = min((hi == ri ? r_size : q_size) - 1 - qri, 64), lim_d = min(t_size - 1 - ti, 64);
            if (skip_run > 0) skip_run--;
            else if (last_mv == F_INS || last_mv == F_DEL) {
                const bool ins = last_mv == F_INS;
                const int lm = ins ? lim_i : lim_d;
                int run = 0;
                if (lm >= 2) {
                    const int tq = ins ? qri + lane : qri, tt = ins ? ti : ti + lane;
                    bool plain = false;
                    if (lane < lm) {
                        int colr = tq;
                        bool in_w = true;
                        if (banded) { colr = tq - blo[hi * t_size + tt]; in_w = colr >= 0 && colr < d.band_w; }
                        if (in_w) {
                            const int pb = mat[hi][size_t(tt) * d.pitch[hi] + colr] & 31;
                            const bool higher = (pb & (F_MAT | F_SUB)) || (hi == ri && (pb & F_SWP));
                            plain = !higher && (ins ? (pb & F_INS) != 0 : ((pb & F_DEL) != 0 && !(pb & F_INS)));
                        }
                    }
                    const unsigned long long stop = ~__ballot(plain);
                    run = stop ? int(__builtin_ctzll(stop)) : 64;
                }
                if (run >= 2 && n + run <= d.path_cap) {
                    if (lane < run) {       // the cells ENTERED by the run's steps: edits, never sync points (dist.cpp:949-968)
                        const int xq = ins ? qri + lane + 1 : qri, xt = ins ? ti : ti + lane + 1;
                        const int tr = wt_[xt].x;
                        const int qr = (hi == ri) ? xq : wq_[xq].x;
                        path[n + lane] = PathEnt{uint32_t(xq) | (uint32_t(hi) << 31), uint32_t(xt) | (1u << 30), qr, tr};
                    }
                    n += run;
                    if (ins) qri += run; else ti += run;
                    singles = 0;
                    continue;
                }
                skip_run = 3;
            }
            else if (lim >= 2) {
                // (MAT or SUB, each cell by the walk's priority: a stretch the two call sets spell differently is thousands of SUB
                // steps down one diagonal -- 1.2 - 1.8 us each step by step, 11.6 of a 12 ms walk launch on sv_synth)
                const int tq = qri + lane, tt = ti + lane;
                bool plain = false, sub = false;
                if (lane < lim) {
                    int colr = tq;
                    bool in_w = true;
                    if (banded) { colr = tq - blo[hi * t_size + tt]; in_w = colr >= 0 && colr < d.band_w; }
                    if (in_w) {
                        const int pb = mat[hi][size_t(tt) * d.pitch[hi] + colr] & 31;
                        const bool swp_first = hi == ri && (pb & F_SWP);
                        sub = !swp_first && !(pb & F_MAT) && (pb & F_SUB);
                        plain = !swp_first && (pb & (F_MAT | F_SUB));
                    }
                }
                const unsigned long long stop = ~__ballot(plain);
                const int run = stop ? int(__builtin_ctzll(stop)) : 64;
                if (run >= 2 && n + run <= d.path_cap) {
                    if (lane < run) {
                        const int xq = qri + lane + 1, xt = ti + lane + 1;
                        const int2 rw = wt_[xt];
                        const int2 cw = (hi == ri) ? wr_[xq] : wq_[xq];
                        const int tflv = rw.y & 0xff, tr = rw.x;
                        const int qflv = cw.y & 0xff, qr = (hi == ri) ? xq : cw.x;
                        const bool ins_loc = ((rw.y | cw.y) & insmask) != 0;
                        const bool in_t = (tflv & PV) && !(tflv & PB);
                        const bool in_q = hi == qi && (qflv & PV) && !(qflv & PB);
                        const bool sync = !in_t && !in_q && !ins_loc && tr == qr;
                        path[n + lane] = PathEnt{uint32_t(xq) | (uint32_t(hi) << 31), uint32_t(xt) | (uint32_t(sync) << 31) | (uint32_t(sub) << 30), qr, tr};
                    }
                    n += run; qri += run; ti += run;
                    singles = 0;
                    continue;
                }
                skip_run = 3;
            }
        }
        int p;
        if (WAVE && wide) {
            // dense rows / windows wider than the tile: the tile holds WALK_TW columns of each plane around the walk's
            // position (the other plane's around the position it maps to) and is staged again when the walk leaves it
            const int r0 = ti - tile_t0;
            int colw = (r0 >= 0 && r0 < WALK_TR) ? qri - tblo[hi * WALK_TR] : -1;
            // (a tile is 2 x 32 rows of loads, ~30 us: worth it for a stretch of single steps, not for the one or two between
            // two runs -- a walk that is mostly runs left its tile with every run and staged a new one for the step behind it:
            // the first single steps behind a run read their byte straight from the matrix)
            if ((colw < 0 || colw >= WALK_TW) && singles < 4) {
                const int org = banded ? blo[hi * t_size + ti] : 0;
                const int wdt = banded ? d.band_w : (hi == 0 ? q_size : r_size);
                const int cx = qri - org;
                if (cx < 0 || cx >= wdt) { status |= VPR_ST_ERR_NO_PTR; ok = false; break; }
                p = mat[hi][size_t(ti) * d.pitch[hi] + cx] & 31;
            } else {
            if (colw < 0 || colw >= WALK_TW) {
                tile_t0 = ti;
                const int rows = min(WALK_TR, t_size - ti);
                const int other = col_get(hi, qri).x;       // q2r / r2q of the position
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    const int pos = (pl == hi) ? qri : other;
                    int c0 = max(0, min(pos - 64, (pl == 0 ? q_size : r_size) - WALK_TW));
                    // (all rows of a dense matrix are requested before the first is used, one aligned word per lane and row: row by
                    // row with four byte loads and a wait in front of every LDS store a tile took ~30 us -- 113 tiles of the slowest
                    // dense walk of sv_synth were 3.4 of its 8.4 ms.  Window rows (origins per row) keep the byte loads, eight rows at a time)
                    const int wdt = banded ? d.band_w : (pl == 0 ? q_size : r_size);
                    if (!banded && (d.pitch[pl] & 3) == 0 && (reinterpret_cast<uintptr_t>(mat[pl]) & 3) == 0) {
                        // (an aligned origin: up to three columns further left -- the tile still ends behind the plane's last column)
                        c0 = max(0, min(pos - 64, (pl == 0 ? q_size : r_size) - (WALK_TW - 4))) & ~3;
                        uint32_t v[WALK_TR];
                        const int x = c0 + lane * 4;
#pragma unroll
                        for (int r = 0; r < WALK_TR; r++) {
                            v[r] = 0;
                            if (r < rows && x < d.pitch[pl]) v[r] = *reinterpret_cast<const uint32_t *>(mat[pl] + size_t(ti + r) * d.pitch[pl] + x);
                        }
                        const uint32_t keep = x + 3 < wdt ? 0xffffffffu : (x >= wdt ? 0u : (0xffffffffu >> (8 * (4 - (wdt - x)))));
#pragma unroll
                        for (int r = 0; r < WALK_TR; r++)
                            if (r < rows) *reinterpret_cast<uint32_t *>(tile + (pl * WALK_TR + r) * WALK_TW + lane * 4) = v[r] & keep;
                    } else {
                        for (int r0 = 0; r0 < rows; r0 += 8) {
                            int org[8];
                            uint32_t v[8];
#pragma unroll
                            for (int r = 0; r < 8; r++) org[r] = (banded && r0 + r < rows) ? blo[pl * t_size + ti + r0 + r] : 0;
#pragma unroll
                            for (int r = 0; r < 8; r++) {
                                v[r] = 0;
                                if (r0 + r < rows) {
                                    const uint8_t *src = mat[pl] + size_t(ti + r0 + r) * d.pitch[pl];
#pragma unroll
                                    for (int k = 0; k < 4; k++) {
                                        const int xx = c0 + lane * 4 + k - org[r];
                                        if (xx >= 0 && xx < wdt) v[r] |= uint32_t(src[xx]) << (8 * k);
                                    }
                                }
                            }
#pragma unroll
                            for (int r = 0; r < 8; r++)
                                if (r0 + r < rows) *reinterpret_cast<uint32_t *>(tile + (pl * WALK_TR + r0 + r) * WALK_TW + lane * 4) = v[r];
                        }
                    }
                    if (lane == 0) tblo[pl * WALK_TR] = c0;
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                colw = qri - tblo[hi * WALK_TR];
            }
            p = tile[(hi * WALK_TR + (ti - tile_t0)) * WALK_TW + colw] & 31;
            }
        } else if (WAVE) {
            if (ti >= tile_t0 + WALK_TR) {   // stage the next WALK_TR rows of both planes (uniform branch)
                tile_t0 = ti;
                const int rows = min(WALK_TR, t_size - ti);
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    const int nbytes = rows * d.pitch[pl];
                    const uint8_t *src = mat[pl] + size_t(ti) * d.pitch[pl];
                    for (int k = lane * 16; k < nbytes; k += 64 * 16)
                        *reinterpret_cast<uint4 *>(tile + pl * WALK_TR * WALK_TW + k) = *reinterpret_cast<const uint4 *>(src + k);
                    if (lane < rows) tblo[pl * WALK_TR + lane] = blo[pl * t_size + ti + lane];
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            const int r = ti - tile_t0;
            const int col = qri - tblo[hi * WALK_TR + r];
            if (col < 0 || col >= d.band_w) { status |= VPR_ST_ERR_NO_PTR; ok = false; break; }
            p = tile[hi * WALK_TR * WALK_TW + r * d.pitch[hi] + col] & 31;
        } else if (d.band_w == 16) {
            // stripe-transposed 16-cell layout (pr_q16.hip): record ti>>2 = [plane][column][row & 3]
            const int2 o = reinterpret_cast<const int2 *>(blo)[ti >> 2];
            const int col = qri - (hi ? o.y : o.x);
            if (col < 0 || col >= 16) { status |= VPR_ST_ERR_NO_PTR; ok = false; break; }
            p = mat[0][size_t(ti >> 2) * 128 + hi * 64 + col * 4 + (ti & 3)] & 31;
        } else {
            const int col = banded ? qri - blo[hi * t_size + ti] : qri;
            if (banded && (col < 0 || col >= d.band_w)) { status |= VPR_ST_ERR_NO_PTR; ok = false; break; }
            p = mat[hi][size_t(ti) * d.pitch[hi] + col] & 31;
        }
        int mv; uint32_t edit = 0;
        if (hi == ri && (p & F_SWP)) { mv = F_SWP; hi = qi; qri = WAVE ? col_get(ri, qri).x : r2q[qri]; qri++; ti++; }
        else if (p & F_MAT) { mv = F_MAT; qri++; ti++; }
        else if (p & F_SUB) { mv = F_SUB; qri++; ti++; edit = 1; }
        else if (p & F_INS) { mv = F_INS; qri++; edit = 1; }
        else if (p & F_DEL) { mv = F_DEL; ti++; edit = 1; }
        else if (hi == qi && (p & F_SWP)) { mv = F_SWP; hi = ri; qri = WAVE ? col_get(qi, qri).x : q2r[qri]; qri++; ti++; }
        else { status |= VPR_ST_ERR_NO_PTR; ok = false; break; }
        // dist.cpp:941 breaks out here with edits one entry longer than sync; a valid optimal path never
        // leaves the matrix, so report it like the reference's other walk failure
        if ((hi == qi && qri >= q_size) || (hi == ri && qri >= r_size) || ti >= t_size) {
            status |= VPR_ST_ERR_NO_PTR; ok = false; break;
        }
        last_mv = mv;
        singles++;
        const int consumes = mv & (F_MAT | F_SWP | F_SUB | F_DEL);
        int tflv, qflv, tr, qr;
        bool ins_loc;
        if (WAVE) {
            const int2 rw = row_get(ti), cw = col_get(hi, qri);
            tflv = rw.y & 0xff; tr = rw.x;
            qflv = cw.y & 0xff; qr = (hi == ri) ? qri : cw.x;
            ins_loc = ((rw.y | cw.y) & insmask) != 0;
        } else {
            tflv = tfl[ti]; tr = t2r[ti];
            qflv = (hi == ri) ? 0 : int(qfl[qri]);
            qr = (hi == ri) ? qri : q2r[qri];
            ins_loc = (insQ[tr] | insT[tr] | insQ[qr] | insT[qr]) != 0;
        }
        bool in_t = tflv & PV;
        if (consumes) in_t = in_t && !(tflv & PB);
        bool in_q = (hi == ri) ? false : bool(qflv & PV);
        if (hi == qi && consumes) in_q = in_q && !(qflv & PB);
        const bool sync = !in_t && !in_q && !ins_loc && tr == qr && (mv & (F_MAT | F_SWP | F_SUB));
        if (n >= d.path_cap) { status |= VPR_ST_ERR_LIMIT; ok = false; break; }
        if (lead) path[n] = PathEnt{uint32_t(qri) | (uint32_t(hi) << 31),
                                    uint32_t(ti) | (uint32_t(sync) << 31) | (edit << 30), qr, tr};
        n++;
    }
    if (lead) O.path_len = int32_t(n);
    if (!ok) { if (lead) { atomicOr(&O.status, status); O.n_sec = 0; } return; }
    if (WAVE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // lane 0's path stores are read back below

    credit_walk<WAVE>(B, d, O, a, path, n, status, secs, fp_group, jobs, n_jobs, jobs_cap, lead);
}

// Phase B alone, for paths produced by k_walk_rows (pr_band.hip)
template <bool WAVE>
__global__ void __launch_bounds__(64) k_credit(DevBatch B, const AlnDesc *__restrict__ descs,
                       const int32_t *__restrict__ work, int n_work, AlnOut *__restrict__ outs,
                       const PathEnt *__restrict__ paths, Section *__restrict__ secs,
                       int32_t *const *__restrict__ fp_group,
                       EdJob *__restrict__ jobs, int32_t *__restrict__ n_jobs, int32_t jobs_cap, int my_w, int ok_tag,
                       const int32_t *__restrict__ n_dev, int first = 0) {
    // first: the launch takes the list's entries from there on (the head of a device-built list, longest first, goes to the
    // wave variant: vpr_execute)
    if (n_dev) n_work = min(n_work, *n_dev);
    const int wi = first + (WAVE ? int(blockIdx.x) : int(blockIdx.x * blockDim.x + threadIdx.x));
    if (wi >= n_work) return;
    if (WAVE) __builtin_amdgcn_s_setprio(2);   // one wave per long alignment: a latency chain
    const int a = work[wi];
    if (a < 0) return;   // padding of a device-built work list
    const AlnDesc d = descs[a];
    AlnOut &O = outs[a];
    if (d.band_pad != my_w || (my_w > 0 ? O.band_ok != ok_tag : O.band_ok < 0)) return;   // (< 0: left to the tie pass)
    if (O.status & (VPR_ST_ERR_NO_PTR | VPR_ST_ERR_LIMIT)) return;   // phase A failed: n_sec is already 0
    const bool lead = !WAVE || (threadIdx.x & 63) == 0;
    credit_walk<WAVE>(B, d, O, a, paths + d.path_off, int64_t(O.path_len), 0u, secs, fp_group, jobs, n_jobs, jobs_cap, lead);
}

// ---------------------------------------------------------------------------
// K5: per-variant results from the sections (the float part of calc_prec_recall, dist.cpp:1157-1168 and
// 1284-1353), store_phase (dist.cpp:449-475) and the TP/FP/FN tally.  hipcc's float division is
// correctly rounded (no -ffast-math), so credit = 1 - float(query_ed)/ref_ed has the reference's bits;
// the parity tests compare the bit patterns.
// ---------------------------------------------------------------------------
__global__ void k_finalize(const AlnDesc *__restrict__ descs, int n_aln, const AlnOut *__restrict__ outs,
                           const Section *__restrict__ secs, int32_t *const *__restrict__ fp_group, DevResults R,
                           const uint8_t *__restrict__ sc_limit, const uint8_t *__restrict__ alias) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_aln) return;
    const AlnDesc d = descs[a];
    const AlnOut O = outs[a];      // (an alias: its source's, k_alias_outs)
    uint32_t status = O.status;
    // an alias (k_hap_alias, pr_api.hip) takes its sections and FP marks from the alignment it is a copy of: the same variants
    // under the indices of that alignment's hap slots
    int src_i = -1;
    {
        const int b = alias[d.sc], q = d.aln >> 1, t = d.aln & 1;
        const int si = ((b & 1) ? 0 : q) * 2 + ((b & 2) ? 0 : t);
        if (si != d.aln) src_i = si;
    }
    const AlnDesc ds = src_i >= 0 ? descs[d.sc * 4 + src_i] : d;
    const int64_t dq = d.qv_beg - ds.qv_beg, dt = d.tv_beg - ds.tv_beg;      // variant index of the source -> this alignment's
    if (sc_limit[d.sc] || (status & VPR_ST_ERR_LIMIT)) {
        // beyond an implementation limit (include/vcfdist_pr.h): the alignment is reported as such and its variants stay
        // unevaluated (ERRTYPE_UN); the rest of the batch is not affected
        R.aln_dist[a] = O.s;
        R.aln_end_plane[a] = uint8_t(O.end_plane);
        R.aln_beg_plane[a] = uint8_t(O.beg_plane);
        R.aln_status[a] = (status & VPR_ST_SWAP_TIE) | VPR_ST_ERR_LIMIT;
        return;
    }
    const int swap = (d.aln == 1 || d.aln == 2);
    const VarCols Q = R.v[d.qs][swap], T = R.v[d.ts][swap];
    const float *qq = R.var_qual[d.qs];
    const int32_t *fpg = fp_group[ds.qs * 2 + ((ds.aln == 1 || ds.aln == 2) ? 1 : 0)] - dq;      // indexed by this alignment's variants
    for (int64_t v = d.qv_beg; v < d.qv_end; v++) {   // passed on the REF plane: FP in a group of its own
        const int g = fpg[v];
        if (g >= 0) {
            Q.errtype[v] = VPR_ERRTYPE_FP; Q.sync_group[v] = g; Q.credit[v] = 0; Q.ref_ed[v] = 0; Q.query_ed[v] = 0;
            Q.callq[v] = qq[v];
        }
    }
    if (!(status & (VPR_ST_ERR_NO_PTR | VPR_ST_ERR_UNFINISHED))) {
        for (int k = 0; k < O.n_sec; k++) {
            Section S = secs[ds.sec_off + k];
            S.q_lo += int32_t(dq); S.q_hi += int32_t(dq); S.t_lo += int32_t(dt); S.t_hi += int32_t(dt);
            int ref_ed = S.ref_ed;
            const int query_ed = S.query_ed;
            const bool has_q = S.q_hi != S.q_lo, has_t = S.t_hi != S.t_lo;
            if (!has_t && ref_ed != 0) status |= VPR_ST_WARN_REF_ED;            // dist.cpp:1203
            if (!has_q && query_ed != ref_ed) status |= VPR_ST_WARN_QUERY_ED;   // dist.cpp:1207
            if (query_ed > ref_ed) status |= VPR_ST_WARN_EXCEEDS;               // dist.cpp:1211
            if (ref_ed == 0 && has_t) { status |= VPR_ST_WARN_ZERO_ED; ref_ed = 1; }   // dist.cpp:1219-1223
            float callq = R.max_qual;                                            // dist.cpp:1284-1288
            for (int64_t v = S.q_hi; v > S.q_lo; v--) callq = (qq[v] < callq) ? qq[v] : callq;   // std::min
            // 0/0 (query variants that cancel out, no truth variant): the reference's x86-64 SSE division
            // yields the default "real indefinite" NaN 0xFFC00000 and 1 - NaN keeps it; gfx950 would
            // produce +NaN, so the reference's bit pattern is written explicitly
            const float credit = (ref_ed == 0 && query_ed == 0) ? __uint_as_float(0xFFC00000u)
                                                                : 1 - float(query_ed) / float(ref_ed);
            const bool tp = double(credit) >= R.credit_threshold;
            for (int64_t v = S.q_hi; v > S.q_lo; v--) {
                if (fpg[v] < 0) {   // "don't overwrite FPs", dist.cpp:1295
                    Q.errtype[v] = tp ? VPR_ERRTYPE_TP : VPR_ERRTYPE_FP;
                    Q.sync_group[v] = S.sync_group; Q.credit[v] = credit; Q.ref_ed[v] = ref_ed;
                    Q.query_ed[v] = query_ed; Q.callq[v] = callq;
                }
            }
            for (int64_t v = S.t_hi; v > S.t_lo; v--) {
                T.errtype[v] = tp ? VPR_ERRTYPE_TP : VPR_ERRTYPE_FN;
                T.sync_group[v] = S.sync_group; T.credit[v] = credit; T.ref_ed[v] = ref_ed;
                T.query_ed[v] = query_ed; T.callq[v] = tp ? callq : R.max_qual;
            }
        }
    }
    R.aln_dist[a] = O.s;
    R.aln_end_plane[a] = uint8_t(O.end_plane);
    R.aln_beg_plane[a] = uint8_t(O.beg_plane);
    R.aln_status[a] = status;
}

// store_phase (dist.cpp:456-469) per supercluster, then the tally of the phasing it selects
__global__ void k_phase_tally(const AlnDesc *__restrict__ descs, int n_sc, DevResults R) {
    __shared__ unsigned long long blk[6];
    if (threadIdx.x < 6) blk[threadIdx.x] = 0;
    __syncthreads();
    const int sc = blockIdx.x * blockDim.x + threadIdx.x;
    if (sc < n_sc) {
        const int s0 = R.aln_dist[sc * 4], s1 = R.aln_dist[sc * 4 + 1], s2 = R.aln_dist[sc * 4 + 2], s3 = R.aln_dist[sc * 4 + 3];
        const int orig = s0 + s3, swp = s2 + s1;
        int phase = VPR_PHASE_NONE;
        if (orig != swp) {
            if (orig == 0) phase = VPR_PHASE_ORIG;
            else if (swp == 0) phase = VPR_PHASE_SWAP;
            else if (double(1 - float(swp) / float(orig)) > R.phase_threshold) phase = VPR_PHASE_SWAP;
            else if (double(1 - float(orig) / float(swp)) > R.phase_threshold) phase = VPR_PHASE_ORIG;
        }
        R.sc_phase[sc] = phase;
        R.orig_phase_dist[sc] = orig;
        R.swap_phase_dist[sc] = swp;
        const int w = (phase == VPR_PHASE_SWAP) ? 1 : 0;
        // alignment 0 = (query hap 0, truth hap 0), alignment 3 = (query hap 1, truth hap 1): their variant ranges
        const AlnDesc d0 = descs[sc * 4], d3 = descs[sc * 4 + 3];
        unsigned cnt[6] = {0, 0, 0, 0, 0, 0};
        for (int64_t v = d0.qv_beg; v < d0.qv_end; v++) { const int e = R.v[0][w].errtype[v]; if (e < 3) cnt[e]++; }
        for (int64_t v = d3.qv_beg; v < d3.qv_end; v++) { const int e = R.v[1][w].errtype[v]; if (e < 3) cnt[e]++; }
        for (int64_t v = d0.tv_beg; v < d0.tv_end; v++) { const int e = R.v[2][w].errtype[v]; if (e < 3) cnt[3 + e]++; }
        for (int64_t v = d3.tv_beg; v < d3.tv_end; v++) { const int e = R.v[3][w].errtype[v]; if (e < 3) cnt[3 + e]++; }
        for (int k = 0; k < 6; k++) if (cnt[k]) atomicAdd(&blk[k], (unsigned long long)cnt[k]);
    }
    __syncthreads();
    if (threadIdx.x < 6 && blk[threadIdx.x]) atomicAdd(&R.tally[threadIdx.x], blk[threadIdx.x]);
}

// ---------------------------------------------------------------------------
// K4: edit distance of deferred sections; one wave per section, row sweep over
// the shorter string with lanes over the longer one (same scan as K1, one plane).
// ---------------------------------------------------------------------------
// K4d: the same edit distance swept by ANTI-DIAGONALS, one workgroup per deferred section.  The cells of an
// anti-diagonal are independent (no in-row scan), three diagonals of 16-bit distances and both strings live in LDS,
// and a step costs one barrier: the longest section of a batch (its two strings are thousands of bases when a
// supercluster holds an SV-sized indel) is a chain of nx + ny cheap steps instead of ny rows of nx / 64 scanned tiles.
// LDS: 3 * (ny + 1) uint16 + nx + ny bytes (dynamic); the host falls back to k_ed when that does not fit.
#define ED_NT 256
__global__ void __launch_bounds__(ED_NT) k_ed_diag(DevBatch B, const AlnDesc *__restrict__ descs,
                                                   const EdJob *__restrict__ jobs, int n_jobs, Section *__restrict__ secs,
                                                   int max_short) {
    extern __shared__ __align__(16) uint8_t ed_lds[];
    const int j = blockIdx.x;
    if (j >= n_jobs) return;
    const EdJob J = jobs[j];
    const AlnDesc d = descs[J.aln];
    const uint8_t *Rs = B.ref_seq + d.r_off + J.ref_beg;
    const uint8_t *Ts = B.hap_seq[d.ts] + d.t_off + J.tru_beg;
    // X = longer string (x = 0..nx), Y = shorter (y = 0..ny); a diagonal is indexed by y
    const uint8_t *X = Rs, *Y = Ts;
    int nx = J.ref_len, ny = J.tru_len;
    if (nx < ny) { X = Ts; Y = Rs; const int tmp = nx; nx = ny; ny = tmp; }
    const int pitch = (max_short + 2 + 7) & ~7;
    uint16_t *dg = reinterpret_cast<uint16_t *>(ed_lds);          // [3][pitch]
    uint8_t *sx = ed_lds + size_t(3) * pitch * 2;                 // X then Y
    uint8_t *sy = sx + nx;
    const int tid = threadIdx.x;
    for (int i = tid; i < nx; i += ED_NT) sx[i] = X[i];
    for (int i = tid; i < ny; i += ED_NT) sy[i] = Y[i];
    if (tid == 0) dg[0] = 0;                                      // diagonal 0: the cell (0, 0)
    __syncthreads();
    // diagonal k holds the cells (x = k - y, y) for y in [max(0, k - nx), min(ny, k)], stored at index y
    for (int k = 1; k <= nx + ny; k++) {
        uint16_t *cur = dg + (k % 3) * pitch;
        const uint16_t *p1 = dg + ((k + 2) % 3) * pitch, *p2 = dg + ((k + 1) % 3) * pitch;   // diagonals k-1, k-2
        const int ylo = max(0, k - nx), yhi = min(ny, k);
        for (int y = ylo + tid; y <= yhi; y += ED_NT) {
            const int x = k - y;
            int v;
            if (y == 0) v = x;
            else if (x == 0) v = y;
            else {
                const int left = p1[y] + 1;            // (x-1, y)   on diagonal k-1
                const int up = p1[y - 1] + 1;          // (x, y-1)   on diagonal k-1
                const int di = p2[y - 1] + (sx[x - 1] != sy[y - 1]);   // (x-1, y-1) on diagonal k-2
                v = min(min(left, up), di);
            }
            cur[y] = uint16_t(v);
        }
        __syncthreads();
    }
    if (tid == 0) {
        Section &S = secs[d.sec_off + J.sec];
        S.ref_ed = dg[((nx + ny) % 3) * pitch + ny];
        S.flags &= ~SEC_DEFERRED;
    }
}

// K4w: the same edit distance by WAVEFRONTS (furthest-reaching points per diagonal, the formulation of the reference's own
// wf_ed, dist.cpp:1406-1506): step d holds, for every diagonal k = x - y within d edits of the main one, the largest x
// reachable with d edits; the answer is the first d at which diagonal nx - ny reaches x = nx.  O(d^2 + nx + ny) instead of
// nx * ny: the deferred sections are the reference / truth segments around SV-sized indels -- thousands of bases that
// differ by one block -- so d is the indel's size and most of the work is one long run of matches.
// One workgroup per section; both strings and two rows of 16-bit furthest-reaching points in LDS
// (2 * (2 * max_long + 3) * 2 + nx + ny bytes; the host picks the anti-diagonal kernel when that does not fit).
#define EDW_NEG (-30000)
#define EDW_NT 1024        // a step of a large section holds thousands of diagonals
// sel: the jobs of this launch (the host launches size classes apart: the rows in LDS are as long as the class's largest
// section, and a launch sized by a 12 000-base section would leave one workgroup per compute unit for hundreds of small ones)
__global__ void __launch_bounds__(EDW_NT) k_ed_wf(DevBatch B, const AlnDesc *__restrict__ descs,
                                                 const EdJob *__restrict__ jobs, const int32_t *__restrict__ sel, int n_jobs,
                                                 Section *__restrict__ secs, int max_long) {
    extern __shared__ __align__(16) uint8_t ed_lds[];
    if (int(blockIdx.x) >= n_jobs) return;
    const int j = sel[blockIdx.x];
    const EdJob J = jobs[j];
    const AlnDesc d = descs[J.aln];
    const uint8_t *X = B.ref_seq + d.r_off + J.ref_beg;
    const uint8_t *Y = B.hap_seq[d.ts] + d.t_off + J.tru_beg;
    int nx = J.ref_len, ny = J.tru_len;
    const int W = 2 * max_long + 3, O = max_long + 1;             // row width, index of diagonal 0
    int16_t *fr = reinterpret_cast<int16_t *>(ed_lds);            // [2][W]
    uint8_t *sx = ed_lds + size_t(2) * W * 2;
    uint8_t *sy = sx + nx;
    const int tid = threadIdx.x;
    for (int i = tid; i < nx; i += EDW_NT) sx[i] = X[i];
    for (int i = tid; i < ny; i += EDW_NT) sy[i] = Y[i];
    for (int i = tid; i < 2 * W; i += EDW_NT) fr[i] = EDW_NEG;
    // The edit distance of two strings is that of what is left of them without their common prefix and suffix, and a
    // deferred section is mostly that: flanks around one block.  (A section of 2 000 + 12 000 bases took 10 000 steps of a
    // band thousands of diagonals wide; trimmed, the shorter side is often empty and the distance is the other's length.)
    __shared__ int trim_pre, trim_suf;
    if (tid == 0) { trim_pre = min(nx, ny); trim_suf = 0; }
    __syncthreads();
    {
        const int m = min(nx, ny);
        for (int i = tid; i < m; i += EDW_NT) if (sx[i] != sy[i]) { atomicMin(&trim_pre, i); break; }
        __syncthreads();
        const int pre = trim_pre;
        if (tid == 0) trim_suf = m - pre;
        __syncthreads();
        for (int i = tid; i < m - pre; i += EDW_NT) if (sx[nx - 1 - i] != sy[ny - 1 - i]) { atomicMin(&trim_suf, i); break; }
        __syncthreads();
        const int suf = trim_suf;
        sx += pre; sy += pre; nx -= pre + suf; ny -= pre + suf;
    }
    const int kt = nx - ny;
    auto extend = [&](int x, int k) {
        while (x < nx && x - k < ny && sx[x] == sy[x - k]) x++;
        return x;
    };
    if (tid == 0) fr[O] = int16_t(extend(0, 0));
    __syncthreads();
    int dist = 0;
    if (nx == 0 || ny == 0) dist = max(nx, ny);
    else if (!(kt == 0 && fr[O] >= nx)) {
        for (int e = 1; e <= nx + ny; e++) {
            const int16_t *prev = fr + ((e + 1) & 1) * W;
            int16_t *cur = fr + (e & 1) * W;
            // diagonals that can still matter: a path through diagonal k at step e needs at least |kt - k| more edits, and the
            // distance is at most max(nx, ny) (substitutions along the shorter string, the rest as one block).  For the
            // typical deferred section -- one side thousands of bases longer than the other -- that leaves a band as wide as
            // the shorter string instead of 2 e + 1 diagonals.  (Entries outside keep older, smaller values: still reachable.)
            const int slack = max(nx, ny) - e;
            const int klo = max(-min(e, ny), kt - slack), khi = min(min(e, nx), kt + slack);
            for (int k = klo + tid; k <= khi; k += EDW_NT) {
                int best = EDW_NEG;
                const int a = prev[O + k - 1], b = prev[O + k], c = prev[O + k + 1];
                if (a >= 0 && a + 1 <= nx) best = a + 1;                                     // a base of X alone
                if (b >= 0) best = max(best, (b + 1 <= nx && b + 1 - k <= ny) ? b + 1 : b);   // a substitution (at the diagonal's end: nothing)
                if (c >= 0 && c - k <= ny && c - k >= 0) best = max(best, c);                 // a base of Y alone
                if (best >= 0) best = extend(best, k);
                cur[O + k] = int16_t(best);
            }
            __syncthreads();
            if (cur[O + kt] >= nx) { dist = e; break; }      // (one barrier per step: the next step writes the other row)
        }
    }
    if (tid == 0) {
        Section &S = secs[d.sec_off + J.sec];
        S.ref_ed = dist;
        S.flags &= ~SEC_DEFERRED;
    }
}

__global__ void __launch_bounds__(64) k_ed(DevBatch B, const AlnDesc *__restrict__ descs,
                                           const EdJob *__restrict__ jobs, int n_jobs,
                                           Section *__restrict__ secs, int32_t *__restrict__ scratch,
                                           int64_t scratch_stride) {
    const int j = blockIdx.x;
    if (j >= n_jobs) return;
    const EdJob J = jobs[j];
    const AlnDesc d = descs[J.aln];
    Section &S = secs[d.sec_off + J.sec];
    const uint8_t *Rs = B.ref_seq + d.r_off + J.ref_beg;
    const uint8_t *Ts = B.hap_seq[d.ts] + d.t_off + J.tru_beg;
    // X = longer string (lanes), Y = shorter (rows)
    const uint8_t *X = Rs, *Y = Ts;
    int nx = J.ref_len, ny = J.tru_len;
    if (nx < ny) { X = Ts; Y = Rs; const int tmp = nx; nx = ny; ny = tmp; }
    int32_t *row = scratch + int64_t(j) * scratch_stride;   // nx+1 ints: D[y][x], x = 0..nx
    const int lane = threadIdx.x;
    for (int x = lane; x <= nx; x += 64) row[x] = x;
    __syncthreads();
    for (int y = 1; y <= ny; y++) {
        const uint8_t yc = Y[y - 1];
        int carry = D_INF;      // prefix-min of (base - x) over everything left of the current tile
        int diag_in = y - 1;    // D[y-1][x0-1] for the first tile = D[y-1][0]... handled below
        for (int x0 = 0; x0 <= nx; x0 += 64) {
            const int x = x0 + lane;
            int up = D_INF, dg = D_INF;
            if (x <= nx) up = row[x];
            dg = __shfl_up(up, 1);
            if (lane == 0) dg = (x0 == 0) ? D_INF : diag_in;
            const int last_up = __shfl(up, 63);
            int b = D_INF;
            if (x <= nx) {
                if (x == 0) b = y;
                else b = min(up + 1, dg + (X[x - 1] != yc));
            }
            int v = b - x;
            int iv = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int tv = __shfl_up(iv, o);
                if (lane >= o) iv = min(iv, tv);
            }
            iv = min(iv, carry);
            if (x <= nx) row[x] = iv + x;
            carry = __shfl(iv, 63);
            diag_in = last_up;
        }
        __syncthreads();
    }
    if (lane == 0) { S.ref_ed = row[nx]; S.flags &= ~SEC_DEFERRED; }
}

