// pr_api.hip -- host side of the C ABI (include/vcfdist_pr.h): device memory,
// work planning, kernel launches on the library's own streams, result download
// and the float finalisation of calc_prec_recall (dist.cpp:1284-1353).
//
// Execution plan (band_mode 1, the default): every alignment first runs the exact banded
// single-wave kernels (pr_band.hip) with a 64-cell window; alignments whose window fails the
// exit test (exit_min <= s) are re-run with 256, then 1024 cells, then with the dense
// workgroup-per-alignment kernels (pr_kernels.hip).  band_mode 0 runs the dense kernels only.
//
// There is no CPU fallback: without a HIP device every entry point that needs one
// returns VPR_ERR_DEVICE.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/vcfdist_pr.h"
#include "pr_device.h"
#include "pr_plan.h"
extern "C" int vpr_batch_skeleton_from_variants(const vpr_variants *v, vpr_owned_batch **out);   // generate.cpp
#include "pr_kernels.hip"
#include "pr_band.hip"
#include "pr_walkseg.hip"
#include "pr_q16.hip"
#include "pr_zl.hip"
#include "pr_d1.hip"
#include "pr_ed.hip"
#include "pr_gen.hip"
#include "pr_wide.hip"
#include "pr_strip.hip"
#include "pr_tie.hip"

#include "pr_host.h"

namespace {


// window level an alignment has in a plan of level lv (a 16-cell plan holds its long alignments at LONG_LV)
inline int plan_level_of(int lv, int long_lt, int Lt) { return (lv <= LV_Q16 && Lt >= long_lt) ? LONG_LV : lv; }

// descriptor k of a plan
AlnDesc plan_desc(const vpr_handle *h, const Plan &P, size_t k) {
    if (!P.lazy) return P.descs[k];
    AlnDesc d = h->descs[size_t(P.work[k])];
    (void)window_layout(d, plan_level_of(P.lv, P.long_lt, d.Lt), int64_t(P.off128[k]) * 128, P.tag_or);
    return d;
}

// the device's copy of a lazy plan: descriptors in plan order (plan_descs) and by alignment (descs)
__global__ void k_build_plan(BatchOffsets O, const int32_t *__restrict__ work, const uint32_t *__restrict__ off128, int n, int lv,
                             int long_lt, int tag_or, AlnDesc *__restrict__ plan_descs, AlnDesc *__restrict__ descs) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int a = work[k];
    AlnDesc d = base_desc(O, a);
    const int dl = (lv <= LV_Q16 && d.Lt >= long_lt) ? 2 /* LONG_LV */ : lv;
    (void)window_layout(d, dl, int64_t(off128[k]) * 128, tag_or);
    plan_descs[k] = d;
    descs[a] = d;
}

// ---------------------------------------------------------------------------
// Planning round 0 on the device (plan0_device below): keys, needs, the plan order's offsets, the lane level's wave headers.
// ---------------------------------------------------------------------------
// per alignment a (input order): sort key = truth rows of a short alignment, 0xffff for a long one (they sort to the front and
// are then replaced by the host's order: longest matrices first), its own id, and its workspace need in 128-byte units
__global__ void k_plan_keys(BatchOffsets O, int n_al, int lv, int long_lt, uint16_t *__restrict__ keys, int32_t *__restrict__ vals,
                            uint32_t *__restrict__ need) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_al) return;
    const int sc = a >> 2, i = a & 3, qs = i >> 1, ts = 2 + (i & 1);
    AlnDesc d{};
    d.Lq = int32_t(O.hap_off[qs][sc + 1] - O.hap_off[qs][sc]);
    d.Lt = int32_t(O.hap_off[ts][sc + 1] - O.hap_off[ts][sc]);
    d.Lr = int32_t(O.ref_off[sc + 1] - O.ref_off[sc]);
    d.path_cap = d.Lq + d.Lr + d.Lt + 4;
    const int dl = (lv <= LV_Q16 && d.Lt >= long_lt) ? 2 /* LONG_LV */ : lv;
    keys[a] = d.Lt >= long_lt ? uint16_t(0xffff) : uint16_t(d.Lt);
    vals[a] = a;
    need[a] = uint32_t(window_layout(d, dl, 0, 0) / 128);
}
__global__ void k_plan_gather(const int32_t *__restrict__ order, const uint32_t *__restrict__ need, uint32_t *__restrict__ out,
                              int32_t *__restrict__ pos, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int a = order[k];
    out[k] = need[a];
    pos[a] = k;          // (the inverse: an alignment's place in the plan, vpr_handle::plan0_pos)
}
// wave headers of the lane level's short part (order: the part's work list): largest lengths per 64 alignments ...
__global__ void __launch_bounds__(64) k_zl_hdr(BatchOffsets O, const int32_t *__restrict__ order, int n_short, ZlWave *__restrict__ hdr) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const int j = w * 64 + lane;
    int lq = 0, lr = 0, lt = 0;
    if (j < n_short) {
        const int a = order[j];
        const int sc = a >> 2, i = a & 3, qs = i >> 1, ts = 2 + (i & 1);
        lq = int32_t(O.hap_off[qs][sc + 1] - O.hap_off[qs][sc]);
        lt = int32_t(O.hap_off[ts][sc + 1] - O.hap_off[ts][sc]);
        lr = int32_t(O.ref_off[sc + 1] - O.ref_off[sc]);
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
// ... and their offsets (one workgroup).  totals: {position words, log units (16 B), largest length}
__global__ void __launch_bounds__(1024) k_zl_scan(ZlWave *__restrict__ hdr, int n_waves, int64_t *__restrict__ totals) {
    __shared__ int64_t s_in[1024], s_log[1024];
    __shared__ int s_len;
    const int tid = threadIdx.x;
    if (tid == 0) s_len = 0;
    const int per = (n_waves + 1023) / 1024;
    const int b = min(n_waves, tid * per), e = min(n_waves, b + per);
    int64_t si = 0, sl = 0;
    int lm = 0;
    for (int k = b; k < e; k++) {
        const ZlWave W = hdr[k];
        si += 64 * (int64_t(W.mq) + W.mr + W.mt); sl += 80 * int64_t(W.mt);
        lm = max(lm, max(W.mq, max(W.mr, W.mt)));
    }
    s_in[tid] = si; s_log[tid] = sl;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int64_t ai = tid >= o ? s_in[tid - o] : 0, al = tid >= o ? s_log[tid - o] : 0;
        __syncthreads();
        s_in[tid] += ai; s_log[tid] += al;
        __syncthreads();
    }
    int64_t oi = s_in[tid] - si, ol = s_log[tid] - sl;
    for (int k = b; k < e; k++) {
        ZlWave W = hdr[k];
        W.in_off = oi; W.log_off = ol;
        oi += 64 * (int64_t(W.mq) + W.mr + W.mt); ol += 80 * int64_t(W.mt);
        hdr[k] = W;
    }
    if (lm) atomicMax(&s_len, lm);
    __syncthreads();
    if (tid == 1023) { totals[0] = s_in[1023]; totals[1] = s_log[1023]; totals[2] = s_len; }
}


int class_of(int len) {
    for (int k = 0; k < N_CLASSES; k++)
        if (len <= CLASSES[k].max_len) return k;
    return -1;
}

size_t fwd_lds_bytes(int cls, int Lq, int Lr) {
    const int C = CLASSES[cls].c, NT = CLASSES[cls].nt;
    const size_t PQ = (Lq + C - 1) / C * C, PR = (Lr + C - 1) / C * C;
    return (PQ + PR + 8 + 2 * (NT / 64)) * 4;
}
// s16: score rows as int16 (k_bwd<NT, C, true>)
size_t bwd_lds_bytes(int cls, int Lq, int Lr, bool s16 = false) {
    const int C = CLASSES[cls].c, NT = CLASSES[cls].nt;
    const size_t PQ = (Lq + C - 1) / C * C, PR = (Lr + C - 1) / C * C;
    const size_t FQ = (PQ + 16 + 15) & ~size_t(15), FR = (PR + 16 + 15) & ~size_t(15);
    return (((PQ + PR + 8) * (s16 ? 2 : 4) + 15) & ~size_t(15)) + 2 * (FQ + FR) + 4 * (NT / 64) * 4 + 16;
}
// largest score the int16 rows must hold: query-variant entries on a path <= query variants of the alignment
bool s16_ok(const AlnDesc &d) { return d.qv_end - d.qv_beg < 32000; }

// the one-workgroup dense kernels can hold the alignment's rows in LDS (the backward one maybe only as int16)
bool dense_fits(const AlnDesc &d) {
    const int cls = class_of(std::max(d.Lq, d.Lr));
    if (cls < 0 || cls >= STRIP_ONLY_CLS) return false;
    return fwd_lds_bytes(cls, d.Lq, d.Lr) <= LDS_MAX &&
           (bwd_lds_bytes(cls, d.Lq, d.Lr) <= LDS_MAX || (s16_ok(d) && bwd_lds_bytes(cls, d.Lq, d.Lr, true) <= LDS_MAX));
}

typedef void (*AlnKernel)(DevBatch, const AlnDesc *, const int32_t *, uint8_t *, AlnOut *, const int32_t *);
AlnKernel fwd_kernel(int cls) {
    switch (cls) {
        case 0: return k_fwd<64, 1>;
        case 1: return k_fwd<64, 4>;
        case 2: return k_fwd<256, 4>;
        case 3: return k_fwd<256, 8>;
        case 4: return k_fwd<1024, 8>;
        case 5: return k_fwd<1024, 16>;
        default: return k_fwd<1024, 32>;
    }
}
AlnKernel bwd_kernel(int cls, bool s16 = false) {
    if (s16) {
        switch (cls) {
            case 0: return k_bwd<64, 1, true>;
            case 1: return k_bwd<64, 4, true>;
            case 2: return k_bwd<256, 4, true>;
            case 3: return k_bwd<256, 8, true>;
            case 4: return k_bwd<1024, 8, true>;
            case 5: return k_bwd<1024, 16, true>;
            default: return k_bwd<1024, 32, true>;
        }
    }
    switch (cls) {
        case 0: return k_bwd<64, 1, false>;
        case 1: return k_bwd<64, 4, false>;
        case 2: return k_bwd<256, 4, false>;
        case 3: return k_bwd<256, 8, false>;
        case 4: return k_bwd<1024, 8, false>;
        case 5: return k_bwd<1024, 16, false>;
        default: return k_bwd<1024, 32, false>;
    }
}
typedef void (*BandFwd)(DevBatch, const AlnDesc *, const int32_t *, uint8_t *, int32_t *, AlnOut *);
typedef void (*BandBwd)(DevBatch, const AlnDesc *, const int32_t *, uint8_t *, const int32_t *, AlnOut *, int);
// one-alignment-per-workgroup window kernels: 64 cells (one wave, striped), 256 and 1024 cells (4 / 16 waves)
BandFwd band_fwd_kernel(int lv) { return lv == LV_C1 ? BandFwd(k_fwd_stripe) : lv == LV_C4 ? BandFwd(k_fwd_wide<4>) : BandFwd(k_fwd_wide<16>); }
BandBwd band_bwd_kernel(int lv) { return lv == LV_C1 ? BandBwd(k_bwd_stripe) : lv == LV_C4 ? BandBwd(k_bwd_wide<4>) : BandBwd(k_bwd_wide<16>); }
const char *band_fwd_name(int lv) { return lv == LV_Z ? "k_zero_lane" : lv == LV_Q16 ? "k_fwd_q16" : lv == LV_C1 ? "k_fwd_stripe" : lv == LV_C4 ? "k_fwd_wide<4>" : "k_fwd_wide<16>"; }
const char *band_bwd_name(int lv) { return lv <= LV_Q16 ? "k_bwd_q16" : lv == LV_C1 ? "k_bwd_stripe" : lv == LV_C4 ? "k_bwd_wide<4>" : "k_bwd_wide<16>"; }

int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Append the alignments whose window was rejected to the round's fail list (aggregated per workgroup of 256:
// contended single-word atomics would cost more than the forward sweep itself).
__global__ void k_collect_fails(const int32_t *__restrict__ work, int n, const AlnOut *__restrict__ outs,
                                int32_t *__restrict__ fail_list, int32_t *__restrict__ cnt, int pad4,
                                const int32_t *__restrict__ n_dev) {
    if (n_dev) n = min(n, *n_dev);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const int a = live ? work[i] : -1;
    const bool bad = live && a >= 0 && !outs[a].band_ok;
    const unsigned long long mbad = __ballot(bad);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // pad4: the list feeds a four-alignments-per-wave kernel directly.  A workgroup's rejected ids are consecutive in the
    // (row-count sorted) work list, so they are kept together and padded with -1 to a multiple of four: groups of four then
    // never mix alignments of very different lengths.  (Padded per workgroup of 256 list entries, not per wavefront of 64: at
    // 9 % rejects a wavefront has five or six, and rounding each wavefront's up left a quarter of the 16-cell round's lanes idle.)
    const int npop = __popcll(mbad);
    // one atomic per workgroup (a single contended counter serialises in L2: ~10 ns each)
    __shared__ int wcnt[4], wbase, wtot, wpad;
    if (lane == 0) wcnt[wave] = npop;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        const int totp = pad4 ? ((tot + 3) & ~3) : tot;
        wbase = totp ? atomicAdd(cnt, totp) : 0;
        wtot = tot; wpad = totp - tot;
    }
    __syncthreads();
    int base = wbase;
    for (int w = 0; w < wave; w++) base += wcnt[w];
    if (bad) fail_list[base + __popcll(mbad & ((1ull << lane) - 1ull))] = a;
    if (int(threadIdx.x) < wpad) fail_list[wbase + wtot + int(threadIdx.x)] = -1;
}

// The same list IN THE ORDER OF THE WORK LIST (three small launches instead of one): the zero-distance level's rejects feed lane
// kernels (pr_d1.hip), whose 64 lanes run in lockstep -- the work list is sorted by truth rows, and a list appended in the order
// in which the workgroups happen to reach the counter mixes alignments of a hundred rows with alignments of a thousand.
//   k_fails_count: rejected entries per workgroup of 256 (padded to a multiple of four with pad4) -> blk[b]
//   k_fails_scan:  exclusive prefix sums of blk[0 .. nb) in place (one workgroup), the total -> *cnt
//   k_fails_scatter: the ids to fail_list[blk[b] ...], -1 behind them up to the padded length
__global__ void k_fails_count(const int32_t *__restrict__ work, int n, const AlnOut *__restrict__ outs, int32_t *__restrict__ blk,
                              int pad4, const int32_t *__restrict__ n_dev) {
    if (n_dev) n = min(n, *n_dev);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = i < n ? work[i] : -1;
    const bool bad = a >= 0 && !outs[a].band_ok;
    const int npop = __popcll(__ballot(bad));
    __shared__ int wcnt[4];
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = npop;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        blk[blockIdx.x] = pad4 ? ((tot + 3) & ~3) : tot;
    }
}

// (256 threads: a workgroup of 1 024 needs sixteen free wave slots on ONE compute unit at once, and beside a launch of tens of
// thousands of long-lived one-wave workgroups -- the stress workload's long part -- that never happens before the launch drains:
// this kernel waited 35 - 45 ms for a slot, twice per step, whatever its stream's priority.  Four waves find room at once.)
#define FAILS_SCAN_NT 256
__global__ void __launch_bounds__(FAILS_SCAN_NT) k_fails_scan(int32_t *__restrict__ blk, int nb, int32_t *__restrict__ cnt) {
    __shared__ int32_t s_sum[FAILS_SCAN_NT];
    const int tid = threadIdx.x;
    const int per = (nb + FAILS_SCAN_NT - 1) / FAILS_SCAN_NT;
    const int b = min(nb, tid * per), e = min(nb, b + per);
    int32_t sum = 0;
    for (int k = b; k < e; k++) sum += blk[k];
    s_sum[tid] = sum;
    __syncthreads();
    for (int o = 1; o < FAILS_SCAN_NT; o <<= 1) {
        const int32_t v = tid >= o ? s_sum[tid - o] : 0;
        __syncthreads();
        s_sum[tid] += v;
        __syncthreads();
    }
    int32_t off = s_sum[tid] - sum;
    for (int k = b; k < e; k++) { const int32_t c = blk[k]; blk[k] = off; off += c; }
    if (tid == FAILS_SCAN_NT - 1) *cnt = s_sum[FAILS_SCAN_NT - 1];
}

__global__ void k_fails_scatter(const int32_t *__restrict__ work, int n, const AlnOut *__restrict__ outs, const int32_t *__restrict__ blk,
                                int32_t *__restrict__ fail_list, int pad4, const int32_t *__restrict__ n_dev) {
    if (n_dev) n = min(n, *n_dev);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = i < n ? work[i] : -1;
    const bool bad = a >= 0 && !outs[a].band_ok;
    const unsigned long long mbad = __ballot(bad);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int wcnt[4];
    if (lane == 0) wcnt[wave] = __popcll(mbad);
    __syncthreads();
    const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    const int totp = pad4 ? ((tot + 3) & ~3) : tot;
    const int wbase = blk[blockIdx.x];
    int base = wbase;
    for (int w = 0; w < wave; w++) base += wcnt[w];
    if (bad) fail_list[base + __popcll(mbad & ((1ull << lane) - 1ull))] = a;
    if (int(threadIdx.x) < totp - tot) fail_list[wbase + tot + int(threadIdx.x)] = -1;
}

// Mirror a fail list and its length into host-pinned memory (launched on the stream that produced them, right behind
// k_collect_fails, so it is never queued behind another stream's bulk kernel).
__global__ void k_publish_fails(const int32_t *__restrict__ list, const int32_t *__restrict__ cnt, int32_t *__restrict__ h_list,
                                int32_t *__restrict__ h_cnt, int copy_list) {
    const int n = *cnt;
    if (copy_list)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) h_list[i] = list[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) *h_cnt = n;
    __threadfence_system();
}

// Stage a retry plan from host-pinned memory in one launch: descriptors scattered to their alignment's slot, the work
// list copied, the ladder's fail counters cleared (n_zero > 0 on the first plan of a round).
__global__ void k_stage(const AlnDesc *__restrict__ src, const int32_t *__restrict__ src_work, int n, AlnDesc *__restrict__ dst,
                        int32_t *__restrict__ dst_work, int32_t *__restrict__ zero, int n_zero) {
    // (one descriptor per thread, read straight from host memory.  Beside the bulk kernels a launch takes 1 - 6 ms, and just
    // as long when a wave fetches its 64 descriptors in linear order through LDS: it waits for workgroup slots, not for the link)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_zero) zero[i] = 0;
    if (i >= n) return;
    const AlnDesc d = src[i];
    dst[d.sc * 4 + d.aln] = d;
    dst_work[i] = src_work[i];
}

// alignments the backward sweeps left to the tie pass: {alignment, level tag} (AlnOut::band_ok = TIE_MARK(tag))
__global__ void k_collect_ties(const AlnOut *__restrict__ outs, int n, int4 *__restrict__ list, int32_t *__restrict__ cnt, int cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = outs[i].band_ok;
    if (b >= 0) return;
    const int k = atomicAdd(cnt, 1);
    if (k < cap) list[k] = make_int4(i, -b - 1, outs[i].n_sec, outs[i].s);
}

// the same over a work list (the long / short part of round 0, right behind its backward sweep)
__global__ void k_collect_ties_list(const int32_t *__restrict__ work, int n, const AlnOut *__restrict__ outs, int4 *__restrict__ list,
                                    int32_t *__restrict__ cnt, int cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = work[i];
    if (a < 0) return;
    const int b = outs[a].band_ok;
    if (b >= 0) return;
    const int k = atomicAdd(cnt, 1);
    if (k < cap) list[k] = make_int4(a, -b - 1, outs[a].n_sec, outs[a].s);
}
// long alignments accepted at `tag` whose forward sweep met a tied cell within the alignment's distance
// (AlnOut::path_len = that cell's distance + 1, k_fwd_stripe): candidates for a speculative replay
__global__ void k_collect_spec(const int32_t *__restrict__ work, int n, const AlnOut *__restrict__ outs, int tag, int4 *__restrict__ list,
                               int32_t *__restrict__ cnt, int cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = work[i];
    if (a < 0) return;
    const AlnOut o = outs[a];
    if (o.band_ok != tag || o.path_len <= 0 || o.path_len - 1 > o.s) return;
    const int k = atomicAdd(cnt, 1);
    if (k < cap) list[k] = make_int4(a, tag, 0, o.s);
}
__global__ void k_publish_ties(const int4 *__restrict__ list, const int32_t *__restrict__ cnt, int4 *__restrict__ h_list,
                               int32_t *__restrict__ h_cnt, int cap) {
    const int n = min(*cnt, cap);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) h_list[i] = list[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) *h_cnt = *cnt;
    __threadfence_system();
}

// everything vpr_execute starts from, in one launch: per-alignment scalars zeroed, "passed on the REF plane" groups unset,
// the per-variant columns at the reference's initial values (ERRTYPE_UN, 0; variant.cpp:45-52), counters zeroed
struct InitArgs { AlnOut *outs; int64_t na; int32_t *fp[4]; int64_t n_fp[4]; VarCols v[4][2]; int64_t n_var[4];
                  unsigned long long *tally; int32_t *njobs; int32_t *cnt; int n_cnt; };
__global__ void k_init_execute(InitArgs A) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < A.na) {
        int4 *o = reinterpret_cast<int4 *>(A.outs + i);      // sizeof(AlnOut) == 40: two 16-byte stores + 8 bytes
        int32_t *w = reinterpret_cast<int32_t *>(A.outs + i);
#pragma unroll
        for (int k = 0; k < int(sizeof(AlnOut) / 4); k++) w[k] = 0;
        (void)o;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (i < A.n_fp[q]) A.fp[q][i] = -1;
        if (i < A.n_var[q]) {
#pragma unroll
            for (int w = 0; w < 2; w++) {
                const VarCols &V = A.v[q][w];
                V.errtype[i] = VPR_ERRTYPE_UN; V.sync_group[i] = 0; V.credit[i] = 0.0f; V.ref_ed[i] = 0; V.query_ed[i] = 0; V.callq[i] = 0.0f;
            }
        }
    }
    if (i < 6) A.tally[i] = 0ull;
    if (i == 0) *A.njobs = 0;
    if (i < A.n_cnt) A.cnt[i] = 0;
}

__global__ void k_flag(int32_t *__restrict__ h_flag, int32_t v) {
    *h_flag = v;
    __threadfence_system();
}

// ---- identical alignments of a supercluster.  Its four alignments pair query hap 0 / 1 with truth hap 0 / 1; where a callset
// is homozygous over the whole supercluster its two haps are the same strings with the same pointers and flags, and alignments
// that differ only in which of them they name are the same computation with the same result -- unless a swap tie can occur,
// see below.  k_hap_alias finds such
// superclusters (bit 0: query hap 1 == query hap 0, bit 1: truth hap 1 == truth hap 0: every Level A array and the variant
// positions compared); the planner leaves the aliases out, k_alias_outs gives them their source's alignment scalars and
// k_finalize writes their variants' results from the source's sections (with their own qualities).
__global__ void k_hap_alias(DevBatch B, uint8_t *__restrict__ bits) {
    const int sc = blockIdx.x * blockDim.x + threadIdx.x;
    if (sc >= B.n_sc) return;
    auto same = [&](int a, int b) -> bool {
        const int64_t oa = B.hap_off[a][sc], ob = B.hap_off[b][sc];
        const int64_t n = B.hap_off[a][sc + 1] - oa;
        if (n != B.hap_off[b][sc + 1] - ob) return false;
        const int64_t va = B.var_off[a][sc], vb = B.var_off[b][sc];
        const int64_t nv = B.var_off[a][sc + 1] - va;
        if (nv != B.var_off[b][sc + 1] - vb) return false;
        for (int64_t k = 0; k < nv; k++) if (B.var_pos[a][va + k] != B.var_pos[b][vb + k]) return false;
        for (int64_t k = 0; k < n; k++)
            if (B.hap_seq[a][oa + k] != B.hap_seq[b][ob + k] || B.hap_ptr[a][oa + k] != B.hap_ptr[b][ob + k] ||
                B.hap_flag[a][oa + k] != B.hap_flag[b][ob + k]) return false;
        return true;
    };
    // Not where a cell can have two swap sources (a position with a second candidate, k_prep_cand): which of two optimal
    // sources the reference keeps depends on the iteration order of its hash containers, and its cell keys carry the
    // alignment's index (hi = 2 i + plane, dist.cpp:286-289) -- there the four alignments differ even on identical strings.
    for (int q = 0; q < 2; q++) {
        const int64_t qo = B.hap_off[q][sc], nq = B.hap_off[q][sc + 1] - qo, ro = B.ref_off[sc], nr = B.ref_off[sc + 1] - ro;
        for (int64_t k = 0; k < nq; k++) if (B.cand_q[q][qo + k].y >= 0) { bits[sc] = 0; return; }
        for (int64_t k = 0; k < nr; k++) if (B.cand_r[q][ro + k].y >= 0) { bits[sc] = 0; return; }
    }
    uint8_t r = 0;
    if (same(0, 1)) {       // (the reference-side arrays of a query hap: pointers into it and the flags of its variants' positions)
        const int64_t ro = B.ref_off[sc], nr = B.ref_off[sc + 1] - ro;
        bool eq = true;
        for (int64_t k = 0; k < nr && eq; k++) eq = B.ref_ptr[0][ro + k] == B.ref_ptr[1][ro + k] && B.ref_flag[0][ro + k] == B.ref_flag[1][ro + k];
        if (eq) r |= 1;
    }
    if (same(2, 3)) r |= 2;
    bits[sc] = r;
}
// (alias_source -- source of alignment i of a supercluster with alias bits b: pr_device.h)
__global__ void k_alias_descs(BatchOffsets O, const uint8_t *__restrict__ bits, int n_sc, AlnDesc *__restrict__ descs) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= 4 * n_sc) return;
    if (alias_source(bits[a >> 2], a & 3) >= 0) descs[a] = base_desc(O, a);
}
__global__ void k_alias_outs(const uint8_t *__restrict__ bits, int n_sc, AlnOut *__restrict__ outs) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= 4 * n_sc) return;
    const int src = alias_source(bits[a >> 2], a & 3);
    if (src >= 0) outs[a] = outs[(a & ~3) | src];
}

__global__ void k_scatter_descs(const AlnDesc *__restrict__ src, int n, AlnDesc *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AlnDesc d = src[i];
    dst[d.sc * 4 + d.aln] = d;
}
// the same for the plan positions in `pos` only (what a previous execute's retry rounds replaced)
__global__ void k_scatter_some(const AlnDesc *__restrict__ src, const int32_t *__restrict__ pos, int n, AlnDesc *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AlnDesc d = src[pos[i]];
    dst[d.sc * 4 + d.aln] = d;
}

// Assign arena offsets (flag matrices, band origins, walk scratch) to `alns` and cut them into chunks
// that fit the arena.  lv: window level of the plan (LV_DENSE: dense layout + kernel classes).
// tag_or: TIE_TAG_BIT for the plan of a tie round (see pr_device.h), else 0
int make_plan(vpr_handle *h, const std::vector<int32_t> &alns, int lv, Plan &P, uint8_t *arena, int64_t arena_bytes, int tag_or = 0,
              bool lazy = false) {
    {   // (the work list and the offsets keep their pages: plan 0 of every upload is as large as the last one)
        std::vector<int32_t> kw; std::vector<uint32_t> ko;
        kw.swap(P.work); ko.swap(P.off128);
        P = Plan();
        kw.clear(); ko.clear();
        P.work.swap(kw); P.off128.swap(ko);
    }
    P.lv = lv;
    P.arena = arena;
    P.tag_or = tag_or;
    const int LLT = h->long_lt;
    P.long_lt = LLT;
    P.lazy = lazy && lv != LV_DENSE;
    auto level_of = [&](const AlnDesc &d) { return plan_level_of(lv, LLT, d.Lt); };
    // Order: the long alignments first, longest first (they are latency chains and must start early).  The
    // rest keeps its input order, except at LV_Q16 where four alignments share a wave in lockstep and are
    // therefore grouped by their number of truth rows (counting sort, longest first, stable).
    const size_t n_al = alns.size();
    const bool dbg = (h->debug || getenv("VPR_TIMING")) && n_al >= 1000000;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double T0 = now();
    auto lap = [&](const char *what) { if (dbg) fprintf(stderr, "[vpr]   make_plan %-24s %.3f s\n", what, now() - T0); };
    // compact per-alignment figures (the descriptors are computed on demand: h->descs): truth rows, and the workspace
    // bytes at the alignment's level in 128-byte units (0 with `bad` set: cannot be placed)
    std::vector<int32_t> &lt = h->scratch_i32[0], &lq = h->scratch_i32[2], &lr = h->scratch_i32[3];
    std::vector<uint32_t> &need128 = h->scratch_u32[0];
    lt.resize(n_al); lq.resize(n_al); lr.resize(n_al); need128.resize(n_al);
    std::atomic<size_t> bad{n_al};          // first alignment (in input order) that cannot be placed
    // (lazy plans: what pass 3 would gather in plan order -- the parts' sums, the levels -- is taken here, in input order)
    struct Sums { int64_t cells = 0, in_bytes = 0, part_cells[2] = {0, 0}, part_in[2] = {0, 0}, part_dense[2] = {0, 0}, part_rows[2] = {0, 0}; };
    std::vector<Sums> sums1(PAR_MAX);
    std::vector<std::vector<int32_t>> big_of(PAR_MAX);      // per slice: positions of the long alignments
    par_for(n_al, [&](size_t b, size_t e, int tid) {
        Sums &S1 = sums1[size_t(tid)];
        std::vector<int32_t> &bigs = big_of[size_t(tid)];
        AlnDesc d{};
        for (size_t i = b; i < e; i++) {
            h->descs.lens(size_t(alns[i]), d.Lq, d.Lr, d.Lt);       // (all this pass looks at: window_layout reads the lengths and path_cap)
            d.path_cap = d.Lq + d.Lr + d.Lt + 4;
            lt[i] = d.Lt; lq[i] = d.Lq; lr[i] = d.Lr;
            const int dl = level_of(d);
            if (lv == LV_DENSE || d.Lt >= LLT) bigs.push_back(int32_t(i));
            if (P.lazy) {
                const int W = LV_WINDOW[dl], part = d.Lt >= LLT ? 0 : 1;
                S1.part_cells[part] += int64_t(std::min(W, d.Lq) + std::min(W, d.Lr)) * d.Lt;
                S1.part_in[part] += 6 * int64_t(d.Lq) + 6 * int64_t(d.Lt) + 6 * int64_t(d.Lr);
                S1.part_dense[part] += int64_t(d.Lq + d.Lr) * d.Lt;
                S1.part_rows[part] += d.Lt;
                h->level[size_t(alns[i])] = uint8_t(dl);
            }
            int64_t need;
            bool too_long = false;
            if (dl != LV_DENSE) {
                need = window_layout(d, dl, 0, 0);
            } else {
                // (the backward kernel's int32 score rows may not fit: it then runs with int16 rows, 4 B per cell; what fits
                // neither is left to the column strips, which have no size limit -- unless they are switched off)
                too_long = !dense_fits(d) && h->no_strips;
                const int64_t pb = round_up(int64_t(d.path_cap) * int64_t(sizeof(PathEnt)) + 128, 128);
                need = round_up(int64_t(round_up(d.Lq, 32)) * d.Lt, 128) + round_up(int64_t(round_up(d.Lr, 32)) * d.Lt, 128) + pb + 128;
            }
            need128[i] = uint32_t(std::min<int64_t>(need / 128, 0xffffffffll));
            if (too_long || need > arena_bytes) {
                size_t cur = bad.load();
                while (i < cur && !bad.compare_exchange_weak(cur, i)) {}
            }
        }
    });
    lap("pass 1 (rows, need)");
    if (bad.load() < n_al) {
        const AlnDesc d = h->descs[size_t(alns[bad.load()])];
        bool too_long = false;
        if (level_of(d) == LV_DENSE) too_long = !dense_fits(d) && h->no_strips;
        if (too_long)
            return fail(h, VPR_ERR_ARG, "supercluster %d alignment %d too long for the dense kernels (Lq=%d Lr=%d)", d.sc, d.aln, d.Lq, d.Lr);
        h->last_need = int64_t(need128[bad.load()]) * 128;
        return fail(h, VPR_ERR_NOMEM, "workspace (%lld bytes) too small for supercluster %d alignment %d (%lld bytes)",
                    (long long)arena_bytes, d.sc, d.aln, (long long)h->last_need);
    }
    std::vector<int32_t> &order = P.work;   // plan order: positions in `alns` first, alignment ids in the end
    std::vector<uint32_t> &off = P.off128;  // need, then offset inside the chunk (128-byte units)
    size_t n_big = 0;                       // long alignments, in front of the plan
    {
        std::vector<std::pair<int64_t, int32_t>> big;
        for (const auto &bl : big_of)
            for (const int32_t i32 : bl) {
                const size_t i = size_t(i32);
                const AlnDesc d = h->descs[size_t(alns[i])];
                const int W = LV_WINDOW[level_of(d)];
                const int64_t mat = W ? int64_t(round_up(std::min(W, d.Lq), 16) + round_up(std::min(W, d.Lr), 16)) * d.Lt
                                      : int64_t(round_up(d.Lq, 32) + round_up(d.Lr, 32)) * d.Lt;
                big.emplace_back(-mat, int32_t(i));
            }
        std::sort(big.begin(), big.end());
        order.reserve(n_al);
        for (auto &b : big) order.push_back(b.second);
        n_big = big.size();
        if (lv <= LV_Q16) {
            // counting sort by rows, longest first, stable: as many bins as the longest short alignment has rows; every
            // thread counts its slice, then scatters it behind the slices in front of it
            const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
            const size_t nt = std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(hw, PAR_MAX), n_al / 32768));
            std::vector<int32_t> tops(nt, 0);
            par_for(n_al, [&](size_t b, size_t e, int tid) {
                int32_t top = 0;
                for (size_t i = b; i < e; i++) if (lt[i] < LLT) top = std::max(top, lt[i]);
                tops[size_t(tid)] = top;
            });
            int32_t top = 0;
            for (int32_t t : tops) top = std::max(top, t);
            const size_t NB = size_t(top) + 1;
            std::vector<int64_t> cnt(NB * nt, 0);       // [bin][thread]
            par_for(n_al, [&](size_t b, size_t e, int tid) {
                for (size_t i = b; i < e; i++) if (lt[i] < LLT) cnt[size_t(top - lt[i]) * nt + size_t(tid)]++;
            });
            int64_t run = 0;
            for (size_t k = 0; k < NB * nt; k++) { const int64_t c = cnt[k]; cnt[k] = run; run += c; }
            const size_t base = order.size();
            order.resize(base + size_t(run));
            par_for(n_al, [&](size_t b, size_t e, int tid) {
                for (size_t i = b; i < e; i++)
                    if (lt[i] < LLT) order[base + size_t(cnt[size_t(top - lt[i]) * nt + size_t(tid)]++)] = int32_t(i);
            });
        } else if (lv != LV_DENSE) {
            for (size_t i = 0; i < n_al; i++)
                if (lt[i] < LLT) order.push_back(int32_t(i));
        }
    }
    lap("order");
    // pass 2 (one thread, 4 bytes per alignment): offsets and the cut into chunks that fit the workspace
    const size_t n = order.size();
    off.resize(n);
    {
        // the needs in plan order; when everything fits one chunk (the usual case) the offsets are a parallel prefix sum
        const int64_t cap128 = arena_bytes / 128;
        std::vector<int64_t> part(PAR_MAX + 1, 0);
        par_for(n, [&](size_t b, size_t e, int tid) {
            int64_t sum = 0;
            for (size_t k = b; k < e; k++) { const uint32_t v = need128[size_t(order[k])]; off[k] = v; sum += v; }
            part[size_t(tid) + 1] = sum;
        });
        for (int t = 0; t < PAR_MAX; t++) part[size_t(t) + 1] += part[size_t(t)];
        const int64_t total = part[PAR_MAX];
        P.total_need = total * 128;
        if (total <= cap128) {
            par_for(n, [&](size_t b, size_t e, int tid) {
                int64_t used = part[size_t(tid)];
                for (size_t k = b; k < e; k++) { const uint32_t v = off[k]; off[k] = uint32_t(used); used += v; }
            });
            if (n) {
                Chunk ch;
                ch.work_off = 0; ch.count = int32_t(n);
                P.chunks.push_back(std::move(ch));
                P.arena_used = total * 128;
            }
        } else {
            int64_t used = 0;
            size_t k0 = 0;
            for (size_t k = 0; k < n; k++) {
                const int64_t need = off[k];
                if (k > k0 && used + need > cap128) {
                    Chunk ch;
                    ch.work_off = int64_t(k0); ch.count = int32_t(k - k0);
                    P.chunks.push_back(std::move(ch));
                    P.arena_used = std::max(P.arena_used, used * 128);
                    k0 = k; used = 0;
                }
                off[k] = uint32_t(used);
                used += need;
            }
            if (n > k0) {
                Chunk ch;
                ch.work_off = int64_t(k0); ch.count = int32_t(n - k0);
                P.chunks.push_back(std::move(ch));
                P.arena_used = std::max(P.arena_used, used * 128);
            }
        }
    }
    lap("pass 2 (offsets)");
    if (!P.lazy) P.descs.resize(n);
    const size_t n_ch = P.chunks.size();
    if (lv != LV_DENSE)
        for (Chunk &ch : P.chunks) {
            ch.n_long = int32_t(std::min<int64_t>(std::max<int64_t>(int64_t(n_big) - ch.work_off, 0), ch.count));
            // a LV_Q16 plan's long alignments use the 64-cell layout and cannot share a launch with the rest
            if (lv > LV_Q16 && (ch.n_long == ch.count || ch.count < 4096)) ch.n_long = 0;     // nothing to overlap with
        }
    std::vector<Sums> sums(size_t(PAR_MAX) * n_ch);
    if (P.lazy && n_ch == 1) {
        // one chunk, descriptors on demand: the sums are pass 1's (an alignment is in part 0 iff it is long, unless the chunk
        // has no long part), the levels are set; what is left is positions -> alignment ids
        par_for(n, [&](size_t b, size_t e, int) { for (size_t k = b; k < e; k++) order[k] = alns[size_t(order[k])]; });
        Sums &S = sums[0];
        const bool no_long = P.chunks[0].n_long == 0;
        for (const Sums &T : sums1)
            for (int q = 0; q < 2; q++) {
                const int part = no_long ? 1 : q;
                S.part_cells[part] += T.part_cells[q]; S.part_in[part] += T.part_in[q];
                S.part_dense[part] += T.part_dense[q]; S.part_rows[part] += T.part_rows[q];
                S.cells += T.part_cells[q]; S.in_bytes += T.part_in[q];
            }
    } else
    // pass 3 (parallel): positions -> alignment ids, levels, the chunks' sums, and (unless lazy) the descriptors
    par_for(n, [&](size_t b, size_t e, int tid) {
        size_t ci = 0;
        while (ci + 1 < n_ch && size_t(P.chunks[ci + 1].work_off) <= b) ci++;       // (chunks are few)
        for (size_t k = b; k < e; k++) {
            while (ci + 1 < n_ch && size_t(P.chunks[ci + 1].work_off) <= k) ci++;
            const size_t pos = size_t(order[k]);
            const int32_t a = alns[pos];
            order[k] = a;
            AlnDesc d;
            if (P.lazy) { d.Lq = lq[pos]; d.Lr = lr[pos]; d.Lt = lt[pos]; d.path_cap = d.Lq + d.Lr + d.Lt + 4; }   // (all the sums look at)
            else d = h->descs[size_t(a)];
            const int dl = level_of(d);
            const int64_t used = int64_t(off[k]) * 128;
            if (dl != LV_DENSE) {
                (void)window_layout(d, dl, used, tag_or);
            } else {
                d.band_w = 0;
                d.pitch[0] = int32_t(round_up(d.Lq, 32)); d.pitch[1] = int32_t(round_up(d.Lr, 32));
                const int64_t m0 = round_up(int64_t(d.pitch[0]) * d.Lt, 128), m1 = round_up(int64_t(d.pitch[1]) * d.Lt, 128);
                d.band_pad = LV_TAG[dl] | tag_or;
                d.mat_off[0] = used; d.mat_off[1] = used + m0;
                d.blo_off = (used + m0 + m1) / 4;
                d.path_off = (used + m0 + m1) / int64_t(sizeof(PathEnt));
            }
            const int W = LV_WINDOW[dl];
            Sums &S = sums[size_t(tid) * n_ch + ci];
            S.cells += W ? int64_t(std::min(W, d.Lq) + std::min(W, d.Lr)) * d.Lt : int64_t(d.Lq + d.Lr) * d.Lt;
            S.in_bytes += 6 * int64_t(d.Lq) + 6 * int64_t(d.Lt) + 6 * int64_t(d.Lr);
            if (lv != LV_DENSE) {
                const int part = int64_t(k) - P.chunks[ci].work_off < P.chunks[ci].n_long ? 0 : 1;
                S.part_cells[part] += int64_t(std::min(d.band_w, d.Lq) + std::min(d.band_w, d.Lr)) * d.Lt;
                S.part_in[part] += 6 * int64_t(d.Lq) + 6 * int64_t(d.Lt) + 6 * int64_t(d.Lr);
                S.part_dense[part] += int64_t(d.Lq + d.Lr) * d.Lt;
                S.part_rows[part] += d.Lt;
            }
            h->level[size_t(a)] = uint8_t(dl);
            if (!P.lazy) P.descs[k] = d;
        }
    });
    lap("pass 3 (descriptors)");
    for (size_t ci = 0; ci < n_ch; ci++)
        for (int t = 0; t < PAR_MAX; t++) {
            const Sums &S = sums[size_t(t) * n_ch + ci];
            Chunk &ch = P.chunks[ci];
            ch.cells += S.cells;
            ch.in_bytes += S.in_bytes;
            for (int q = 0; q < 2; q++) { ch.part_cells[q] += S.part_cells[q]; ch.part_in[q] += S.part_in[q]; ch.part_dense[q] += S.part_dense[q]; ch.part_rows[q] += S.part_rows[q]; }
        }
    for (Chunk &ch : P.chunks) {
        if (lv == LV_DENSE) {
            // dense plan: group the chunk's alignments by kernel class (stable), one launch per class
            std::vector<int32_t> idx(ch.count);
            for (int32_t w = 0; w < ch.count; w++) idx[w] = w;
            auto cls_of = [&](int32_t w) { const AlnDesc &d = P.descs[ch.work_off + w]; return class_of(std::max(d.Lq, d.Lr)); };
            std::stable_sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) { return cls_of(x) < cls_of(y); });
            std::vector<int32_t> w2(ch.count);
            std::vector<AlnDesc> d2(ch.count);
            for (int32_t w = 0; w < ch.count; w++) { w2[w] = P.work[ch.work_off + idx[w]]; d2[w] = P.descs[ch.work_off + idx[w]]; }
            std::copy(w2.begin(), w2.end(), P.work.begin() + ch.work_off);
            std::copy(d2.begin(), d2.end(), P.descs.begin() + ch.work_off);
            int32_t w = 0;
            while (w < ch.count) {
                const int c = cls_of(w);   // P.descs now in class order
                int32_t e = w;
                while (e < ch.count && class_of(std::max(P.descs[ch.work_off + e].Lq, P.descs[ch.work_off + e].Lr)) == c) e++;
                ch.launches.push_back(Launch{c, ch.work_off + w, e - w});
                w = e;
            }
        }
    }
    return VPR_OK;
}


int zl_finish(vpr_handle *h, size_t n_waves, int64_t in_words, int64_t log_max, int64_t in_chunk_max, int64_t n_short_max, int32_t len_max);

// Zero-distance lane kernel (pr_zl.hip): cut the short part of every chunk of plan 0 into waves of 64 alignments, size
// each wave's interleaved input block and log blocks, and write the position words (k_prep_zl; behind K0 and the
// descriptor scatter on the upload stream).
int prep_zero_lane(vpr_handle *h) {
    const Plan &P = h->plan0;
    std::vector<ZlWave> &hdr = h->zl_hdr_host;      // (kept in the handle: the upload below reads it asynchronously)
    hdr.clear();
    h->zl_wave0.assign(P.chunks.size(), 0);
    int64_t in_words = 0, log_max = 0, in_chunk_max = 0, n_short_max = 0;
    int32_t len_max = 0;
    for (size_t ci = 0; ci < P.chunks.size(); ci++) {
        const Chunk &ch = P.chunks[ci];
        h->zl_wave0[ci] = int64_t(hdr.size());
        const int64_t first = ch.work_off + ch.n_long, n_short = ch.count - ch.n_long;
        const int64_t in_words0 = in_words;
        n_short_max = std::max(n_short_max, n_short);
        const size_t w0 = hdr.size(), nw = size_t((n_short + 63) / 64);
        hdr.resize(w0 + nw);
        par_for(nw * 64, [&](size_t b0, size_t e0, int) {           // the largest lengths of every wave's 64 alignments
            for (size_t wv = (b0 + 63) / 64; wv * 64 < e0; wv++) {
                ZlWave W;
                memset(&W, 0, sizeof(W));
                for (int64_t j = int64_t(wv) * 64; j < std::min<int64_t>(n_short, int64_t(wv) * 64 + 64); j++) {
                    int32_t Lq, Lr, Lt;
                    h->descs.lens(size_t(P.work[size_t(first + j)]), Lq, Lr, Lt);
                    W.mq = std::max(W.mq, Lq); W.mr = std::max(W.mr, Lr); W.mt = std::max(W.mt, Lt);
                }
                hdr[w0 + wv] = W;
            }
        });
        int64_t log_cur = 0;
        for (size_t wv = 0; wv < nw; wv++) {
            ZlWave &W = hdr[w0 + wv];
            W.in_off = in_words;
            W.log_off = log_cur;
            in_words += 64 * (int64_t(W.mq) + W.mr + W.mt);
            log_cur += 80 * int64_t(W.mt);         // per row and lane: 8 flag bytes, a path_ptr word, an 8-byte step
            len_max = std::max(len_max, std::max(W.mq, std::max(W.mr, W.mt)));
        }
        log_max = std::max(log_max, log_cur);
        in_chunk_max = std::max(in_chunk_max, in_words - in_words0);
    }
    if (hdr.empty()) return VPR_OK;
    int rc;
    if ((rc = dev_alloc(h, &h->d_zl_hdr, hdr.size()))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_zl_hdr, hdr.data(), hdr.size() * sizeof(ZlWave), hipMemcpyHostToDevice, h->stream));
    return zl_finish(h, hdr.size(), in_words, log_max, in_chunk_max, n_short_max, len_max);
}

// the lane levels' blocks for wave headers that are on the device (h->d_zl_hdr, h->zl_wave0 set): position words, logs, the
// distance-1 level's blocks; then the position words are written (k_prep_zl, on the upload stream)
int zl_finish(vpr_handle *h, size_t n_waves, int64_t in_words, int64_t log_max, int64_t in_chunk_max, int64_t n_short_max, int32_t len_max) {
    const Plan &P = h->plan0;
    int rc;
    if ((rc = dev_alloc(h, &h->d_zl_in, size_t(in_words)))) return rc;
    if ((rc = dev_alloc(h, &h->d_zl_log, size_t(log_max)))) return rc;
    // The distance-1 lane level works on the zero level's rejects (pr_d1.hip): its blocks hold D1_SHARE of what the largest
    // chunk would need if EVERY alignment were rejected (two words per QUERY / REF position instead of one, 40 B of log per
    // row instead of 20), at least D1_FLOOR -- small batches then fit whole --; waves that do not fit stay with the
    // in-place 16-cell round.  VPR_NO_D1 switches the level off.
    if (!getenv("VPR_NO_D1")) {
        if (const char *e = getenv("VPR_D1_MAX_ROWS")) h->d1_max_rows = std::max(2, atoi(e));
        const double D1_SHARE = 0.4;
        const int64_t D1_FLOOR = int64_t(4) << 20;
        const int64_t slack = 64 * 5 * int64_t(len_max + 1);
        const int64_t in_all = 2 * in_chunk_max + slack, log_all = 2 * log_max + 160 * int64_t((n_short_max + 63) / 64) + 160 * int64_t(len_max + 1);
        h->d1_in_cap = std::min(in_all, std::max(int64_t(D1_SHARE * double(in_all)), D1_FLOOR + slack));
        h->d1_log_cap = std::min(log_all, std::max(int64_t(D1_SHARE * double(log_all)), D1_FLOOR + slack));
        const int64_t cap_ip = std::min<int64_t>(n_short_max, std::max<int64_t>(4096, (n_short_max / 4 + 3) & ~int64_t(3)));
        h->d1_wave_cap = int32_t((cap_ip + 63) / 64);
        h->d1_fail_cap = int32_t(n_short_max + n_short_max / 8 + 1024);
        // (the level is optional -- VPR_NO_D1 gives the same results --, so its blocks, several GB on a million superclusters, are
        // optional growth: where the device has no room for them the upload goes on without the level.  ADVICE r5)
        h->soft_alloc = true;
        rc = dev_alloc(h, &h->d_d1_hdr, size_t(h->d1_wave_cap));
        if (!rc) rc = dev_alloc(h, &h->d_d1_in, size_t(h->d1_in_cap));
        if (!rc) rc = dev_alloc(h, &h->d_d1_log, size_t(h->d1_log_cap));
        if (!rc) rc = dev_alloc(h, &h->d_d1_fail, size_t(h->d1_fail_cap));
        if (!rc) rc = dev_alloc(h, &h->d_d1_info, 16);
        if (!rc) rc = dev_alloc(h, &h->d_d1_blk, size_t(h->d1_fail_cap / 256 + 8));
        h->soft_alloc = false;
        if (rc) {
            if (h->debug) fprintf(stderr, "[vpr] distance-1 lane level: no room for its blocks (%.2f GB), the level is off for this batch\n",
                                  double(h->d1_in_cap) * 4e-9 + double(h->d1_log_cap) * 16e-9);
            h->d_d1_hdr = nullptr; h->d_d1_in = nullptr; h->d_d1_log = nullptr; h->d_d1_fail = nullptr; h->d_d1_info = nullptr; h->d_d1_blk = nullptr;
            h->err.clear();
            (void)hipGetLastError();
        } else {
            HIPCHK(h, hipMemsetAsync(h->d_d1_info, 0, 64, h->stream));
        }
    }
    for (size_t ci = 0; ci < P.chunks.size(); ci++) {
        const Chunk &ch = P.chunks[ci];
        const int64_t n_short = ch.count - ch.n_long;
        if (n_short <= 0) continue;
        hipLaunchKernelGGL(k_prep_zl, dim3(unsigned((n_short + 63) / 64)), dim3(256), 0, h->stream, h->dB, h->d_descs,
                           P.d_work + ch.work_off + ch.n_long, int(n_short), h->d_zl_hdr + h->zl_wave0[ci], h->d_zl_in);
    }
    HIPCHK(h, hipGetLastError());
    if (h->debug) fprintf(stderr, "[vpr] zero-distance lane level: %zu waves, %.2f GB position words, %.2f GB log\n", n_waves,
                          double(in_words) * 4e-9, double(log_max) * 16e-9);
    return VPR_OK;
}

}  // namespace

namespace {
// ---------------------------------------------------------------------------
// Round 0's plan of a whole batch built ON THE DEVICE (the usual case: every alignment of the batch, in input order, 16-cell
// layout for the short part, everything in one chunk).  What make_plan does on the host for four million alignments -- a
// counting sort by rows, the prefix sums of the workspace needs in plan order, the wave headers of the lane level: 40 of the
// 60 ms a host thread spends per uploaded batch -- is a radix sort, a scan and two small kernels here (pr_plan.hip, k_plan_*,
// k_zl_*), on a stream of their own beside the K0 kernels; the host keeps one light pass (levels, sums, the few thousand long
// alignments, the checks), and gets the order, the offsets and the inverse back with three copies.  Results identical to
// make_plan's (tests: VPR_HOST_PLAN=1 selects the host planner; the arrays are compared).
//   ev_off: recorded on the upload stream behind the copies of the batch's offsets.  *done: false = not applicable here
//   (several chunks, an alignment that does not fit): the caller plans on the host.
// ---------------------------------------------------------------------------
int plan0_device(vpr_handle *h, int lv0, hipEvent_t ev_off, bool *done) {
    *done = false;
    const size_t na = h->descs.size();
    if (na == 0 || na > size_t(0x7fffff00) || lv0 > LV_Q16) return VPR_OK;
    Plan &P = h->plan0;
    const int LLT = h->long_lt;
    const int64_t cap128 = h->arena_bytes / 128;
    // ---- what the upload's pass over the superclusters left (vpr_handle::Plan0Pass): levels are set, sums and the long list here
    if (!h->p0.valid) return VPR_OK;
    const vpr_handle::Plan0Pass &S0 = h->p0;
    const int64_t total = S0.need128;
    if (S0.bad || S0.need_max > h->arena_bytes || total > cap128 || total > int64_t(0xfffffff0)) {      // (several chunks / an error message: the host planner)
        h->level.assign(na, uint8_t(LV_DENSE));
        return VPR_OK;
    }
    std::vector<std::pair<int64_t, int32_t>> big = S0.big;
    std::sort(big.begin(), big.end());
    const size_t n = na, n_big = big.size(), n_short = n - n_big;
    // ---- the plan's fields
    {
        std::vector<int32_t> kw; std::vector<uint32_t> ko;
        kw.swap(P.work); ko.swap(P.off128);
        P = Plan();
        P.work.swap(kw); P.off128.swap(ko);
    }
    P.lv = lv0; P.arena = h->d_arena; P.tag_or = 0; P.long_lt = LLT; P.lazy = true;
    P.total_need = total * 128; P.arena_used = total * 128;
    {
        Chunk ch;
        ch.work_off = 0; ch.count = int32_t(n);
        ch.n_long = int32_t(std::min<size_t>(n_big, n));
        const bool no_long = ch.n_long == 0;
        for (int q = 0; q < 2; q++) {
            const int part = no_long ? 1 : q;
            ch.part_cells[part] += S0.part_cells[q]; ch.part_in[part] += S0.part_in[q];
            ch.part_dense[part] += S0.part_dense[q]; ch.part_rows[part] += S0.part_rows[q];
            ch.cells += S0.part_cells[q]; ch.in_bytes += S0.part_in[q];
        }
        P.chunks.push_back(std::move(ch));
    }
    // ---- device: keys -> order -> offsets -> inverse -> wave headers
    int rc;
    hipStream_t ps = h->cls_stream[0];          // (the class streams are idle during an upload)
    uint16_t *d_key = nullptr, *d_key2 = nullptr;
    int32_t *d_val = nullptr, *d_pos = nullptr;
    uint32_t *d_need = nullptr, *d_need2 = nullptr, *d_off = nullptr;
    int64_t *d_tot = nullptr;
    if ((rc = dev_alloc(h, &P.d_descs, n))) return rc;
    if ((rc = dev_alloc(h, &P.d_work, n))) return rc;
    if ((rc = dev_alloc(h, &d_key, n))) return rc;
    if ((rc = dev_alloc(h, &d_key2, n))) return rc;
    if ((rc = dev_alloc(h, &d_val, n))) return rc;
    if ((rc = dev_alloc(h, &d_pos, n))) return rc;
    if ((rc = dev_alloc(h, &d_need, n))) return rc;
    if ((rc = dev_alloc(h, &d_need2, n))) return rc;
    if ((rc = dev_alloc(h, &d_off, n))) return rc;
    if ((rc = dev_alloc(h, &d_tot, 4))) return rc;
    size_t tmp_sort = 0, tmp_scan = 0;
    if (vplan_sort_pairs_desc(nullptr, &tmp_sort, d_key, d_key2, d_val, P.d_work, n, ps) != 0 ||
        vplan_exclusive_scan_u32(nullptr, &tmp_scan, d_need2, d_off, n, ps) != 0)
        return fail(h, VPR_ERR_DEVICE, "device planner: workspace query failed");
    uint8_t *d_tmp = nullptr;
    if ((rc = dev_alloc(h, &d_tmp, std::max(tmp_sort, tmp_scan) + 256))) return rc;
    void *pin = nullptr;
    { int rc_pin = pin_alloc(h, &pin, n * 12 + n_big * 4 + 64); if (rc_pin) return rc_pin; }
    int32_t *hp_work = static_cast<int32_t *>(pin);
    uint32_t *hp_off = reinterpret_cast<uint32_t *>(hp_work + n);
    int32_t *hp_pos = reinterpret_cast<int32_t *>(hp_off + n);
    int32_t *hp_big = hp_pos + n;
    int64_t *hp_tot = reinterpret_cast<int64_t *>((reinterpret_cast<uintptr_t>(hp_big + n_big) + 15) & ~uintptr_t(15));
    for (size_t k = 0; k < n_big; k++) hp_big[k] = big[k].second;
    BatchOffsets DO;
    for (int s_ = 0; s_ < 4; s_++) { DO.hap_off[s_] = h->dB.hap_off[s_]; DO.var_off[s_] = h->dB.var_off[s_]; }
    DO.ref_off = h->dB.ref_off;
    auto blocks = [](int64_t n_) { return dim3(unsigned((n_ + 255) / 256)); };
    HIPCHK(h, hipStreamWaitEvent(ps, ev_off, 0));
    hipLaunchKernelGGL(k_plan_keys, blocks(int64_t(n)), dim3(256), 0, ps, DO, int(n), lv0, LLT, d_key, d_val, d_need);
    if (vplan_sort_pairs_desc(d_tmp, &tmp_sort, d_key, d_key2, d_val, P.d_work, n, ps) != 0) return fail(h, VPR_ERR_DEVICE, "device planner: sort failed");
    if (n_big) HIPCHK(h, hipMemcpyAsync(P.d_work, hp_big, n_big * 4, hipMemcpyHostToDevice, ps));      // the long part in the host's order
    hipLaunchKernelGGL(k_plan_gather, blocks(int64_t(n)), dim3(256), 0, ps, P.d_work, d_need, d_need2, d_pos, int(n));
    if (vplan_exclusive_scan_u32(d_tmp, &tmp_scan, d_need2, d_off, n, ps) != 0) return fail(h, VPR_ERR_DEVICE, "device planner: scan failed");
    HIPCHK(h, hipMemcpyAsync(hp_work, P.d_work, n * 4, hipMemcpyDeviceToHost, ps));
    HIPCHK(h, hipMemcpyAsync(hp_off, d_off, n * 4, hipMemcpyDeviceToHost, ps));
    HIPCHK(h, hipMemcpyAsync(hp_pos, d_pos, n * 4, hipMemcpyDeviceToHost, ps));
    const size_t nw = lv0 == LV_Z ? (n_short + 63) / 64 : 0;
    if (nw) {
        if ((rc = dev_alloc(h, &h->d_zl_hdr, nw))) return rc;
        hipLaunchKernelGGL(k_zl_hdr, dim3(unsigned(nw)), dim3(64), 0, ps, DO, P.d_work + n_big, int(n_short), h->d_zl_hdr);
        hipLaunchKernelGGL(k_zl_scan, dim3(1), dim3(1024), 0, ps, h->d_zl_hdr, int(nw), d_tot);
        HIPCHK(h, hipMemcpyAsync(hp_tot, d_tot, 32, hipMemcpyDeviceToHost, ps));
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, x_sync(h, ps, SITE));
    // ---- the host's copies (plan_desc, the tie rounds and vpr_download_path look single entries up)
    P.work.resize(n); P.off128.resize(n);
    h->plan0_pos.resize(na);
    par_for(n, [&](size_t b0, size_t e0, int) {
        memcpy(P.work.data() + b0, hp_work + b0, (e0 - b0) * 4);
        memcpy(P.off128.data() + b0, hp_off + b0, (e0 - b0) * 4);
        memcpy(h->plan0_pos.data() + b0, hp_pos + b0, (e0 - b0) * 4);
    });
    // ---- descriptors in both orders; the lane level's blocks and position words (behind K0 on the upload stream)
    hipLaunchKernelGGL(k_build_plan, blocks(int64_t(n)), dim3(256), 0, h->stream, DO, P.d_work, d_off, int(n), P.lv, P.long_lt, P.tag_or,
                       P.d_descs, h->d_descs);
    if (nw) {
        h->zl_wave0.assign(1, 0);
        const int64_t in_words = hp_tot[0], log_max = hp_tot[1];
        if ((rc = zl_finish(h, nw, in_words, log_max, in_words, int64_t(n_short), int32_t(hp_tot[2])))) return rc;
    }
    *done = true;
    return VPR_OK;
}
}  // namespace


extern "C" {

const char *vpr_last_error(const vpr_handle *h) { return h ? h->err.c_str() : vpr_create_error(); }

/* The tie replay (pr_tie.hip) reproduces the iteration order of the reference's std::unordered_set from the bucket counts the
   container grows through.  Those counts are a property of the libstdc++ the REFERENCE would be built against on this machine
   (_Prime_rehash_policy lives in the shared library): measured here on the running library and compared with the model's table,
   so that a box with a different libstdc++ fails loudly instead of silently diverging from the reference on tied alignments.
   Returns how many leading entries of the table were checked (all equal), or -(1 + index of the first differing entry). */
int vpr_selfcheck_tie_model(int max_entries) {
    std::unordered_set<int> t;
    t.insert(0);
    uint64_t B = t.bucket_count();
    int k = 0;
    while (k < TIE_N_BUCKETS && k < max_entries) {
        if (B != TIE_BUCKETS_HOST[k]) return -(1 + k);
        k++;
        if (k >= TIE_N_BUCKETS || k >= max_entries) break;
        std::unordered_set<int> u;
        u.rehash(2 * B);
        B = u.bucket_count();
    }
    return k;
}

int vpr_create(const vpr_config *cfg, vpr_handle **out) {
    if (!cfg || !out) return fail(nullptr, VPR_ERR_ARG, "vpr_create: null argument");
    *out = nullptr;
    {
        static const int tie_model = vpr_selfcheck_tie_model(14);       // (once per process: up to 172 933 buckets, 1.4 MB)
        if (tie_model < 0)
            return fail(nullptr, VPR_ERR_STATE, "this machine's libstdc++ grows std::unordered_set through other bucket counts than the tie "
                        "replay's model (entry %d of TIE_BUCKET_LIST, pr_tie.hip): results on alignments with tied swap predecessors "
                        "would differ from a reference built here", -tie_model - 1);
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, VPR_ERR_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, VPR_ERR_ARG, "device %d out of range (have %d)", cfg->device, ndev);
    vpr_handle *h = new vpr_handle();
    h->cfg = *cfg;
    h->debug = getenv("VPR_DEBUG") != nullptr;
    h->no_strips = getenv("VPR_NO_STRIPS") != nullptr;
    h->no_round_overlap = getenv("VPR_NO_ROUND_OVERLAP") != nullptr;
    h->stall_log = getenv("VPR_STALL_LOG") != nullptr;
    h->seq_walk = getenv("VPR_SEQ_WALK") != nullptr;
    h->no_flag_save = getenv("VPR_NO_FLAG_SAVE") != nullptr;
    h->side_credit = exp_getenv("VPR_NO_SIDE_CREDIT") == nullptr;
    if (const char *e = exp_getenv("VPR_ZL_LDS_KB")) h->zl_lds_bytes = std::max(0, std::min(64, atoi(e))) * 1024;
    if (const char *e = exp_getenv("VPR_LANE_PRIO_ROWS")) h->lane_prio_rows = std::max(1, atoi(e));     // diagnostic
    for (int k = 0; k < 2; k++)
        if (hipEventCreateWithFlags(&h->ev_cred[k], hipEventDisableTiming) != hipSuccess) return fail(nullptr, VPR_ERR_DEVICE, "hipEventCreate failed");
    if (const char *e = getenv("VPR_LONG_LT")) { const int v = atoi(e); if (v >= 64 && v <= 2048) h->long_lt = v; }     // diagnostic
    memset(&h->dB, 0, sizeof(h->dB));
    memset(&h->timing, 0, sizeof(h->timing));
    // (diagnostic) VPR_STREAM_PRIO: one letter per stream -- class streams 0..7, tie streams 0..3, the handle's main stream --
    // h(igh), n(ormal), l(ow)
    const char *prio_map = exp_getenv("VPR_STREAM_PRIO");
    if (prio_map && strlen(prio_map) != size_t(N_CLASSES + 5)) prio_map = nullptr;
    if (hipSetDevice(cfg->device) != hipSuccess) { delete h; return fail(nullptr, VPR_ERR_DEVICE, "hipSetDevice failed"); }
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    auto prio_of = [&](int idx, int dflt) { return !prio_map ? dflt : (prio_map[idx] == 'h' ? prio_hi : (prio_map[idx] == 'l' ? prio_lo : (prio_hi + prio_lo) / 2)); };
    // (non-blocking: a stream that synchronises with the null stream takes a process-wide lock on every launch -- with three host
    // threads uploading batches that cost the one-pass leg 5 - 7 %; the library never relies on the null stream.  VPR_MAIN_BLOCKING: as before)
    if (hipStreamCreateWithPriority(&h->stream, exp_getenv("VPR_MAIN_BLOCKING") ? hipStreamDefault : hipStreamNonBlocking, prio_of(N_CLASSES + 4, (prio_hi + prio_lo) / 2)) != hipSuccess) {
        delete h;
        return fail(nullptr, VPR_ERR_DEVICE, "hipStreamCreate failed");
    }
    {   // bounds of what grows on demand during an execute: fractions of the DEVICE's memory, fixed here
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { delete h; return fail(nullptr, VPR_ERR_DEVICE, "hipMemGetInfo failed"); }
        h->dev_total = int64_t(total_b);
        h->tie_scratch_max = std::max<int64_t>(h->dev_total / 16, int64_t(64) << 20);
        h->lad_arena_max = std::max<int64_t>(h->dev_total / 8, int64_t(64) << 20);
    }
    // Streams 0, 2, 3 carry latency chains (long alignments, retry ladders) and get the highest priority, so their
    // few workgroups are dispatched ahead of the millions of the bulk stream (1) instead of behind them.
    if (const char *e = exp_getenv("VPR_STREAM_PAD")) {
        // (diagnostic) "lo:hi,lo:hi,...": the i-th handle of the process creates that many idle streams of either priority
        // before its own, which shifts the hardware queues the runtime maps its streams onto
        static std::atomic<int> n_handles{0};
        const int me = n_handles.fetch_add(1);
        const char *p = e;
        for (int i = 0; i < me && p; i++) { p = strchr(p, ','); if (p) p++; }
        if (p) {
            const int n_lo = atoi(p);
            const char *c = strchr(p, ':');
            const char *stop = strchr(p, ',');
            const int n_hi = (c && (!stop || c < stop)) ? atoi(c + 1) : 0;
            for (int k = 0; k < n_lo + n_hi && k < 64; k++) {
                hipStream_t st = nullptr;
                if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, k < n_lo ? prio_lo : prio_hi) == hipSuccess) h->pad_streams.push_back(st);
            }
        }
    }
    for (int k = 0; k < N_CLASSES; k++) {
        // (tried: the retry rounds' side streams 5, 6 at high priority -- their back halves then start 2 ms earlier and the step
        // with two batches in flight gets 2 ms LONGER: they take the bulk kernels' slots)
        const int prio = prio_of(k, (k == 0 || k == 2 || k == 3) ? prio_hi : prio_lo);
        if (hipStreamCreateWithPriority(&h->cls_stream[k], hipStreamNonBlocking, prio) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_join[k], hipEventDisableTiming) != hipSuccess) {
            return fail(nullptr, VPR_ERR_DEVICE, "hipStreamCreate failed");
        }
    }
    for (int k = 0; k < 2 + 4 * LadderCtx::N_SLOTS; k++)
        if (hipEventCreateWithFlags(&h->ev_slot[k], hipEventDisableTiming) != hipSuccess)
            return fail(nullptr, VPR_ERR_DEVICE, "hipEventCreate failed");
    for (int k = 0; k < 4; k++)
        if (hipStreamCreateWithPriority(&h->tie_stream[k], hipStreamNonBlocking, prio_of(N_CLASSES + k, prio_hi)) != hipSuccess)
            return fail(nullptr, VPR_ERR_DEVICE, "hipStreamCreate failed");
    for (int k = 0; k < 2; k++)
        if (hipEventCreateWithFlags(&h->ev_tie[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_tie2[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_side[k], hipEventDisableTiming) != hipSuccess)
            return fail(nullptr, VPR_ERR_DEVICE, "hipEventCreate failed");
    if (hipEventCreateWithFlags(&h->ev_spec, hipEventDisableTiming) != hipSuccess) return fail(nullptr, VPR_ERR_DEVICE, "hipEventCreate failed");
    if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess)
        return fail(nullptr, VPR_ERR_DEVICE, "hipEventCreate failed");
    // allow the big dense classes to use the whole 160 KiB LDS of a CU
    for (int k = 0; k < N_CLASSES; k++) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fwd_kernel(k)),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(LDS_MAX));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bwd_kernel(k)),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(LDS_MAX));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bwd_kernel(k, true)),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(LDS_MAX));
    }
    *out = h;
    return VPR_OK;
}

void vpr_destroy(vpr_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    free_batch(h);
    for (auto &b : h->dev_cache) (void)x_free(h, b.p, SITE);
    for (int k = 0; k < 4; k++)
        for (int e = 0; e < 2; e++) if (h->lad[k].tie_scratch[e]) (void)x_free(h, h->lad[k].tie_scratch[e], SITE);
    for (auto &b : h->pin_cache) (void)hipHostFree(b.p);
    if (h->d_ctg_seq) (void)x_free(h, h->d_ctg_seq, SITE);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    for (int k = 0; k < 8; k++) {
        if (h->cls_stream[k]) (void)hipStreamDestroy(h->cls_stream[k]);
        if (h->ev_join[k]) (void)hipEventDestroy(h->ev_join[k]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    for (int k = 0; k < 2 + 4 * LadderCtx::N_SLOTS; k++)
        if (h->ev_slot[k]) (void)hipEventDestroy(h->ev_slot[k]);
    for (int k = 0; k < 4; k++) if (h->tie_stream[k]) (void)hipStreamDestroy(h->tie_stream[k]);
    for (hipStream_t st : h->pad_streams) (void)hipStreamDestroy(st);
    for (int k = 0; k < 2; k++) {
        if (h->ev_tie[k]) (void)hipEventDestroy(h->ev_tie[k]);
        if (h->ev_tie2[k]) (void)hipEventDestroy(h->ev_tie2[k]);
        if (h->ev_side[k]) (void)hipEventDestroy(h->ev_side[k]);
    }
    if (h->ev_spec) (void)hipEventDestroy(h->ev_spec);
    if (h->ev_offsets) (void)hipEventDestroy(h->ev_offsets);
    for (int k = 0; k < 2; k++) if (h->ev_cred[k]) (void)hipEventDestroy(h->ev_cred[k]);
    delete h;
}

int vpr_upload(vpr_handle *h, const vpr_batch *b) {
    if (!h || !b) return VPR_ERR_ARG;
    if (b->n_sc < 0) return fail(h, VPR_ERR_ARG, "negative n_sc");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const bool dbg = h->debug;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double T0 = now();
    const bool tim = getenv("VPR_TIMING") != nullptr;       // host-side laps without device synchronisation
    auto lap = [&](const char *what) {
        if (dbg) { (void)hipDeviceSynchronize(); fprintf(stderr, "[vpr] upload %-28s %.3f s\n", what, now() - T0); }
        else if (tim) fprintf(stderr, "[vpr] upload (host) %-28s %.3f s\n", what, now() - T0);
    };
    free_batch(h);
    lap("previous batch released");
    const int n = b->n_sc;
    h->n_sc = n;
    DevBatch &D = h->dB;
    memset(&D, 0, sizeof(D));
    D.n_sc = n;
    int rc;
    int64_t hap_len[4], ref_len = b->ref_off[n];
    for (int s = 0; s < 4; s++) {
        hap_len[s] = b->hap_off[s][n];
        h->n_var[s] = b->var_off[s][n];
        if ((rc = dev_upload(h, &D.hap_off[s], b->hap_off[s], n + 1))) return rc;
        if ((rc = dev_upload(h, &D.hap_seq[s], b->hap_seq[s], hap_len[s]))) return rc;
        if ((rc = dev_upload(h, &D.hap_ptr[s], b->hap_ptr[s], hap_len[s]))) return rc;
        if ((rc = dev_upload(h, &D.hap_flag[s], b->hap_flag[s], hap_len[s]))) return rc;
        if ((rc = dev_upload(h, &D.var_off[s], b->var_off[s], n + 1))) return rc;
        if ((rc = dev_upload(h, &D.var_pos[s], b->var_pos[s], h->n_var[s]))) return rc;
        h->var_off[s].assign(b->var_off[s], b->var_off[s] + n + 1);
        h->var_qual[s].assign(b->var_qual[s], b->var_qual[s] + h->n_var[s]);
        if ((rc = dev_alloc(h, &D.has_ins[s], ref_len))) return rc;
        if ((rc = dev_alloc(h, &D.vs_hap[s], hap_len[s]))) return rc;
        if ((rc = dev_alloc(h, &D.dspan[s], size_t(n)))) return rc;
        if ((rc = dev_alloc(h, &D.sc_hap[s], hap_len[s]))) return rc;
        HIPCHK(h, hipMemsetAsync(D.has_ins[s], 0, std::max<int64_t>(ref_len, 1), h->stream));
        if (s == 0) {
            if ((rc = dev_alloc(h, &D.sc_limit, size_t(std::max(b->n_sc, 1))))) return rc;
            HIPCHK(h, hipMemsetAsync(D.sc_limit, 0, size_t(std::max(b->n_sc, 1)), h->stream));
        }
    }
    lap("  inputs: haplotype arrays");
    if ((rc = dev_upload(h, &D.ref_off, b->ref_off, n + 1))) return rc;
    if ((rc = dev_upload(h, &D.ref_seq, b->ref_seq, ref_len))) return rc;
    if ((rc = dev_alloc(h, &D.sc_ref, ref_len))) return rc;
    for (int q = 0; q < 2; q++) {
        if ((rc = dev_upload(h, &D.ref_ptr[q], b->ref_ptr[q], ref_len))) return rc;
        if ((rc = dev_upload(h, &D.ref_flag[q], b->ref_flag[q], ref_len))) return rc;
        if ((rc = dev_alloc(h, &D.cand_q[q], hap_len[q]))) return rc;
        if ((rc = dev_alloc(h, &D.cand_r[q], ref_len))) return rc;
        if ((rc = dev_alloc(h, &D.cand2_q[q], hap_len[q]))) return rc;
        if ((rc = dev_alloc(h, &D.cand2_r[q], ref_len))) return rc;
        if ((rc = dev_alloc(h, &D.fk_q[q], hap_len[q]))) return rc;
        if ((rc = dev_alloc(h, &D.fk_r[q], ref_len))) return rc;
        if ((rc = dev_alloc(h, &D.bk_q[q], hap_len[q]))) return rc;
        if ((rc = dev_alloc(h, &D.bk_r[q], ref_len))) return rc;
        if ((rc = dev_alloc(h, &D.vs_ref[q], ref_len))) return rc;
        if ((rc = dev_alloc(h, &D.fk4_q[q], hap_len[q]))) return rc;
        if ((rc = dev_alloc(h, &D.fk4_r[q], ref_len))) return rc;
        if ((rc = dev_alloc(h, &D.tk[q], hap_len[2 + q]))) return rc;
        if ((rc = dev_alloc(h, &D.tj[q], hap_len[2 + q] + 8))) return rc;
        if ((rc = dev_alloc(h, &D.wk_q[q], hap_len[q]))) return rc;
        if ((rc = dev_alloc(h, &D.wk_r[q], ref_len))) return rc;
        if ((rc = dev_alloc(h, &D.wk_t[q], hap_len[2 + q]))) return rc;
        if ((rc = dev_alloc(h, &D.xb_q[q], hap_len[q]))) return rc;
        if ((rc = dev_alloc(h, &D.xb_r[q], ref_len))) return rc;
        HIPCHK(h, hipMemsetAsync(D.cand_q[q], 0xff, std::max<int64_t>(hap_len[q], 1) * sizeof(int4), h->stream));
        HIPCHK(h, hipMemsetAsync(D.cand_r[q], 0xff, std::max<int64_t>(ref_len, 1) * sizeof(int4), h->stream));
        HIPCHK(h, hipMemsetAsync(D.cand2_q[q], 0xff, std::max<int64_t>(hap_len[q], 1) * sizeof(int4), h->stream));
        HIPCHK(h, hipMemsetAsync(D.cand2_r[q], 0xff, std::max<int64_t>(ref_len, 1) * sizeof(int4), h->stream));
    }
    lap("  inputs: reference arrays, constants' blocks");
    if ((rc = dev_alloc(h, &h->d_err, 1))) return rc;
    HIPCHK(h, hipMemsetAsync(h->d_err, 0, 4, h->stream));
    if (!h->ev_offsets) HIPCHK(h, hipEventCreateWithFlags(&h->ev_offsets, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(h->ev_offsets, h->stream));       // the batch's offsets are on the device: what the device planner reads
    if (h->gen_src) {       // the strings, pointers and flags are written on the device from the variant tables (pr_gen.hip)
        const vpr_variants *v = h->gen_src;
        GenTables G;
        GenOut O;
        memset(&G, 0, sizeof(G));
        const int64_t ctg_total = v->ctg_off[v->n_ctg];
        if (ctg_total > h->ctg_bytes) {
            if (h->d_ctg_seq) (void)x_free(h, h->d_ctg_seq, SITE);
            h->d_ctg_seq = nullptr; h->ctg_bytes = 0;
            void *q = nullptr;
            if (x_malloc(h, &q, size_t(ctg_total) + 256, SITE) != hipSuccess) return fail(h, VPR_ERR_NOMEM, "contig sequence (%lld bytes)", (long long)ctg_total);
            h->d_ctg_seq = static_cast<uint8_t *>(q);
            h->ctg_bytes = ctg_total;
        }
        if (ctg_total) HIPCHK(h, hipMemcpyAsync(h->d_ctg_seq, v->ctg_seq, size_t(ctg_total), hipMemcpyHostToDevice, h->stream));
        G.ctg_seq = h->d_ctg_seq;
        if ((rc = dev_upload(h, &G.ctg_off, v->ctg_off, size_t(v->n_ctg) + 1))) return rc;
        if ((rc = dev_upload(h, &G.sc_ctg, v->sc_ctg, size_t(n)))) return rc;
        if ((rc = dev_upload(h, &G.sc_beg, v->sc_beg, size_t(n)))) return rc;
        if ((rc = dev_upload(h, &G.sc_end, v->sc_end, size_t(n)))) return rc;
        for (int s = 0; s < 4; s++) {
            const size_t nv = size_t(h->n_var[s]);
            std::vector<int64_t> pl(PAR_MAX, 1);
            par_for(nv, [&](size_t b0, size_t e0, int tid) {
                int64_t m = 1;
                for (size_t k = b0; k < e0; k++)
                    m = std::max(m, std::max(v->var_ref_off[s][k] + v->var_ref_len[s][k], v->var_alt_off[s][k] + v->var_alt_len[s][k]));
                pl[size_t(tid)] = m;
            });
            int64_t pool_len = 1;
            for (int64_t m : pl) pool_len = std::max(pool_len, m);
            G.var_pos_rel[s] = D.var_pos[s];
            if ((rc = dev_upload(h, &G.var_type[s], v->var_type[s], nv))) return rc;
            if ((rc = dev_upload(h, &G.ref_off[s], v->var_ref_off[s], nv))) return rc;
            if ((rc = dev_upload(h, &G.alt_off[s], v->var_alt_off[s], nv))) return rc;
            if ((rc = dev_upload(h, &G.ref_len[s], v->var_ref_len[s], nv))) return rc;
            if ((rc = dev_upload(h, &G.alt_len[s], v->var_alt_len[s], nv))) return rc;
            if ((rc = dev_upload(h, &G.pool[s], v->allele_pool[s], size_t(pool_len)))) return rc;
            O.hap_seq[s] = const_cast<uint8_t *>(D.hap_seq[s]); O.hap_ptr[s] = const_cast<int32_t *>(D.hap_ptr[s]);
            O.hap_flag[s] = const_cast<uint8_t *>(D.hap_flag[s]);
        }
        O.ref_seq = const_cast<uint8_t *>(D.ref_seq);
        for (int q = 0; q < 2; q++) { O.ref_ptr[q] = const_cast<int32_t *>(D.ref_ptr[q]); O.ref_flag[q] = const_cast<uint8_t *>(D.ref_flag[q]); }
        if (n > 0) hipLaunchKernelGGL(k_generate, dim3(unsigned((n + 4 * GEN_SC_PER_WAVE - 1) / (4 * GEN_SC_PER_WAVE)), 4), dim3(256), 0, h->stream, D, G, O);
        HIPCHK(h, hipGetLastError());
    }
    lap("inputs copied");

    // ---- K0: position attributes
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0));
    HIPCHK(h, hipEventCreate(&e1));
    HIPCHK(h, hipEventRecord(e0, h->stream));
    auto blocks = [](int64_t n_) { return dim3(unsigned((n_ + 255) / 256)); };
    if (n > 0) hipLaunchKernelGGL(k_prep_scof, dim3(unsigned((n + 255) / 256), 5), dim3(256), 0, h->stream, D);
    for (int q = 0; q < 2; q++) {
        if (hap_len[q] > 0)
            hipLaunchKernelGGL(k_prep_cand, blocks(hap_len[q]), dim3(256), 0, h->stream, D, q, 0, hap_len[q], h->d_err);
        if (ref_len > 0)
            hipLaunchKernelGGL(k_prep_cand, blocks(ref_len), dim3(256), 0, h->stream, D, q, 1, ref_len, h->d_err);
    }
    for (int q = 0; q < 2; q++) {
        if (hap_len[q] > 0) hipLaunchKernelGGL(k_prep_pack, blocks(hap_len[q]), dim3(256), 0, h->stream, D, q, 0, hap_len[q]);
        if (ref_len > 0) hipLaunchKernelGGL(k_prep_pack, blocks(ref_len), dim3(256), 0, h->stream, D, q, 1, ref_len);
    }
    lap("  prep: scof, cand, pack");
    for (int s = 0; s < 4; s++)
        if (hap_len[s] > 0)
            hipLaunchKernelGGL(k_prep_ins, blocks(hap_len[s]), dim3(256), 0, h->stream, D, s, hap_len[s]);
    if (n > 0) hipLaunchKernelGGL(k_prep_suffix, dim3((n + 4 * SUFFIX_SC_PER_WAVE - 1) / (4 * SUFFIX_SC_PER_WAVE), 6), dim3(256), 0, h->stream, D);
    {
        void *pd = nullptr;
        { int rc_pin = pin_alloc(h, &pd, size_t(std::max(n, 1)) * 4 * sizeof(int2)); if (rc_pin) return rc_pin; }
        for (int s = 0; s < 4; s++) {
            h->hp_dspan[s] = static_cast<int2 *>(pd) + size_t(s) * size_t(n);
            if (n > 0) HIPCHK(h, hipMemcpyAsync(h->hp_dspan[s], D.dspan[s], size_t(n) * sizeof(int2), hipMemcpyDeviceToHost, h->stream));
        }
    }
    lap("  prep: ins, suffix, spans");
    for (int q = 0; q < 2; q++) {
        if (hap_len[q] > 0) hipLaunchKernelGGL(k_prep_xb, blocks(hap_len[q]), dim3(256), 0, h->stream, D, q, 0, hap_len[q]);
        if (ref_len > 0) hipLaunchKernelGGL(k_prep_xb, blocks(ref_len), dim3(256), 0, h->stream, D, q, 1, ref_len);
    }
    for (int s = 0; s < 2; s++)
        if (hap_len[2 + s] > 0) hipLaunchKernelGGL(k_prep_tj, blocks(hap_len[2 + s]), dim3(256), 0, h->stream, D, s, hap_len[2 + s]);
    for (int w = 0; w < 6; w++) {
        const int64_t np = w < 2 ? hap_len[w] : (w < 4 ? ref_len : hap_len[w - 2]);
        if (np > 0) hipLaunchKernelGGL(k_prep_q16, blocks(np), dim3(256), 0, h->stream, D, w, np);
        if (np > 0) hipLaunchKernelGGL(k_prep_wk, blocks(np), dim3(256), 0, h->stream, D, w, np);
    }
    HIPCHK(h, hipEventRecord(e1, h->stream));

    lap("prep kernels");
    // ---- base descriptors: a function of the offsets (BaseDescs); here only the sums and the checks
    h->descs.set(b, h->var_off);
    h->ed_max_len = -1;
    int64_t sec_total = 0, jobs_total = 0;
    int64_t cells = 0, bytes_alg = 0;
    {
        for (int s = 0; s < 4; s++) sec_total += 2 * b->var_off[s][n];
        sec_total += 16 * int64_t(n);
        // ONE pass over the superclusters: the checks and sums of the batch and -- for a windowed round 0 over all alignments, the
        // usual case -- everything the host contributes to round 0's plan (plan0_device): the level of every alignment, the
        // parts' sums, the total workspace need (which is also what the arena has to hold), the long alignments.  (Three
        // passes over four million alignments until round 4, two of them through the full descriptors.)
        const int lv0_ = h->cfg.band_mode == 0 ? LV_DENSE : h->cfg.band_mode == 2 ? LV_C1 : h->cfg.band_mode == 3 ? LV_Q16 : LV_Z;
        const bool plan_pass = lv0_ <= LV_Q16 && !(h->cfg.flags & VPR_CFG_HAP_DEDUP) && n > 0;
        const int LLT = h->long_lt;
        h->p0 = vpr_handle::Plan0Pass();
        if (plan_pass) h->level.resize(size_t(n) * 4);
        struct Acc { int64_t jobs = 0, cells = 0, bytes = 0; int bad = -1; vpr_handle::Plan0Pass P; };
        std::vector<Acc> acc(PAR_MAX);
        par_for(size_t(n), [&](size_t b0, size_t e0, int tid) {
            Acc &A = acc[size_t(tid)];
            AlnDesc d{};
            for (size_t scu = b0; scu < e0; scu++) {
                const int sc = int(scu);
                const int64_t Lr = b->ref_off[sc + 1] - b->ref_off[sc];
                int64_t Lh[4];
                for (int s = 0; s < 4; s++) Lh[s] = b->hap_off[s][sc + 1] - b->hap_off[s][sc];
                int64_t nv = 0;
                for (int s = 0; s < 4; s++) nv += b->var_off[s][sc + 1] - b->var_off[s][sc];
                A.bytes += 6 * (Lh[0] + Lh[1] + Lh[2] + Lh[3]) + 11 * Lr + 26 * nv;
                if ((Lh[0] < 1 || Lh[1] < 1 || Lh[2] < 1 || Lh[3] < 1 || Lr < 1) && A.bad < 0) A.bad = sc;
                for (int i = 0; i < 4; i++) {
                    const int64_t Lq = Lh[i >> 1], Lt = Lh[2 + (i & 1)];
                    // (deferred sections: both segments longer than 32, or one longer than ED_INLINE_LONG -- segments are disjoint)
                    A.jobs += std::min(Lr, Lt) / 33 + Lr / (ED_INLINE_LONG + 1) + Lt / (ED_INLINE_LONG + 1);
                    A.cells += (Lq + Lr) * Lt;
                    if (!plan_pass) continue;
                    d.Lq = int32_t(Lq); d.Lr = int32_t(Lr); d.Lt = int32_t(Lt);
                    d.path_cap = d.Lq + d.Lr + d.Lt + 4;
                    const int dl = plan_level_of(lv0_, LLT, d.Lt);
                    const int W = LV_WINDOW[dl], part = d.Lt >= LLT ? 0 : 1;
                    A.P.part_cells[part] += int64_t(std::min(W, d.Lq) + std::min(W, d.Lr)) * d.Lt;
                    A.P.part_in[part] += 6 * int64_t(d.Lq) + 6 * int64_t(d.Lt) + 6 * int64_t(d.Lr);
                    A.P.part_dense[part] += int64_t(d.Lq + d.Lr) * d.Lt;
                    A.P.part_rows[part] += d.Lt;
                    h->level[size_t(sc) * 4 + size_t(i)] = uint8_t(dl);
                    if (d.Lt >= LLT)
                        A.P.big.emplace_back(-(int64_t(round_up(std::min(W, d.Lq), 16) + round_up(std::min(W, d.Lr), 16)) * d.Lt), int32_t(sc * 4 + i));
                    const int64_t need = window_layout(d, dl, 0, 0);
                    A.P.need128 += need / 128;
                    A.P.need_max = std::max(A.P.need_max, need);
                    if ((d.Lt < LLT && d.Lt > 0xfffe) || d.Lq < 1 || d.Lr < 1 || d.Lt < 1) A.P.bad = true;
                }
            }
        });
        int bad = -1;
        for (const Acc &A : acc) {
            jobs_total += A.jobs; cells += A.cells; bytes_alg += A.bytes;
            if (A.bad >= 0 && (bad < 0 || A.bad < bad)) bad = A.bad;
            for (int q = 0; q < 2; q++) {
                h->p0.part_cells[q] += A.P.part_cells[q]; h->p0.part_in[q] += A.P.part_in[q];
                h->p0.part_dense[q] += A.P.part_dense[q]; h->p0.part_rows[q] += A.P.part_rows[q];
            }
            h->p0.need128 += A.P.need128; h->p0.need_max = std::max(h->p0.need_max, A.P.need_max);
            h->p0.bad = h->p0.bad || A.P.bad;
            h->p0.big.insert(h->p0.big.end(), A.P.big.begin(), A.P.big.end());
        }
        h->p0.valid = plan_pass;
        if (bad >= 0) return fail(h, VPR_ERR_ARG, "supercluster %d has an empty string", bad);
    }
    bytes_alg += 2 * cells;
    h->timing.cells_dense = cells;
    h->timing.bytes_algorithmic = bytes_alg;

    lap("descriptors");
    const size_t na = h->descs.size();
    if ((rc = dev_alloc(h, &h->d_descs, na))) return rc;
    if ((rc = dev_alloc(h, &h->d_outs, na))) return rc;
    h->n_secs_cap = sec_total;
    if ((rc = dev_alloc(h, &h->d_secs, size_t(sec_total)))) return rc;
    for (int q = 0; q < 4; q++)
        if ((rc = dev_alloc(h, &h->d_fp[q], size_t(h->n_var[q >> 1])))) return rc;
    if ((rc = dev_alloc(h, &h->d_fp_table, 4))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_fp_table, h->d_fp, sizeof(h->d_fp), hipMemcpyHostToDevice, h->stream));
    h->jobs_cap = int32_t(std::min<int64_t>(jobs_total + 1, 0x7fffffff));
    if ((rc = dev_alloc(h, &h->d_jobs, size_t(h->jobs_cap)))) return rc;
    if ((rc = dev_alloc(h, &h->d_njobs, 1))) return rc;
    // device-side result columns
    DevResults &R = h->dR;
    memset(&R, 0, sizeof(R));
    for (int s = 0; s < 4; s++) {
        const float *vq = nullptr;
        if ((rc = dev_upload(h, &vq, b->var_qual[s], size_t(h->n_var[s])))) return rc;
        R.var_qual[s] = vq;
    }
    // every column vpr_download returns lives in ONE device region (256-byte aligned pieces), so that a caller whose
    // result block mirrors it (vpr_results_alloc) gets everything with a single copy
    {
        uint8_t *base = nullptr;
        size_t off = 0;
        auto place = [&](auto *&p, size_t cnt) {
            using T = std::remove_reference_t<decltype(*p)>;
            if (base) p = reinterpret_cast<T *>(base + off);
            off += (std::max<size_t>(cnt, 1) * sizeof(T) + 255) & ~size_t(255);
        };
        for (int pass = 0; pass < 2; pass++) {
            off = 0;
            for (int s = 0; s < 4; s++)
                for (int w = 0; w < 2; w++) {
                    const size_t nv = size_t(h->n_var[s]);
                    place(R.v[s][w].errtype, nv); place(R.v[s][w].sync_group, nv); place(R.v[s][w].credit, nv);
                    place(R.v[s][w].ref_ed, nv); place(R.v[s][w].query_ed, nv); place(R.v[s][w].callq, nv);
                }
            place(R.aln_dist, na); place(R.aln_end_plane, na); place(R.aln_beg_plane, na); place(R.aln_status, na);
            place(R.sc_phase, size_t(n)); place(R.orig_phase_dist, size_t(n)); place(R.swap_phase_dist, size_t(n));
            if (pass == 0) {
                if ((rc = dev_alloc(h, &base, off))) return rc;
                h->res_dev = base; h->res_bytes = off;
            }
        }
    }
    if ((rc = dev_alloc(h, &R.tally, 6))) return rc;
    R.max_qual = h->cfg.max_qual;
    R.credit_threshold = h->cfg.credit_threshold;
    R.phase_threshold = h->cfg.phase_threshold;
    h->tie_list_cap = int32_t(std::min<size_t>(std::max<size_t>(na, 4096), size_t(1) << 20));
    if ((rc = dev_alloc(h, &h->d_tie_list, size_t(h->tie_list_cap)))) return rc;
    if ((rc = dev_alloc(h, &h->d_tie_cnt, 8))) return rc;
    if ((rc = dev_alloc(h, &h->d_tie_ndec, TIE_DEC_SLOTS))) return rc;
    // fail lists: round 0 in [0, na + na/16 + 256) (a list that feeds a kernel directly is padded), the retry rounds of
    // the two ladders behind, the tie rounds' (which reject nothing) last
    const size_t n_fail = 4 * na + na / 16 + 1024, n_slots = 2 + 4 * LadderCtx::N_SLOTS;
    if ((rc = dev_alloc(h, &h->d_fail, n_fail))) return rc;
    if ((rc = dev_alloc(h, &h->d_cnt, n_slots))) return rc;
    {
        void *pf = nullptr, *pc = nullptr, *pl = nullptr, *pt = nullptr;
        { int rc_pin = pin_alloc(h, &pf, n_fail * sizeof(int32_t)); if (rc_pin) return rc_pin; }
        { int rc_pin = pin_alloc(h, &pc, n_slots * sizeof(int32_t)); if (rc_pin) return rc_pin; }
        { int rc_pin = pin_alloc(h, &pl, size_t(h->tie_list_cap) * sizeof(int4)); if (rc_pin) return rc_pin; }
        { int rc_pin = pin_alloc(h, &pt, 32 * sizeof(int32_t)); if (rc_pin) return rc_pin; }
        h->hp_fail = static_cast<int32_t *>(pf);
        h->hp_cnt = static_cast<int32_t *>(pc);
        h->hp_tie_list = static_cast<int4 *>(pl);
        h->hp_tie_cnt = static_cast<int32_t *>(pt);
        h->hp_flag = h->hp_tie_cnt + 8;     // [0..1] fail lists of round 0, [2..3] tie lists, [4..7] ladders idle, [8] speculative list
        memset(h->hp_cnt, 0, n_slots * sizeof(int32_t));
        memset(h->hp_tie_cnt, 0, 32 * sizeof(int32_t));
    }

    lap("result/aux allocations");
    // ---- arena for flag matrices, band origins and walks
    size_t free_b = 0, total_b = 0;
    HIPCHK(h, hipMemGetInfo(&free_b, &total_b));
    {   // (the driver's figure lags behind this process's own large frees: dev_books)
        const int64_t fb = books_free(int64_t(free_b), int64_t(total_b));
        if (h->debug && fb != int64_t(free_b)) fprintf(stderr, "[vpr] memory plan: the driver reports %.1f GB free, the books %.1f\n", double(free_b) / 1e9, double(fb) / 1e9);
        free_b = size_t(fb);
    }
    // what the batch may take: the free memory (the blocks kept from the previous batch count: the allocations below take
    // them first) less the tenth of the device the library leaves alone; on a device mostly taken by others, half of what is free
    int64_t avail = int64_t(free_b), avail_cache = 0;
    for (const auto &c : h->dev_cache) avail_cache += int64_t(c.bytes);
    avail += avail_cache;
    int64_t scratch_kept = 0;            // (the replay scratches have the handle's lifetime: part of the plan, and taken already)
    for (int k = 0; k < 4; k++) for (int e = 0; e < 2; e++) scratch_kept += h->lad[k].tie_scratch_bytes[e];
    avail += scratch_kept;
    h->mem_reserve = int64_t(double(total_b) * free_share());
    avail = std::max<int64_t>(avail - h->mem_reserve, avail / 2);
    // What the upload allocates BEHIND this plan comes off first (the ladders' and replays' halves were promised memory
    // the lane levels then took): the lane levels' interleaved position words (4 B per position of the short part) and logs
    // (20 B per truth row), padded to the waves' longest alignments (measured: < 15 %), the distance-1 level's blocks (0.8 of
    // both, zl_finish), the device planner's temporaries (28 B per alignment) and the saved forward flags of the long part.
    int64_t lane_est = 0;
    if (h->p0.valid && h->cfg.band_mode != 0 && h->cfg.band_mode != 2) {
        const double zl = 1.15 * (double(h->p0.part_in[1]) / 6.0 * 4.0 + 20.0 * double(h->p0.part_rows[1]));
        lane_est = int64_t(zl * (getenv("VPR_NO_D1") ? 1.0 : 1.8)) + 28 * int64_t(h->descs.size());
        avail = std::max<int64_t>(avail - lane_est, avail / 2);
    }
    int64_t budget = h->cfg.workspace_bytes > 0 ? h->cfg.workspace_bytes : int64_t(double(avail) * arena_share());
    if (budget < (8 << 20)) budget = 8 << 20;
    // do not allocate more than round 0 can use (its layout per alignment: make_plan)
    int64_t want = 0;
    {
        const int bm = h->cfg.band_mode;
        const bool q16ok = (bm == 1 || bm == 3);
        std::vector<int64_t> wants(PAR_MAX, 0);
        if (h->p0.valid) wants[0] = h->p0.need128 * 128 + 4096;       // (exactly what round 0's plan will ask for)
        else
        par_for(h->descs.size(), [&](size_t b0, size_t e0, int tid) {
          int64_t want = 0;
          for (size_t k = b0; k < e0; k++) {
            const AlnDesc &d = h->descs[k];
            int64_t flags;
            if (bm != 0 && q16ok && d.Lt < h->long_lt) {
                const int64_t nstr = (int64_t(d.Lt) + 3) / 4;
                flags = nstr * 128 + round_up(nstr * 8, 64);
            } else if (bm != 0) {
                const int W = (q16ok && bm != 2) ? LV_WINDOW[LONG_LV] : 64;
                flags = round_up(round_up(std::min(W, d.Lq), 16) * int64_t(d.Lt), 64) +
                        round_up(round_up(std::min(W, d.Lr), 16) * int64_t(d.Lt), 64) + round_up(8 * int64_t(d.Lt), 64);
            } else {
                flags = round_up(round_up(d.Lq, 32) * int64_t(d.Lt), 64) + round_up(round_up(d.Lr, 32) * int64_t(d.Lt), 64);
            }
            want += flags + round_up(int64_t(d.path_cap) * int64_t(sizeof(PathEnt)) + 128, 128) + 256;
          }
          wants[size_t(tid)] = want;
        });
        for (int64_t w : wants) want += w;
        if (h->cfg.workspace_bytes <= 0) budget = std::min(budget, std::max<int64_t>(want, 256 << 20));
    }
    h->arena_bytes = budget;
    if ((rc = dev_alloc(h, &h->d_arena, size_t(budget) + 256))) return rc;
    // What round 0 left is split between the ladders' workspaces and the replay scratches.  The ladders (two retry ladders
    // beside round 0, two for the tie rounds) start with a sixteenth of round 0's need each -- the short part's with at most
    // 2 GB -- and grow on demand inside their half (lad_grow); the replay scratches are allocated by the first tie round
    // (cfg.workspace_bytes bounds each workspace instead).
    {
        const int64_t rest = std::max<int64_t>(avail - budget, 0);
        h->lad_budget = rest / 2; h->tie_budget = rest - rest / 2;
        h->lad_bytes = 0; h->tie_bytes = scratch_kept;
        h->want0 = want;
        // (a handle that has released a batch knows what its ladders grew to: the same fraction of this batch's round-0 need,
        // all four scaled down together if that is more than the ladders' half)
        int64_t first[4] = {0, 0, 0, 0}, first_sum = 0;
        for (int k = 0; k < 4; k++) {
            int64_t b2 = std::min<int64_t>(h->lad_budget / 4, int64_t(ladder_share() * double(std::max<int64_t>(want / 16, int64_t(1) << 30))));
            if (k & 1) b2 = std::min<int64_t>(b2, int64_t(2) << 30);
            first[k] = std::max<int64_t>(b2, std::min<int64_t>(int64_t(h->lad_hw[k] * double(want)), h->lad_arena_max));
            first_sum += first[k];
        }
        const double shrink = first_sum > h->lad_budget && first_sum > 0 ? double(h->lad_budget) / double(first_sum) : 1.0;
        if (h->debug)
            fprintf(stderr, "[vpr] memory plan: free %.1f GB + kept blocks %.1f + scratches %.1f, reserve %.1f, lane levels %.1f -> %.1f to plan; round 0 wants %.1f, gets %.1f; "
                            "ladders' half %.1f (first workspaces %.1f %.1f %.1f %.1f), replays' half %.1f\n",
                    double(free_b) / 1e9, double(avail_cache) / 1e9, double(scratch_kept) / 1e9, double(h->mem_reserve) / 1e9, double(lane_est) / 1e9, double(avail) / 1e9,
                    double(want) / 1e9, double(budget) / 1e9, double(h->lad_budget) / 1e9, double(first[0]) * shrink / 1e9, double(first[1]) * shrink / 1e9,
                    double(first[2]) * shrink / 1e9, double(first[3]) * shrink / 1e9, double(h->tie_budget) / 1e9);
        for (int k = (h->cfg.band_mode != 0 ? 0 : 2); k < 4; k++) {   // (dense mode: only the tie rounds need one)
            int64_t b2 = h->cfg.workspace_bytes;
            if (b2 <= 0) b2 = int64_t(double(first[k]) * shrink);
            if (b2 < (8 << 20)) b2 = 8 << 20;
            h->lad[k].arena_bytes = b2;
            h->lad_bytes += b2;
            if ((rc = dev_alloc(h, &h->lad[k].arena, size_t(b2) + 256))) return rc;
        }
    }

    lap("arena allocation");
    // ---- round-0 plan over all alignments, cached (descriptors + work list live on the device)
    std::vector<int32_t> &all = h->scratch_i32[1];
    all.resize(na);
    h->alias.assign(size_t(b->n_sc), 0);
    h->n_aliased = 0;
    if ((rc = dev_alloc(h, &h->d_alias, size_t(std::max(b->n_sc, 1))))) return rc;
    HIPCHK(h, hipMemsetAsync(h->d_alias, 0, size_t(std::max(b->n_sc, 1)), h->stream));
    if (b->n_sc && (h->cfg.flags & VPR_CFG_HAP_DEDUP)) {
        hipLaunchKernelGGL(k_hap_alias, blocks(int64_t(b->n_sc)), dim3(256), 0, h->stream, D, h->d_alias);
        HIPCHK(h, hipMemcpyAsync(h->alias.data(), h->d_alias, size_t(b->n_sc), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, x_sync(h, h->stream, SITE));
        size_t np = 0;
        for (size_t a = 0; a < na; a++)
            if (alias_source(h->alias[a >> 2], int(a & 3)) < 0) all[np++] = int32_t(a);
        h->n_aliased = int64_t(na - np);
        all.resize(np);
    } else {
        par_for(na, [&](size_t b0, size_t e0, int) { for (size_t k = b0; k < e0; k++) all[k] = int32_t(k); });
    }
    const size_t np = all.size();      // alignments that are computed
    if (!h->p0.valid) h->level.assign(na, uint8_t(LV_DENSE));      // (else: set by the upload's pass over the superclusters)
    const int lv0 = h->cfg.band_mode == 0 ? LV_DENSE
                    : h->cfg.band_mode == 2 ? LV_C1
                    : h->cfg.band_mode == 3 ? LV_Q16 : LV_Z;
    bool dev_plan = false;
    if (np == na && lv0 <= LV_Q16 && !getenv("VPR_HOST_PLAN") && (rc = plan0_device(h, lv0, h->ev_offsets, &dev_plan))) return rc;
    if (!dev_plan && (rc = make_plan(h, all, lv0, h->plan0, h->d_arena, h->arena_bytes, 0, true))) return rc;
    lap(dev_plan ? "plan on the device" : "make_plan");
    // the long part's forward flags are kept a second time (tie rounds copy them back instead of repeating the sweep:
    // k_fwd_stripe_save / k_restore_stripe): a region as large as the part's share of the workspace -- the front of every
    // chunk -- unless that is more than 1 GB (a batch of tens of thousands of long alignments is a throughput problem, not a
    // latency chain)
    h->d_save = nullptr; h->save_bytes = 0;
    if (lv0 <= LV_Q16 && LONG_LV == LV_C1 && !h->no_flag_save && !h->plan0.off128.empty()) {
        int64_t need = 0;
        for (const Chunk &ch : h->plan0.chunks)
            if (ch.n_long > 0)
                need = std::max<int64_t>(need, ch.n_long < ch.count ? int64_t(h->plan0.off128[size_t(ch.work_off + ch.n_long)]) * 128 : h->plan0.arena_used);
        if (need > 0 && need <= (int64_t(1) << 30)) {
            // (optional: without the copy the tie rounds of the long part repeat the forward sweep, as with VPR_NO_FLAG_SAVE)
            h->soft_alloc = true;
            const int rc_save = dev_alloc(h, &h->d_save, size_t(need) + 256);
            h->soft_alloc = false;
            if (rc_save == VPR_OK) h->save_bytes = need;
            else { h->d_save = nullptr; h->err.clear(); (void)hipGetLastError(); }
        }
    }
    h->level0 = h->level;
    if (!dev_plan) {
        h->plan0_pos.assign(na, -1);
        par_for(np, [&](size_t b0, size_t e0, int) { for (size_t k = b0; k < e0; k++) h->plan0_pos[size_t(h->plan0.work[k])] = int32_t(k); });
        if ((rc = dev_alloc(h, &h->plan0.d_descs, np))) return rc;
        if ((rc = dev_alloc(h, &h->plan0.d_work, np))) return rc;
    }
    lap("plan0_pos");
    if (h->n_aliased) {
        BatchOffsets DO;
        for (int s = 0; s < 4; s++) { DO.hap_off[s] = D.hap_off[s]; DO.var_off[s] = D.var_off[s]; }
        DO.ref_off = D.ref_off;
        hipLaunchKernelGGL(k_alias_descs, blocks(int64_t(na)), dim3(256), 0, h->stream, DO, h->d_alias, b->n_sc, h->d_descs);
    }
    if (dev_plan) {
        // (descriptors, wave headers and position words are already enqueued: plan0_device)
    } else if (np && h->plan0.lazy) {
        // the plan crosses the link as (alignment, workspace offset) pairs; the device builds both descriptor tables
        void *pw = nullptr;
        { int rc_pin = pin_alloc(h, &pw, np * 8); if (rc_pin) return rc_pin; }
        int32_t *hw = static_cast<int32_t *>(pw);
        uint32_t *ho = reinterpret_cast<uint32_t *>(hw + np);
        const Plan &P0 = h->plan0;
        par_for(np, [&](size_t b0, size_t e0, int) {
            memcpy(hw + b0, P0.work.data() + b0, (e0 - b0) * 4);
            memcpy(ho + b0, P0.off128.data() + b0, (e0 - b0) * 4);
        });
        uint32_t *d_off = nullptr;
        if ((rc = dev_alloc(h, &d_off, np))) return rc;
        HIPCHK(h, hipMemcpyAsync(h->plan0.d_work, hw, np * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(d_off, ho, np * 4, hipMemcpyHostToDevice, h->stream));
        BatchOffsets DO;
        for (int s = 0; s < 4; s++) { DO.hap_off[s] = D.hap_off[s]; DO.var_off[s] = D.var_off[s]; }
        DO.ref_off = D.ref_off;
        hipLaunchKernelGGL(k_build_plan, blocks(int64_t(np)), dim3(256), 0, h->stream, DO, h->plan0.d_work, d_off, int(np), P0.lv,
                           P0.long_lt, P0.tag_or, h->plan0.d_descs, h->d_descs);
    } else if (np) {
        HIPCHK(h, hipMemcpyAsync(h->plan0.d_descs, h->plan0.descs.data(), np * sizeof(AlnDesc), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->plan0.d_work, h->plan0.work.data(), np * 4, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_scatter_descs, blocks(int64_t(np)), dim3(256), 0, h->stream, h->plan0.d_descs, int(np), h->d_descs);
    }
    if (lv0 == LV_Z && !dev_plan && (rc = prep_zero_lane(h))) return rc;

    HIPCHK(h, x_sync(h, h->stream, SITE));
    HIPCHK(h, hipGetLastError());
    lap("plan + upload");
    uint32_t err = 0;
    HIPCHK(h, hipMemcpy(&err, h->d_err, 4, hipMemcpyDeviceToHost));
    // (err: some supercluster has more than eight swap sources on one position -- DevBatch::sc_limit says which; its four
    // alignments come back with VPR_ST_ERR_LIMIT, everything else is evaluated)
    h->n_limit_sc = err ? 1 : 0;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    h->timing.ms_prep = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    h->uploaded = true;
    return VPR_OK;
}

int vpr_upload_variants(vpr_handle *h, const vpr_variants *v) {
    if (!h || !v) return VPR_ERR_ARG;
    // the host sizes and checks every region (the planner needs the lengths), the device writes the arrays: what crosses
    // the link is the variant tables, the region bounds and the contig
    vpr_owned_batch *ob = nullptr;
    const auto t0_ = std::chrono::steady_clock::now();
    int rc = vpr_batch_skeleton_from_variants(v, &ob);
    if (getenv("VPR_TIMING")) fprintf(stderr, "[vpr] upload (host) sizing pass %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count());
    if (rc) return fail(h, rc, "vpr_upload_variants: unsorted / overlapping variants or bad coordinates (%d)", rc);
    h->gen_src = v;
    rc = vpr_upload(h, vpr_owned_batch_view(ob));
    h->gen_src = nullptr;
    vpr_owned_batch_free(ob);
    if (getenv("VPR_TIMING")) fprintf(stderr, "[vpr] upload (host) vpr_upload_variants %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count());
    return rc;
}

/* test aid: the resident Level A arrays back to the host (dst's arrays sized by the offsets of the uploaded batch) */
int vpr_download_level_a(vpr_handle *h, vpr_batch *dst) {
    if (!h || !dst || !h->uploaded) return VPR_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const DevBatch &D = h->dB;
    const int n = h->n_sc;
    std::vector<int64_t> off(size_t(n) + 1);
    auto copy = [&](void *d, const void *s, size_t bytes) { return !bytes || !d || hipMemcpy(d, s, bytes, hipMemcpyDeviceToHost) == hipSuccess; };
    bool ok = true;
    int64_t ref_len = 0;
    ok = ok && copy(off.data(), D.ref_off, (size_t(n) + 1) * 8);
    ref_len = off[size_t(n)];
    ok = ok && copy(const_cast<int64_t *>(dst->ref_off), D.ref_off, (size_t(n) + 1) * 8);
    ok = ok && copy(const_cast<uint8_t *>(dst->ref_seq), D.ref_seq, size_t(ref_len));
    for (int q = 0; q < 2; q++) {
        ok = ok && copy(const_cast<int32_t *>(dst->ref_ptr[q]), D.ref_ptr[q], size_t(ref_len) * 4);
        ok = ok && copy(const_cast<uint8_t *>(dst->ref_flag[q]), D.ref_flag[q], size_t(ref_len));
    }
    for (int s = 0; s < 4; s++) {
        ok = ok && copy(off.data(), D.hap_off[s], (size_t(n) + 1) * 8);
        const int64_t hl = off[size_t(n)];
        ok = ok && copy(const_cast<int64_t *>(dst->hap_off[s]), D.hap_off[s], (size_t(n) + 1) * 8);
        ok = ok && copy(const_cast<uint8_t *>(dst->hap_seq[s]), D.hap_seq[s], size_t(hl));
        ok = ok && copy(const_cast<int32_t *>(dst->hap_ptr[s]), D.hap_ptr[s], size_t(hl) * 4);
        ok = ok && copy(const_cast<uint8_t *>(dst->hap_flag[s]), D.hap_flag[s], size_t(hl));
    }
    return ok ? VPR_OK : fail(h, VPR_ERR_DEVICE, "vpr_download_level_a: copy failed");
}

}  // extern "C"

namespace {

// One vpr_execute call: the state its phases share and the phases themselves.  Round 0 (round0_windowed / run_dense), the
// retry ladders (lad_start / lad_flush), the tie rounds (tie_round, tie_replay, tie_patch, spec_round), the final tie pass,
// the deferred edit distances (K4) and the finalisation (K5) all enqueue on the handle's streams; run() is the sequence.
const int64_t LONG_BULK = 8192;      // long alignments from which the long part is throughput work (Exec::Exec)
const int SPEC_MAX_DIST = 256;       // largest distance of an alignment whose tie replay is started speculatively (spec_round)

struct Exec {
    vpr_handle *h;
    hipStream_t st;
    std::chrono::steady_clock::time_point wall0, call0 = std::chrono::steady_clock::now();
    hipEvent_t t0 = nullptr, t1 = nullptr;      // bracket the whole call on the main stream
    int32_t flag_exp[16] = {};                  // value post_flag asked the device to write into hp_flag[idx]
    int64_t n_fwd = 0, cells_touched = 0, n_retry = 0;
    // ---- tie rounds (pr_tie.hip).  tie_full: FIFO logs and stamp grids sized for the worst case (second attempt, after a
    // capped job gave up)
    bool tie_full = false;
    size_t tie_job_cur = 0;
    int64_t n_tie_jobs = 0;
    int tie_dec_slot = 0;                          // decision lists handed out in this execute
    int64_t tie_dec_cur = 0;
    struct TieEarly { int32_t n_used; int32_t pos; const Plan *plan; int mode; };   // mode 1 / 2: TieJob::mode; 3: decided speculatively
    std::unordered_map<int32_t, TieEarly> tie_early;   // alignment -> where the bytes of its marking round are
    const Chunk *tie_resident = nullptr;               // the chunk of plan 0 a tie round's early entries lie in (tie_round)
    int64_t fwd_save_delta = 0;                        // != 0: the next 64-cell forward sweep also writes its flags that far behind (round 0's long part)
    std::unordered_map<int32_t, int32_t> tie_s;        // alignment -> its distance (from the tie lists): bounds the stamp grids
    int tie_patch_slot = -1; int64_t tie_patch_off = 0, tie_patch_cap = 0;   // decision list of the last early launch
    bool tie_patch_spec = false;                   // the part has alignments whose decisions are in the speculative list
    int spec_slot = -1; int64_t spec_off = 0, spec_cap = 0;
    std::unordered_map<int32_t, int> spec_set;     // alignments with a speculative replay in flight / done
    LadderCtx *tie_ctx;                            // the tie ladder whose round is being enqueued
    bool lad_tie_wait[2] = {false, false};         // a retry ladder's tie list is on its way
    std::vector<std::pair<TieJob *, size_t>> tie_job_blocks;   // (debug) the job blocks of this execute
    size_t tie_job_total = 0;
    int32_t n_jobs_pre = 0; bool n_jobs_final = false;        // deferred edit distances, as read with the final tie pass's count
    // ---- round 0 of a windowed plan
    const Plan &P0;
    const int64_t na_;
    hipStream_t s_long, s_short;
    LadderCtx &LL, &LS;                            // the retry ladders of the long / short part
    // regions of the tie list buffer: long part's marks, short part's marks, long part's speculative candidates (the retry
    // ladders' regions follow)
    int32_t tie_cap[3], tie_off[3];
    Plan spec_plan;

    explicit Exec(vpr_handle *h_)
        : h(h_), st(h_->stream), tie_ctx(&h_->lad[2]), P0(h_->plan0), na_(int64_t(h_->descs.size())), s_long(h_->cls_stream[0]),
          s_short(h_->cls_stream[1]), LL(h_->lad[0]), LS(h_->lad[1]) {
        tie_cap[0] = h->tie_list_cap / 8; tie_cap[1] = h->tie_list_cap / 4; tie_cap[2] = h->tie_list_cap / 8;
        tie_off[0] = 0; tie_off[1] = h->tie_list_cap / 4; tie_off[2] = h->tie_list_cap / 8;
        // The long part's stream has the high priority so that a HANDFUL of latency chains is dispatched ahead of the short part's
        // millions of waves.  A batch whose long part is itself bulk -- the stress workload: 35 000 alignments of 1 024+ rows, a
        // wavefront each, more than the device holds at once -- swaps the two streams' roles: its sweeps then run below the
        // ladders' and tie rounds' streams (the step's chain is long part -> its tie round's replays -> the retry round behind
        // them) instead of beside them: 284 against 307 ms per step (two alternating runs each on one box).  (Also tried there:
        // the short part's round 0 in front of the long part's launches -- its small kernels wait 35 - 45 ms each for a slot while
        // such a launch has workgroups left to dispatch --: 311 - 314 against 305 - 307, the short part is not on the chain.)
        int64_t n_long_all = 0;
        for (const Chunk &ch : P0.chunks) n_long_all += ch.n_long;
        if (n_long_all >= LONG_BULK) std::swap(s_long, s_short);
    }

    static dim3 blocks(int64_t n_) { return dim3(unsigned((n_ + 255) / 256)); }

    // HIP events bracket each launch on the stream the kernel is launched on
    bool need_err_check = false;    // a dense round holds an alignment only the strips can take: did the planner cut it?
    // own_events: the launch attaches the pair to ONE kernel itself (hipExtLaunchKernelGGL: start and stop of that kernel as the
    // device saw them); recorded around the launch, the first event completes when the stream reaches it and the interval then also
    // holds the time the kernel waits for the dispatcher behind other streams' work -- with two batches in flight 7 - 10 ms for a
    // 6 ms kernel.  Used for the lane kernel, whose duration the bench line's roofline block divides by.
    hipEvent_t own_a = nullptr, own_b = nullptr;
    template <class F>
    int timed(int kind, const vpr_launch_stat &ls, hipStream_t ks, const char *name, F &&launch, bool own_events = false) {
        EvPair ev; ev.kind = kind; ev.st = ls; ev.st.kind = kind;
        snprintf(ev.st.kernel, sizeof(ev.st.kernel), "%s", name);
        while (h->ev_pool.size() < h->ev_used + 2) {
            hipEvent_t e;
            HIPCHK(h, hipEventCreate(&e));
            h->ev_pool.push_back(e);
        }
        ev.a = h->ev_pool[h->ev_used++];
        ev.b = h->ev_pool[h->ev_used++];
        if (own_events) {
            own_a = ev.a; own_b = ev.b;
            launch();
            own_a = own_b = nullptr;
        } else {
            HIPCHK(h, hipEventRecord(ev.a, ks));
            launch();
            HIPCHK(h, hipEventRecord(ev.b, ks));
        }
        h->events.push_back(ev);
        return VPR_OK;
    }

    std::vector<std::pair<double, std::string>> trace_;      // (VPR_STALL_LOG) host actions with their wall-clock offsets
    double stall_dump_ms = getenv("VPR_STALL_DUMP_MS") ? atof(getenv("VPR_STALL_DUMP_MS")) : 200.0;     // executes slower than this are dumped
    void trace(const char *fmt, ...) {
        if (!h->stall_log) return;
        char buf[160];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        trace_.emplace_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - call0).count(), buf);
    }
    double phase_ms[6] = {0, 0, 0, 0, 0, 0};
    void phase(int k) { phase_ms[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - call0).count(); }
    void lapx(const char *what) {
        if (h->debug) fprintf(stderr, "[vpr] execute %-34s %8.3f ms\n", what,
                              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count());
    }

    void post_flag(int idx, hipStream_t ks) {     // "everything enqueued on ks so far is complete" -> hp_flag[idx]
        flag_exp[idx] = ++h->flag_seq;
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, ks, h->hp_flag + idx, flag_exp[idx]);
        const hipError_t e = hipGetLastError();          // a failed launch (of this or an earlier kernel of the phase) would
        if (e != hipSuccess && launch_err == hipSuccess) launch_err = e;   // leave the flag down for ever: see idle_check
    }
    hipError_t launch_err = hipSuccess;
    int64_t zl_wave0 = 0, zl_rows = 0;             // first wave header / truth rows of the chunk whose zero-distance part is being enqueued

    // the host loop found nothing to serve `idle_polls` times in a row: make sure the device is still alive.  A stream in
    // an error state (a failed launch, a fault in a kernel) never raises its flags; without this vpr_execute would hang.
    int idle_check(int64_t idle_polls, std::chrono::steady_clock::time_point since) {
        if (launch_err != hipSuccess) return fail(h, VPR_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(launch_err));
        if (idle_polls % 2048) return VPR_OK;            // about every 40 ms without progress
        hipStream_t ss[] = {h->stream, h->cls_stream[0], h->cls_stream[1], h->cls_stream[2], h->cls_stream[3], h->cls_stream[4],
                            h->cls_stream[5], h->cls_stream[6], h->cls_stream[7],
                            h->tie_stream[0], h->tie_stream[1], h->tie_stream[2], h->tie_stream[3]};
        bool busy = false;
        for (hipStream_t s_ : ss) {
            const hipError_t e = hipStreamQuery(s_);
            if (e != hipSuccess && e != hipErrorNotReady) return fail(h, VPR_ERR_DEVICE, "stream error while waiting for a round: %s", hipGetErrorString(e));
            busy = busy || e == hipErrorNotReady;
        }
        // every stream idle and still no flag: the work the flag stands behind is lost.  Streams busy: a round may take long
        // (one dense alignment of an SV-sized supercluster is a chain of tens of thousands of rows), so the bound is generous
        // -- VPR_IDLE_TIMEOUT_S in the environment, 1800 s -- and on giving up the device is drained first, so that the
        // caller can release or re-upload the handle without kernels still at work in its memory.
        const double idle_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - since).count();
        static const double limit_s = [] { const char *e = getenv("VPR_IDLE_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return v > 0 ? v : 1800.0; }();
        if ((!busy && idle_s > 10.0) || idle_s > limit_s) {
            (void)hipDeviceSynchronize();
            return fail(h, VPR_ERR_DEVICE, "no round completed for %.0f s (%s)", idle_s, busy ? "streams still busy" : "every stream idle: a flag was lost");
        }
        return VPR_OK;
    }

    bool flag_up(int idx) { return *static_cast<volatile int32_t *>(h->hp_flag + idx) == flag_exp[idx]; }

    // wave = true: one wavefront per alignment with LDS-staged window rows (long banded alignments)
    int walk_launch(const Plan &P, const int32_t *d_list, int32_t count, hipStream_t ks, bool wave, int my_w) {
        vpr_launch_stat ws_;
        memset(&ws_, 0, sizeof(ws_));
        ws_.threads = 64; ws_.n_units = count; ws_.cells_per_thread = wave ? 1 : 0;
        int32_t *a_i32 = reinterpret_cast<int32_t *>(P.arena);
        PathEnt *a_path = reinterpret_cast<PathEnt *>(P.arena);
        return timed(3, ws_, ks, wave ? "k_walk<wave>" : "k_walk<lane>", [&] {
            if (wave)
                hipLaunchKernelGGL(k_walk<true>, dim3(count), dim3(64), 0, ks, h->dB, h->d_descs, d_list, count,
                                   P.arena, a_i32, h->d_outs, a_path, h->d_secs, h->d_fp_table,
                                   h->d_jobs, h->d_njobs, h->jobs_cap, my_w, my_w);
            else
                hipLaunchKernelGGL(k_walk<false>, dim3((count + 63) / 64), dim3(64), 0, ks, h->dB, h->d_descs, d_list,
                                   count, P.arena, a_i32, h->d_outs, a_path, h->d_secs, h->d_fp_table,
                                   h->d_jobs, h->d_njobs, h->jobs_cap, my_w, my_w);
        });
    }

    int tie_replay(const Plan &P, int64_t off, int32_t cnt, hipStream_t ks_main, bool early) {
        if (early) { tie_patch_slot = -1; tie_patch_spec = false; }
        LadderCtx &tc = *tie_ctx;
        hipStream_t ks = early ? tc.ls2 : ks_main;
        const int se = early ? 1 : 0;      // the scratch of the stream the replays run on
        uint32_t *&scratch = tc.tie_scratch[se];
        int64_t &scratch_bytes = tc.tie_scratch_bytes[se];
        // scratch words of every job; the scratch grows to hold the whole launch (all replays concurrent: a long one is a
        // latency chain), bounded by half of the free memory -- beyond that the launch is cut into sub-batches
        struct Need { int32_t k; int64_t cells, cap, bcap, w_st, w_buf, w_bk; int32_t dlo[2], dn[2]; };
        std::vector<Need> needs{};
        needs.reserve(size_t(cnt));
        int64_t total = 0, largest = 0, dec_need = 0;
        for (int32_t k = 0; k < cnt; k++) {
            const AlnDesc d = plan_desc(h, P, size_t(off) + k);
            const auto it = tie_early.find(P.work[size_t(off) + k]);
            if ((it != tie_early.end()) != early) continue;
            if (early && it->second.mode == 3) { tie_patch_spec = true; continue; }
            Need N;
            N.k = k;
            // stamp words: both planes, diagonal-major; the diagonals a cell within the alignment's distance of the
            // "same reference base" track can be on (DevBatch::dspan), or all of them on a second attempt
            {
                const int2 A = h->hp_dspan[d.qs][d.sc], T = h->hp_dspan[d.ts][d.sc];
                const auto its = tie_s.find(P.work[size_t(off) + k]);
                // (+ the haps' own ranges: inside a repeat a hap matches shifted by an indel just as well, and a swap carries
                // that shift to the other plane)
                const int64_t M = (h->cfg.flags & VPR_CFG_TIE_SMALL_LOGS) ? 0
                                  : (its != tie_s.end() ? int64_t(its->second) + 4 + (int64_t(A.y) - A.x) + (int64_t(T.y) - T.x) : int64_t(1) << 30);
                const int64_t lo_all = -(int64_t(d.Lt) - 1);
                int64_t lo0 = lo_all, hi0 = d.Lq - 1, lo1 = lo_all, hi1 = d.Lr - 1;
                if (!tie_full) {
                    lo0 = std::max<int64_t>(lo0, int64_t(A.x) - T.y - M); hi0 = std::min<int64_t>(hi0, int64_t(A.y) - T.x + M);
                    lo1 = std::max<int64_t>(lo1, -int64_t(T.y) - M); hi1 = std::min<int64_t>(hi1, -int64_t(T.x) + M);
                    if (hi0 < lo0) hi0 = lo0;
                    if (hi1 < lo1) hi1 = lo1;
                }
                N.dlo[0] = int32_t(lo0); N.dn[0] = int32_t(hi0 - lo0 + 1);
                N.dlo[1] = int32_t(lo1); N.dn[1] = int32_t(hi1 - lo1 + 1);
            }
            N.cells = int64_t(N.dn[0] + N.dn[1]) * d.Lt + 4;
            if (N.cells >= (int64_t(1) << 32) - 2)
                return fail(h, VPR_ERR_ARG, "supercluster %d alignment %d: (Lq + Lr + 2 Lt) * Lt = %lld stamps exceed the tie replay's 32-bit cell index",
                            d.sc, d.aln, (long long)N.cells);
            // (first attempt: logs of 3 x the alignment's three lengths.  With 16 x, the logs and bucket words of a 16 k-row job
            // were 90 MB against 26 MB of stamps and cut the stress workload's 7 000 replays into 70 serial launches: 2.97 s per
            // step, 2.04 s with 3 x, no job overflowing; 1 x overflows and sends ~900 jobs to the full-log second attempt)
            N.cap = tie_full ? N.cells : std::min<int64_t>(N.cells, 3 * int64_t(d.Lq + d.Lr + d.Lt) + 4096);
            if (!tie_full && (h->cfg.flags & VPR_CFG_TIE_SMALL_LOGS)) N.cap = 32;
            N.cap = (std::max<int64_t>(2, std::min<int64_t>(N.cap, int64_t(1) << 26)) + 1) & ~int64_t(1);   // even: 8-byte entries follow
            int bi = 0;
            while (bi + 1 < TIE_N_BUCKETS && int64_t(TIE_BUCKETS_HOST[bi]) < N.cap) bi++;
            N.bcap = TIE_BUCKETS_HOST[bi];
            N.w_st = (N.cells + 3) & ~int64_t(3); N.w_buf = TIE_BUF_WORDS * N.cap; N.w_bk = (TIE_BKT_WORDS * N.bcap + 8 + 3) & ~int64_t(3);
            const int64_t need = N.w_st + N.w_buf + N.w_bk;
            total += need;
            largest = std::max(largest, need);
            if (early) dec_need += it->second.mode == 2 ? 256 : it->second.n_used;
            needs.push_back(N);
        }
        if (needs.empty()) return VPR_OK;
        trace("    replay: %zu jobs sized", needs.size());
        // largest first: a launch lasts as long as its largest job, so when the scratch forces sub-batches the big jobs
        // should share theirs with each other, not spread over all of them
        std::stable_sort(needs.begin(), needs.end(), [](const Need &x, const Need &y) { return x.cells > y.cells; });
        if (tie_job_cur + needs.size() > h->tie_jobs_cap) return fail(h, VPR_ERR_STATE, "tie round: job buffer overflow");
        // The scratch grows to hold the whole launch, at least doubling, up to a bound fixed at vpr_create (a sixteenth of the
        // device's memory; never from hipMemGetInfo: with other handles at work on the device that figure depends on the
        // moment).  The outgrown block is PARKED, not freed (kernels of this stream may still use it, and hipFree both waits
        // for the whole device and takes seconds for a block of gigabytes): free_batch moves it to the kept blocks.  The
        // scratches themselves have the handle's lifetime.
        // Inside the replays' half of the batch's memory plan: a scratch takes at most two thirds of what is left of it (the
        // launches of the other contexts want theirs), the outgrown block counts until the batch is released; what the
        // largest job needs is granted whatever the plan says.
        const bool must = largest * 4 + 256 > scratch_bytes;
        const int64_t room = std::max<int64_t>(h->tie_budget - h->tie_bytes, 0) / 3 * 2;
        const int64_t nb_plan = std::min<int64_t>(std::max<int64_t>(std::min<int64_t>(total * 4, h->tie_scratch_max), 2 * scratch_bytes), room);
        if (must || (total * 4 + 256 > scratch_bytes && scratch_bytes < h->tie_scratch_max && nb_plan > scratch_bytes + scratch_bytes / 2)) {
            const int64_t nb = std::max<int64_t>(nb_plan, largest * 4) + 256;
            void *q = nullptr;
            int64_t got_b = nb;
            h->soft_alloc = !must;
            hipError_t e_sc = x_malloc(h, &q, size_t(nb), SITE);
            h->soft_alloc = false;
            if (e_sc != hipSuccess && must && largest * 4 + 256 < nb) {     // the whole launch does not fit: what its largest job needs does
                (void)hipGetLastError();
                got_b = largest * 4 + 256;
                e_sc = x_malloc(h, &q, size_t(got_b), SITE);
            }
            if (e_sc != hipSuccess) {
                (void)hipGetLastError();
                if (must) return fail(h, VPR_ERR_NOMEM, "tie replay scratch (%lld bytes)", (long long)got_b);
                // (the launch still fits in sub-batches: keep the block)
            } else {
                if (scratch) h->parked.push_back(vpr_handle::Blk{scratch, size_t(scratch_bytes)});
                scratch = static_cast<uint32_t *>(q);
                scratch_bytes = got_b;
                h->tie_bytes += got_b;
                tc.tie_clean[se] = 0;
            }
        }
        // decision list of an early launch: one region of the decision buffer, its length in a counter of its own
        int4 *dec = nullptr;
        int32_t *n_dec = h->d_tie_ndec;
        int64_t dec_cap = 0;
        if (early) {
            dec_cap = dec_need + 64;
            if (tie_dec_slot >= TIE_DEC_SLOTS) return fail(h, VPR_ERR_STATE, "tie round: out of decision lists");
            if (tie_dec_cur + dec_cap > h->tie_dec_cap) {      // (a region handed out earlier may still be in use: keep the old block)
                const int64_t nc = std::max<int64_t>(2 * (tie_dec_cur + dec_cap), 1 << 16);
                int rc_ = dev_alloc(h, &h->d_tie_dec, size_t(nc));
                if (rc_) return rc_;
                h->tie_dec_cap = nc;
                tie_dec_cur = 0;
            }
            dec = h->d_tie_dec + tie_dec_cur;
            n_dec = h->d_tie_ndec + tie_dec_slot;
            tie_patch_slot = tie_dec_slot++; tie_patch_off = tie_dec_cur; tie_patch_cap = dec_cap;
            tie_dec_cur += dec_cap;
        }
        size_t k0 = 0;
        while (k0 < needs.size()) {
            // sub-batch [k0, k1) that fits the scratch
            int64_t words = 0;
            size_t k1 = k0;
            TieJob *jobs = h->hp_tie_jobs + tie_job_cur;
            while (k1 < needs.size()) {
                const Need &N = needs[k1];
                const int64_t need = N.w_st + N.w_buf + N.w_bk;
                if (k1 > k0 && (words + need) * 4 > scratch_bytes) break;
                TieJob &J = jobs[k1 - k0];
                memset(&J, 0, sizeof(J));
                J.a = P.work[size_t(off) + N.k];
                J.cap = int32_t(N.cap); J.bcap = int32_t(std::min<int64_t>(N.bcap, 0x7fffffff));
                J.stamp_off = words;
                J.dlo[0] = N.dlo[0]; J.dlo[1] = N.dlo[1]; J.dn[0] = N.dn[0]; J.dn[1] = N.dn[1];
                J.buf_off = words + N.w_st;
                J.bkt_off = (words + N.w_st + N.w_buf) / 2;   // (all three terms are multiples of four words)
                if (early) {
                    const TieEarly &E = tie_early[J.a];
                    const AlnDesc o = plan_desc(h, *E.plan, size_t(E.pos));
                    J.mode = E.mode; J.n_used = E.n_used;
                    J.old_band_w = o.band_w; J.old_pitch[0] = o.pitch[0]; J.old_pitch[1] = o.pitch[1];
                    J.old_mat_off[0] = o.mat_off[0]; J.old_mat_off[1] = o.mat_off[1]; J.old_blo_off = o.blo_off;
                    J.old_arena = E.plan->arena;
                }
                words += need;
                k1++;
            }
            const int32_t nj = int32_t(k1 - k0);
            tie_job_cur += size_t(nj);
            n_tie_jobs += nj;
            if (!tc.tie_first_seen[se]) { tc.tie_first_seen[se] = true; tc.tie_first[se] = words * 4; }
            if (words * 4 > tc.tie_clean[se]) HIPCHK(h, hipMemsetAsync(scratch, 0xff, size_t(words) * 4, ks));
            tc.tie_clean[se] = 0;
            vpr_launch_stat ts_;
            memset(&ts_, 0, sizeof(ts_));
            // eight waves per job when the launch's jobs are large (a megaword of stamps and logs on average: the alignments of
            // 10 000+ rows of the stress workload, whose passes over a wave's cells are then half as long -- its step 312 ->
            // 301 ms); four for the thousands of small jobs of a whole-genome batch, where eight cost the step 0.6 ms
            // (VPR_TIE_WIDE_WORDS moves the border: diagnostic)
            static const int64_t wide_words = [] { const char *e = exp_getenv("VPR_TIE_WIDE_WORDS"); return e ? atoll(e) : (int64_t(1) << 20); }();
            const bool wide_job = words / std::max<int32_t>(nj, 1) >= wide_words;
            ts_.threads = wide_job ? 512 : 256; ts_.n_units = nj; ts_.cells_per_thread = early ? 1 : 0;
            int rc = timed(6, ts_, ks, early ? "k_tie_replay<early>" : "k_tie_replay", [&] {
                if (wide_job)
                    hipLaunchKernelGGL(k_tie_replay<8>, dim3(nj), dim3(512), 0, ks, h->dB, h->d_descs, jobs, nj, P.arena,
                                       reinterpret_cast<const int32_t *>(P.arena), h->d_outs, scratch, h->d_tie_cnt + 1,
                                       dec, n_dec, int(dec_cap));
                else
                    hipLaunchKernelGGL(k_tie_replay<4>, dim3(nj), dim3(256), 0, ks, h->dB, h->d_descs, jobs, nj, P.arena,
                                       reinterpret_cast<const int32_t *>(P.arena), h->d_outs, scratch, h->d_tie_cnt + 1,
                                       dec, n_dec, int(dec_cap));
            });
            if (rc) return rc;
            k0 = k1;
        }
        if (early) { HIPCHK(h, hipEventRecord(tc.ev2, ks)); (void)hipStreamQuery(ks); }
        trace("    replay launched");
        return VPR_OK;
    }

    // apply the decisions of the part's early replays to the flags its repeated forward sweep has just written
    int tie_patch(const Plan &P, hipStream_t ks, int tag) {
        if (tie_patch_slot >= 0) {
            HIPCHK(h, hipStreamWaitEvent(ks, tie_ctx->ev2, 0));     // the early replays ran on the side stream
            hipLaunchKernelGGL(k_tie_patch, blocks(tie_patch_cap), dim3(256), 0, ks, h->d_descs, h->d_tie_dec + tie_patch_off,
                               h->d_tie_ndec + tie_patch_slot, int(tie_patch_cap), P.arena, tag);
            tie_patch_slot = -1;
        }
        if (tie_patch_spec && spec_slot >= 0) {
            HIPCHK(h, hipStreamWaitEvent(ks, h->ev_spec, 0));
            hipLaunchKernelGGL(k_tie_patch, blocks(spec_cap), dim3(256), 0, ks, h->d_descs, h->d_tie_dec + spec_off,
                               h->d_tie_ndec + spec_slot, int(spec_cap), P.arena, tag);
            tie_patch_spec = false;
        }
        return VPR_OK;
    }

    // ---- dense plan: per chunk, each kernel class runs K1 -> K2 -> K3 on its own stream (forked from and
    // joined into `base`); one_stream: everything on `base` (retry rounds that run beside other work)
    // tag_or: TIE_TAG_BIT when the plan belongs to a tie round (forward sweep, container-order replay, then the rest)
    // Strip tables of the wide alignments [off, off + cnt) of a dense plan (pr_strip.hip), planned on the device on stream ks.
    struct StripGroup {
        int64_t off = 0; int32_t cnt = 0, grid = 0;
        int32_t *d_base = nullptr, *d_n = nullptr, *d_ok = nullptr, *d_prog = nullptr;     // d_prog: [2][grid] forward / backward
        int32_t *d_order = nullptr;      // the forward sweep's launch order of the strips
        int64_t *d_boff = nullptr;
        StripTab *d_tab = nullptr;
        int2 *d_bnd = nullptr; int4 *d_bbnd = nullptr;
    };
    int strip_plan(const Plan &P, const int32_t *d_work, int64_t off, int32_t cnt, hipStream_t ks, StripGroup &G) {
        G.off = off; G.cnt = cnt;
        // slots per alignment: twice what the longer plane needs (a cut may have to move far back to a clean column)
        auto slots_of = [](const AlnDesc &d) { return 2 * ((std::max(d.Lq, d.Lr) + ST_CAP - 1) / ST_CAP) + 2; };
        int64_t slots_all = 0;
        for (int32_t k = 0; k < cnt; k++) slots_all += slots_of(P.descs[size_t(off) + k]);
        void *pb = nullptr;
        { int rc_pin = exec_pin(h, &pb, size_t(cnt + 1) * 4 + size_t(cnt) * 9 + size_t(slots_all) * 4 + 16); if (rc_pin) return rc_pin; }
        int64_t *h_boff = static_cast<int64_t *>(pb);
        int32_t *h_base = reinterpret_cast<int32_t *>(h_boff + cnt);
        int32_t *h_order = h_base + cnt + 1;
        uint8_t *h_fits = reinterpret_cast<uint8_t *>(h_order + slots_all);
        int64_t slots = 0, rows = 0;
        for (int32_t k = 0; k < cnt; k++) {
            const AlnDesc &d = P.descs[size_t(off) + k];
            const int ns = slots_of(d);
            h_base[k] = int32_t(slots);
            h_boff[k] = rows;
            h_fits[k] = dense_fits(d) ? 1 : 0;
            if (!h_fits[k]) need_err_check = true;
            slots += ns;
            rows += int64_t(ns) * d.Lt;
        }
        h_base[cnt] = int32_t(slots);
        G.grid = int32_t(slots);
        // The forward sweep's workgroups draw their strip from this list in launch order: strip 0 of every alignment first
        // (largest matrix first), then every strip 1, ...  A strip that runs beside its left neighbour holds a compute unit
        // while it waits for that neighbour's blocks (with one workgroup per compute unit and more strips than compute units,
        // 45 % of the launch's workgroup time was such waiting); a strip that starts when a compute unit falls free finds its
        // neighbour's column published and runs through.  (i, j - 1) still precedes (i, j): the wait cannot deadlock.
        {
            std::vector<int32_t> by_size(size_t(cnt), 0);
            for (int32_t k = 0; k < cnt; k++) by_size[size_t(k)] = k;
            auto cells = [&](int32_t k) { const AlnDesc &d = P.descs[size_t(off) + k]; return int64_t(d.Lq + d.Lr) * d.Lt; };
            std::stable_sort(by_size.begin(), by_size.end(), [&](int32_t x, int32_t y) { return cells(x) > cells(y); });
            int64_t w = 0;
            for (int j = 0; w < slots; j++)
                for (int32_t k : by_size)
                    if (j < h_base[k + 1] - h_base[k]) h_order[w++] = h_base[k] + j;
        }
        void *q = nullptr;
        int rc;
        const size_t b_small = size_t(cnt + 1) * 4 + size_t(cnt) * 17 + size_t(slots) * (sizeof(StripTab) + 4) + 4096;
        if ((rc = exec_alloc(h, &q, b_small))) return rc;
        uint8_t *u = static_cast<uint8_t *>(q);
        G.d_boff = reinterpret_cast<int64_t *>(u); u += size_t(cnt) * 8;
        G.d_tab = reinterpret_cast<StripTab *>(u); u += size_t(slots) * sizeof(StripTab);
        G.d_base = reinterpret_cast<int32_t *>(u); u += size_t(cnt + 1) * 4;
        G.d_n = reinterpret_cast<int32_t *>(u); u += size_t(cnt) * 4;
        G.d_ok = reinterpret_cast<int32_t *>(u); u += size_t(cnt) * 4;
        G.d_order = reinterpret_cast<int32_t *>(u); u += size_t(slots) * 4;
        uint8_t *d_fits = u;
        if ((rc = exec_alloc(h, &q, size_t(slots) * 8 + 8))) return rc;       // (+ the forward sweep's ticket counter)
        G.d_prog = static_cast<int32_t *>(q);
        if ((rc = exec_alloc(h, &q, size_t(rows) * sizeof(int4)))) return rc;      // (the forward columns first, then reused by the backward sweep)
        G.d_bnd = static_cast<int2 *>(q);
        G.d_bbnd = static_cast<int4 *>(q);
        HIPCHK(h, hipMemcpyAsync(G.d_boff, h_boff, size_t(cnt) * 8, hipMemcpyHostToDevice, ks));
        HIPCHK(h, hipMemcpyAsync(G.d_base, h_base, size_t(cnt + 1) * 4, hipMemcpyHostToDevice, ks));
        HIPCHK(h, hipMemcpyAsync(d_fits, h_fits, size_t(cnt), hipMemcpyHostToDevice, ks));
        HIPCHK(h, hipMemcpyAsync(G.d_order, h_order, size_t(slots) * 4, hipMemcpyHostToDevice, ks));
        HIPCHK(h, hipMemsetAsync(G.d_prog, 0, size_t(slots) * 8 + 8, ks));
        hipLaunchKernelGGL(k_strip_plan, dim3((cnt + 63) / 64), dim3(64), 0, ks, h->dB, h->d_descs, d_work + off, cnt, G.d_base, G.d_tab,
                           G.d_n, G.d_ok, d_fits, h->d_err, h->d_outs);
        return VPR_OK;
    }

    // from_window: the plan's alignments failed the exit test of a window level (their AlnOut::s is that level's distance)
    int run_dense(const Plan &P, const int32_t *d_work, hipStream_t base, bool one_stream, int tag_or = 0, bool from_window = false) {
        for (const Chunk &ch : P.chunks) {
            if (!one_stream) HIPCHK(h, hipEventRecord(h->ev_fork, base));
            StripGroup G;
            bool strips = false;
            int64_t g_off = -1; int32_t g_cnt = 0;
            if (!h->no_strips)
                for (const Launch &L : ch.launches)
                    if (L.cls >= STRIP_CLS) { if (g_off < 0) g_off = L.work_off; g_cnt += L.count; }
            strips = g_cnt > 0;
            // per launch of a class: LDS of its largest member the one-workgroup kernels can hold, int16 rows or not
            struct ClsLds { size_t f = 0, b = 0; bool s16 = false, any = false; };
            auto cls_lds = [&](const Launch &L) {
                ClsLds R;
                // the backward launch uses int16 score rows when a member's int32 rows do not fit LDS (make_plan
                // has checked that the int16 rows do and that the scores are in range); vpr_config.flags bit 0 forces it.
                // Members the one-workgroup kernels cannot hold at all belong to the strips (the skip list has them).
                R.s16 = (h->cfg.flags & VPR_CFG_DENSE_S16) != 0;
                for (int32_t w = 0; w < L.count; w++) {
                    const AlnDesc &d = P.descs[L.work_off + w];
                    if (!dense_fits(d)) continue;
                    R.any = true;
                    if (bwd_lds_bytes(L.cls, d.Lq, d.Lr) > LDS_MAX) R.s16 = true;
                }
                for (int32_t w = 0; w < L.count; w++) {
                    const AlnDesc &d = P.descs[L.work_off + w];
                    if (dense_fits(d) && R.s16 && !s16_ok(d)) R.s16 = false;    // (only reachable when forced: keep the int32 rows)
                }
                for (int32_t w = 0; w < L.count; w++) {
                    const AlnDesc &d = P.descs[L.work_off + w];
                    if (!dense_fits(d)) continue;
                    R.f = std::max(R.f, fwd_lds_bytes(L.cls, d.Lq, d.Lr));
                    R.b = std::max(R.b, bwd_lds_bytes(L.cls, d.Lq, d.Lr, R.s16));
                }
                return R;
            };
            auto fwd_name = [&](const KernelClass &K) { return std::string("k_fwd<") + std::to_string(K.nt) + "," + std::to_string(K.c) + ">"; };
            auto bwd_name = [&](const KernelClass &K, bool s16) {
                return std::string("k_bwd<") + std::to_string(K.nt) + "," + std::to_string(K.c) + (s16 ? ",s16>" : ">");
            };
            // ---- classes of up to 2048 columns: one workgroup per alignment
            for (const Launch &L : ch.launches) {
                if (strips && L.cls >= STRIP_CLS) continue;
                const KernelClass &K = CLASSES[L.cls];
                // (streams 2 and 3 are the retry ladders' own: a dense class there would hold up the ladder's next round)
                hipStream_t ks = one_stream ? base : h->cls_stream[(L.cls == 2 || L.cls == 3) ? 7 : L.cls];
                if (!one_stream) HIPCHK(h, hipStreamWaitEvent(ks, h->ev_fork, 0));
                const ClsLds Z = cls_lds(L);
                vpr_launch_stat ls;
                memset(&ls, 0, sizeof(ls));
                ls.threads = K.nt; ls.cells_per_thread = K.c; ls.n_units = L.count;
                int64_t in_bytes = 0;
                for (int32_t w = 0; w < L.count; w++) {
                    const AlnDesc &d = P.descs[L.work_off + w];
                    ls.cells += int64_t(d.Lq + d.Lr) * d.Lt;
                    in_bytes += 6 * int64_t(d.Lq) + 6 * int64_t(d.Lt) + 6 * int64_t(d.Lr);
                }
                cells_touched += ls.cells;
                ls.cells_dense = ls.cells;
                ls.bytes_algorithmic = ls.cells + in_bytes;
                int rc = VPR_OK;
                if (tag_or && (rc = tie_replay(P, L.work_off, L.count, ks, true))) return rc;
                rc = timed(1, ls, ks, fwd_name(K).c_str(), [&] {
                    hipLaunchKernelGGL(fwd_kernel(L.cls), dim3(L.count), dim3(K.nt), Z.f, ks, h->dB, h->d_descs,
                                       d_work + L.work_off, P.arena, h->d_outs, static_cast<const int32_t *>(nullptr));
                    hipLaunchKernelGGL(k_fwd_finish, dim3((L.count + 255) / 256), dim3(256), 0, ks,
                                       d_work + L.work_off, L.count, h->d_outs, tag_or);
                });
                if (rc) return rc;
                n_fwd++;
                if (tag_or && ((rc = tie_replay(P, L.work_off, L.count, ks, false)) || (rc = tie_patch(P, ks, tag_or)))) return rc;
                ls.bytes_algorithmic = ls.cells;
                rc = timed(2, ls, ks, bwd_name(K, Z.s16).c_str(), [&] {
                    hipLaunchKernelGGL(bwd_kernel(L.cls, Z.s16), dim3(L.count), dim3(K.nt), Z.b, ks, h->dB, h->d_descs,
                                       d_work + L.work_off, P.arena, h->d_outs, static_cast<const int32_t *>(nullptr));
                });
                if (rc) return rc;
                if ((rc = walk_launch(P, d_work + L.work_off, L.count, ks, L.count < 2048, tag_or))) return rc;
                if (!one_stream) {
                    HIPCHK(h, hipEventRecord(h->ev_join[L.cls], ks));
                    HIPCHK(h, hipStreamWaitEvent(base, h->ev_join[L.cls], 0));
                }
            }
            // ---- the wide alignments of the chunk (its last launches: the classes are in ascending order), spread over
            // several workgroups each; what the strip planner cannot cut is left to the one-workgroup kernels (skip lists).
            // A tie round treats the group as one launch: early replays, forward sweeps, late replays + patch, backward sweeps.
            if (strips) {
                hipStream_t ks = one_stream ? base : h->cls_stream[STRIP_CLS];
                if (!one_stream) HIPCHK(h, hipStreamWaitEvent(ks, h->ev_fork, 0));
                int rc = strip_plan(P, d_work, g_off, g_cnt, ks, G);
                if (rc) return rc;
                if (tag_or && (rc = tie_replay(P, G.off, G.cnt, ks, true))) return rc;
                vpr_launch_stat ls;
                memset(&ls, 0, sizeof(ls));
                ls.threads = ST_NT; ls.cells_per_thread = ST_C; ls.n_units = G.cnt;
                for (int32_t w = 0; w < G.cnt; w++) {
                    const AlnDesc &d = P.descs[size_t(G.off) + w];
                    ls.cells += int64_t(d.Lq + d.Lr) * d.Lt;
                }
                ls.cells_dense = ls.cells; ls.bytes_algorithmic = ls.cells;
                cells_touched += ls.cells;
                rc = timed(1, ls, ks, "k_fwd_strip", [&] {
                    hipLaunchKernelGGL(k_fwd_strip, dim3(G.grid), dim3(ST_NT), 0, ks, h->dB, h->d_descs, d_work + g_off, g_cnt, G.d_base,
                                       G.d_tab, G.d_n, G.d_order, G.d_boff, G.d_bnd, G.d_prog, P.arena, h->d_outs, (from_window && !getenv("VPR_NO_UB")) ? 1 : 0);
                });
                if (rc) return rc;
                n_fwd++;
                vpr_launch_stat l2;
                for (const Launch &L : ch.launches) {      // forward: what the planner could not cut; s / end plane of all
                    if (L.cls < STRIP_CLS) continue;
                    const KernelClass &K = CLASSES[L.cls];
                    const ClsLds Z = cls_lds(L);
                    memset(&l2, 0, sizeof(l2));
                    l2.threads = K.nt; l2.cells_per_thread = K.c; l2.n_units = L.count;
                    rc = timed(1, l2, ks, fwd_name(K).c_str(), [&] {
                        if (Z.any)
                            hipLaunchKernelGGL(fwd_kernel(L.cls), dim3(L.count), dim3(K.nt), Z.f, ks, h->dB, h->d_descs,
                                               d_work + L.work_off, P.arena, h->d_outs, G.d_ok + (L.work_off - G.off));
                        hipLaunchKernelGGL(k_fwd_finish, dim3((L.count + 255) / 256), dim3(256), 0, ks,
                                           d_work + L.work_off, L.count, h->d_outs, tag_or);
                    });
                    if (rc) return rc;
                }
                if (tag_or && ((rc = tie_replay(P, G.off, G.cnt, ks, false)) || (rc = tie_patch(P, ks, tag_or)))) return rc;
                rc = timed(2, ls, ks, "k_bwd_strip", [&] {
                    hipLaunchKernelGGL(k_bwd_strip, dim3(G.cnt), dim3(ST_NT), 0, ks, h->dB, h->d_descs, d_work + G.off, G.cnt, G.d_base,
                                       G.d_tab, G.d_n, G.d_boff, G.d_bbnd, G.d_prog + G.grid, P.arena, h->d_outs);
                });
                if (rc) return rc;
                for (const Launch &L : ch.launches) {      // backward: what the planner could not cut
                    if (L.cls < STRIP_CLS) continue;
                    const KernelClass &K = CLASSES[L.cls];
                    const ClsLds Z = cls_lds(L);
                    if (!Z.any) continue;
                    memset(&l2, 0, sizeof(l2));
                    l2.threads = K.nt; l2.cells_per_thread = K.c; l2.n_units = L.count;
                    rc = timed(2, l2, ks, bwd_name(K, Z.s16).c_str(), [&] {
                        hipLaunchKernelGGL(bwd_kernel(L.cls, Z.s16), dim3(L.count), dim3(K.nt), Z.b, ks, h->dB, h->d_descs,
                                           d_work + L.work_off, P.arena, h->d_outs, G.d_ok + (L.work_off - G.off));
                    });
                    if (rc) return rc;
                }
                if ((rc = walk_launch(P, d_work + G.off, G.cnt, ks, G.cnt < 2048, tag_or))) return rc;
                if (!one_stream) {
                    HIPCHK(h, hipEventRecord(h->ev_join[STRIP_CLS], ks));
                    HIPCHK(h, hipStreamWaitEvent(base, h->ev_join[STRIP_CLS], 0));
                }
            }
        }
        return VPR_OK;
    }

    // ---- one launch sequence of a windowed plan: `cnt` alignments of the plan's work list from `off`, all at
    // level lv, on stream ks:  K1 -> accept test -> rejected ids appended to fail slot `slot` (event
    // ev_slot[slot] marks the list complete) -> K2 -> K3.  K2/K3 skip rejected alignments, so the host can
    // start their retry round while this sequence is still running.
    int enqueue_part(const Plan &P, const int32_t *d_work, int64_t off, int32_t cnt, int lv, hipStream_t ks,
                            int slot, int64_t fail_off, bool long_part, int64_t part_cells, int64_t part_in,
                            int64_t part_dense, int dtag_override = -1, int phases = 7,
                            const int32_t *n_dev = nullptr, int32_t n_all = 0, int tag_or = 0) {
        // phases: 1 = forward sweep + accept test + fail list, 2 = backward sweep, 4 = walk + credit.
        // n_dev: device-side length of a device-built work list; cnt is then the cap of entries processed and
        // n_all the most entries the list can hold (the fail list scans all of them).
        // tag_or: TIE_TAG_BIT for a tie round (its descriptors and its accept test carry the level tag with that bit)
        const int W = LV_WINDOW[lv], C = W / 64, tag = LV_TAG[lv] | tag_or;
        const bool q16 = lv <= LV_Q16, zero = lv == LV_Z;
        const int dtag = dtag_override >= 0 ? dtag_override : tag;   // tag carried by the descriptors of the list
        const int32_t *list = d_work + off;
        int32_t *a_i32 = reinterpret_cast<int32_t *>(P.arena);
        PathEnt *a_path = reinterpret_cast<PathEnt *>(P.arena);
        vpr_launch_stat ls;
        memset(&ls, 0, sizeof(ls));
        ls.threads = q16 ? 16 : 64; ls.cells_per_thread = q16 ? 1 : C; ls.n_units = cnt;
        ls.cells = part_cells;
        ls.cells_dense = part_dense;
        ls.bytes_algorithmic = ls.cells + part_in;
        int rc = VPR_OK;
        if (phases & 1) {
        if (zero) {     // the lane kernel moves bytes per truth ROW, not per window cell: position words 16 B, cell records 8 B written
            // + 8 B read, path_ptr words 4 B (pr_zl.hip; the walk -- k_zero_walk since round 5 -- has the other 20 of the former 56:
            // two position words, the path_ptr word, the 8-byte step)
            ls.bytes_algorithmic = 36 * zl_rows;
            ls.cells = 8 * zl_rows;                               // cell slots per row
        }
        cells_touched += ls.cells;
        // round 0's long part: the flags a second time; a tie round whose alignments all have such a copy: the copy instead of the sweep
        const bool save = lv == LV_C1 && !tag_or && fwd_save_delta != 0;
        RestoreJob *rjobs = nullptr;
        bool restore = false;
        if (lv == LV_C1 && tag_or && h->d_save && tie_resident && !n_dev) {
            void *pb = nullptr;
            { int rc_pin = exec_pin(h, &pb, size_t(cnt) * sizeof(RestoreJob)); if (rc_pin) return rc_pin; }
            rjobs = static_cast<RestoreJob *>(pb);
            restore = true;
            for (int32_t k = 0; k < cnt && restore; k++) {
                const auto it = tie_early.find(P.work[size_t(off) + size_t(k)]);
                if (it == tie_early.end() || it->second.plan != &h->plan0 || (it->second.mode != 1 && it->second.mode != 3) ||
                    it->second.pos < tie_resident->work_off || it->second.pos >= tie_resident->work_off + tie_resident->n_long) { restore = false; break; }
                const AlnDesc od = plan_desc(h, h->plan0, size_t(it->second.pos)), nd = plan_desc(h, P, size_t(off) + size_t(k));
                if (od.band_pad != LV_TAG[LV_C1] || od.pitch[0] != nd.pitch[0] || od.pitch[1] != nd.pitch[1] ||
                    od.mat_off[1] + int64_t(od.pitch[1]) * od.Lt > h->save_bytes) { restore = false; break; }
                rjobs[k] = RestoreJob{{od.mat_off[0], od.mat_off[1]}, od.blo_off};
            }
        }
        rc = timed(1, ls, ks, restore ? "k_restore_stripe" : band_fwd_name(lv), [&] {
            if (zero)        // forward + backward + walk of the zero-distance alignments, one lane each (pr_zl.hip)
                // (tried: this launch -- the first of its batch, 62 000 long-lived waves that take the wave slots the other batch's
                // next kernels wait for -- on a stream of its own below the short part's priority: the waiting stream's queue then
                // keeps the lower one from being served, 30 ms per step instead of 18.6, a lone step 38 instead of 23.6)
                hipExtLaunchKernelGGL(k_zero_lane, dim3((cnt + 63) / 64), dim3(64), h->zl_lds_bytes, ks, own_a, own_b, 0, h->d_descs, list, cnt,
                                      h->d_zl_hdr + zl_wave0, h->d_zl_in, h->d_zl_log, h->d_outs, a_path,
                                      (h->cfg.flags & VPR_CFG_KEEP_PATHS) ? 1 : 0, h->d_d1_hdr ? h->d1_max_rows : 0, h->lane_prio_rows);
            else if (q16)
                hipLaunchKernelGGL(k_fwd_q16, dim3((cnt + 3) / 4), dim3(64), 0, ks, h->dB, h->d_descs, list, cnt,
                                   P.arena, a_i32, h->d_outs, n_dev);
            else if (restore)
                hipLaunchKernelGGL(k_restore_stripe, dim3(cnt), dim3(256), 0, ks, h->d_descs, list, rjobs, h->plan0.arena,
                                   int64_t(reinterpret_cast<intptr_t>(h->d_save) - reinterpret_cast<intptr_t>(h->plan0.arena)), P.arena, a_i32);
            else if (save)
                hipLaunchKernelGGL(k_fwd_stripe_save, dim3(cnt), dim3(64), 0, ks, h->dB, h->d_descs, list, P.arena, a_i32, h->d_outs,
                                   fwd_save_delta);
            else
                hipLaunchKernelGGL(band_fwd_kernel(lv), dim3(cnt), dim3(lv >= LV_C4 ? W : 64), 0,
                                   ks, h->dB, h->d_descs, list, P.arena, a_i32, h->d_outs);
            hipLaunchKernelGGL(k_fwd_band_finish, dim3((cnt + 255) / 256), dim3(256), 0, ks, list, cnt, h->d_outs, tag, n_dev);
        }, zero);
        if (rc) return rc;
        n_fwd++;
        {
            const int32_t nc = n_dev ? n_all : cnt;
            if (zero && h->d_d1_blk) ordered_fails(list, nc, h->d_fail + fail_off, h->d_cnt + slot, n_dev, ks);
            else
            hipLaunchKernelGGL(k_collect_fails, dim3((nc + 255) / 256), dim3(256), 0, ks, list, nc, h->d_outs,
                               h->d_fail + fail_off, h->d_cnt + slot, zero ? 1 : 0, n_dev);
            // (the zero-distance level's list is consumed on the device, in place: the host only wants its length)
            hipLaunchKernelGGL(k_publish_fails, dim3(zero ? 1 : std::min((nc + 255) / 256, 1024)), dim3(256), 0, ks,
                               h->d_fail + fail_off, h->d_cnt + slot, h->hp_fail + fail_off, h->hp_cnt + slot, zero ? 0 : 1);
        }
        HIPCHK(h, hipEventRecord(h->ev_slot[slot], ks));
        }
        if ((phases & 2) && !zero) {
        ls.bytes_algorithmic = ls.cells;
        rc = timed(2, ls, ks, band_bwd_name(lv), [&] {
            if (q16)
                hipLaunchKernelGGL(k_bwd_q16, dim3((cnt + 3) / 4), dim3(64), 0, ks, h->dB, h->d_descs, list, cnt,
                                   P.arena, a_i32, h->d_outs, tag, dtag, n_dev);
            else
                hipLaunchKernelGGL(band_bwd_kernel(lv), dim3(cnt), dim3(lv >= LV_C4 ? W : 64), 0,
                                   ks, h->dB, h->d_descs, list, P.arena, a_i32, h->d_outs, tag);
        });
        if (rc) return rc;
        }
        if (!(phases & 4)) return rc;
        // long part (or a small launch of long retries): wave-per-alignment walk; windows up to 256
        const bool wave_walk = !q16 && (long_part || cnt < 2048);      // (k_walk<wave> stages tiles around the walk for the wider windows)
        // The segment walk follows all 128 possible entry cells of every segment: right for launches that are latency chains (a
        // few hundred long alignments, a retry round of some thousands), 128 times the work for a launch that fills the device
        // by itself (the stress workload's 80 000 alignments of thousands of rows: 216 ms against the lane walk's 40)
        int64_t walk_rows_sum = 0;
        if (lv == LV_C1 && !h->seq_walk)
            for (int32_t k = 0; k < cnt && walk_rows_sum <= WSEG_MAX_ROWS; k++) walk_rows_sum += plan_desc(h, P, size_t(off) + size_t(k)).Lt;
        const bool seg_walk = lv == LV_C1 && !h->seq_walk && walk_rows_sum <= WSEG_MAX_ROWS;
        const bool row_walk = lv == LV_C1 && (wave_walk || seg_walk);
        vpr_launch_stat ws_;
        memset(&ws_, 0, sizeof(ws_));
        ws_.threads = q16 ? 16 : 64; ws_.n_units = cnt; ws_.cells_per_thread = 2;
        if (q16) {
            // 16-cell layout: row-sweep walk, four alignments per wave (phase A) + credit walk (phase B)
            if (!zero) rc = timed(3, ws_, ks, "k_walk_q16", [&] {
                hipLaunchKernelGGL(k_walk_q16, dim3((cnt + 3) / 4), dim3(64), 0, ks, h->dB, h->d_descs, list, cnt,
                                   P.arena, a_i32, h->d_outs, a_path, tag, dtag, n_dev);
            });
            if (rc) return rc;
            ws_.cells_per_thread = 3;
            if (zero) {      // the walk of what the zero level finished (its third pass until round 5: pr_zl.hip), then the credit walk
                ws_.bytes_algorithmic = 20 * zl_rows;
                rc = timed(3, ws_, ks, "k_zero_walk", [&] {
                    hipLaunchKernelGGL(k_zero_walk, dim3((cnt + 63) / 64), dim3(64), 0, ks, h->d_descs, list, cnt, h->d_zl_hdr + zl_wave0,
                                       h->d_zl_in, h->d_zl_log, h->d_outs, a_path, (h->cfg.flags & VPR_CFG_KEEP_PATHS) ? 1 : 0, tag,
                                       h->lane_prio_rows);
                });
                if (rc) return rc;
                ws_.bytes_algorithmic = 0;
            }
            if (zero) rc = timed(3, ws_, ks, "k_zero_credit", [&] {
                hipLaunchKernelGGL(k_zero_credit, dim3((cnt + 63) / 64), dim3(64), 0, ks, h->dB, h->d_descs, list, cnt,
                                   h->d_zl_hdr + zl_wave0, h->d_zl_log, h->d_outs, h->d_secs, h->d_fp_table, h->d_jobs,
                                   h->d_njobs, h->jobs_cap, tag);
            });
            // (a launch of a few thousand alignments -- a tie round, a late retry round -- lasts as long as its longest member:
            // a wavefront per alignment reads the path in coalesced chunks, 0.45 us per row against 1.3 for a lane)
            else if (cnt <= CREDIT_WAVE_MAX && !n_dev) rc = timed(3, ws_, ks, "k_credit<wave>", [&] {
                hipLaunchKernelGGL(k_credit<true>, dim3(cnt), dim3(64), 0, ks, h->dB, h->d_descs, list, cnt,
                                   h->d_outs, a_path, h->d_secs, h->d_fp_table, h->d_jobs, h->d_njobs,
                                   h->jobs_cap, dtag, tag, n_dev);
            });
            else {
                // a device-built list is sorted longest first (ordered_fails): its head -- the alignments of up to 1 023 rows, whose
                // lane walks ARE the launch's duration -- gets a wavefront each (0.45 us per row against 1.3), the rest a lane
                static const int head_max = [] { const char *e = exp_getenv("VPR_CREDIT_HEAD"); return e ? atoi(e) : 4096; }();
                // (sorted only when ordered_fails built it, i.e. with the distance-1 level's blocks in place: k_collect_fails lists
                // the rejects in arrival order, and a head of arbitrary short alignments would only cost wave slots -- ADVICE r5)
                const int32_t head = (n_dev && h->d_d1_blk && cnt > 2 * head_max) ? head_max : 0;
                if (head > 0) {
                    vpr_launch_stat wh_ = ws_;
                    wh_.n_units = head;
                    rc = timed(3, wh_, ks, "k_credit<wave>", [&] {
                        hipLaunchKernelGGL(k_credit<true>, dim3(head), dim3(64), 0, ks, h->dB, h->d_descs, list, head,
                                           h->d_outs, a_path, h->d_secs, h->d_fp_table, h->d_jobs, h->d_njobs,
                                           h->jobs_cap, dtag, tag, n_dev, 0);
                    });
                    if (rc) return rc;
                }
                rc = timed(3, ws_, ks, "k_credit<lane>", [&] {
                    hipLaunchKernelGGL(k_credit<false>, dim3((cnt - head + 63) / 64), dim3(64), 0, ks, h->dB, h->d_descs,
                                       list, cnt, h->d_outs, a_path, h->d_secs, h->d_fp_table, h->d_jobs,
                                       h->d_njobs, h->jobs_cap, dtag, tag, n_dev, int(head));
                });
            }
        } else if (row_walk) {
            // striped 64-cell layout: the walk (phase A) in parallel over segments of 128 truth rows (pr_walkseg.hip; the
            // sequential row sweep, k_walk_rows, with VPR_SEQ_WALK in the environment), then the credit walk (phase B)
            // Round 0's long part (sorted longest first): only the alignments of VPR_WSEG_MIN_ROWS (2 048) rows or more are latency
            // chains worth 128 times the work; the shorter ones -- nine in ten since the long part starts at 1 024 rows -- take the row
            // sweep, a wavefront each (chains of at most 2 047 x 0.6 us), in front of the segment walk of the long ones on the same
            // stream: 1.7 instead of 6.8 ms of 10 000-workgroup launches per step with two batches in flight, step -0.4 ms, alone -0.9.
            static const int wseg_min_rows = [] { const char *e = exp_getenv("VPR_WSEG_MIN_ROWS"); return e ? atoi(e) : 2048; }();      // (0: the segment walk for all)
            int32_t n_seg = cnt;
            if (seg_walk && long_part && wseg_min_rows > 0) {
                n_seg = 0;
                while (n_seg < cnt && plan_desc(h, P, size_t(off) + size_t(n_seg)).Lt >= wseg_min_rows) n_seg++;
            }
            if (!seg_walk || n_seg < cnt) {
                const int32_t k0 = seg_walk ? n_seg : 0;
                vpr_launch_stat wr_ = ws_;
                wr_.n_units = cnt - k0;
                rc = timed(3, wr_, ks, "k_walk_rows", [&] {
                    hipLaunchKernelGGL(k_walk_rows, dim3(cnt - k0), dim3(64), 0, ks, h->dB, h->d_descs, list + k0, cnt - k0,
                                       P.arena, a_i32, h->d_outs, a_path, tag);
                });
                if (rc) return rc;
            }
            if (seg_walk && n_seg > 0) {
                const int32_t cnt_all = cnt;
                const int32_t cnt = n_seg;          // (the segment walk's launches below: the head of the list)
                (void)cnt_all;
                int64_t rows_sum = 0;
                for (int32_t k = 0; k < cnt; k++) rows_sum += plan_desc(h, P, size_t(off) + size_t(k)).Lt;
                ws_.n_units = cnt;
                WsegTables T;
                T.cap = int32_t(std::min<int64_t>(rows_sum / WSEG_ROWS + cnt + 1, 0x7fffffff));
                void *q = nullptr;
                const size_t b_cnt = 256, b_base = round_up(int64_t(cnt) * 4, 256), b_own = round_up(int64_t(T.cap) * 8, 256),
                             b_map = size_t(T.cap) * 128 * 8, b_ent = size_t(T.cap) * 16;
                if ((rc = exec_alloc(h, &q, b_cnt + b_base + b_own + b_map + b_ent))) return rc;
                uint8_t *u = static_cast<uint8_t *>(q);
                T.counter = reinterpret_cast<int32_t *>(u); u += b_cnt;
                T.seg_base = reinterpret_cast<int32_t *>(u); u += b_base;
                T.owner = reinterpret_cast<int2 *>(u); u += b_own;
                T.map = reinterpret_cast<uint2 *>(u); u += b_map;
                T.entry = reinterpret_cast<int4 *>(u);
                HIPCHK(h, hipMemsetAsync(T.counter, 0, 4, ks));
                rc = timed(3, ws_, ks, "k_walk_seg", [&] {
                    hipLaunchKernelGGL(k_wseg_plan, blocks(cnt), dim3(256), 0, ks, h->d_descs, list, cnt, h->d_outs, tag, T);
                    hipLaunchKernelGGL(k_wseg_map, dim3(T.cap), dim3(128), 0, ks, h->dB, h->d_descs, list, P.arena, a_i32, T);
                    hipLaunchKernelGGL(k_wseg_compose, dim3(cnt), dim3(64), 0, ks, h->d_descs, list, cnt, a_i32, h->d_outs, T);
                    hipLaunchKernelGGL(k_wseg_emit, dim3(T.cap), dim3(64), 0, ks, h->dB, h->d_descs, list, P.arena, a_i32, h->d_outs,
                                       a_path, T);
                });
            }
            if (rc) return rc;
            ws_.cells_per_thread = 3;
            ws_.n_units = cnt;
            const bool wave_credit = wave_walk || cnt <= CREDIT_WAVE_MAX;
            rc = timed(3, ws_, ks, wave_credit ? "k_credit<wave>" : "k_credit<lane>", [&] {
                if (wave_credit)
                    hipLaunchKernelGGL(k_credit<true>, dim3(cnt), dim3(64), 0, ks, h->dB, h->d_descs, list, cnt,
                                       h->d_outs, a_path, h->d_secs, h->d_fp_table, h->d_jobs, h->d_njobs,
                                       h->jobs_cap, dtag, tag, n_dev);
                else
                    hipLaunchKernelGGL(k_credit<false>, dim3((cnt + 63) / 64), dim3(64), 0, ks, h->dB, h->d_descs,
                                       list, cnt, h->d_outs, a_path, h->d_secs, h->d_fp_table, h->d_jobs,
                                       h->d_njobs, h->jobs_cap, dtag, tag, n_dev);
            });
        } else {
            rc = walk_launch(P, list, cnt, ks, wave_walk, tag);
        }
        return rc;
    }

    // the rejected entries of a work list in the list's own order, padded per workgroup to groups of four (k_fails_*)
    void ordered_fails(const int32_t *list, int32_t nc, int32_t *out, int32_t *cnt, const int32_t *n_dev, hipStream_t ks) {
        const int nb = (nc + 255) / 256;
        hipLaunchKernelGGL(k_fails_count, dim3(nb), dim3(256), 0, ks, list, nc, h->d_outs, h->d_d1_blk, 1, n_dev);
        hipLaunchKernelGGL(k_fails_scan, dim3(1), dim3(FAILS_SCAN_NT), 0, ks, h->d_d1_blk, nb, cnt);
        hipLaunchKernelGGL(k_fails_scatter, dim3(nb), dim3(256), 0, ks, list, nc, h->d_outs, h->d_d1_blk, out, 1, n_dev);
    }

    // ---- the distance-1 lane level between the zero-distance sweep and the in-place 16-cell round (pr_d1.hip): `list` is the
    // zero level's device-built fail list (length *n_dev, at most n_all entries, the first `cap` of them are worked on).
    // phase 1: headers, position words, the four passes, accept test, and the list of what is still rejected (d_d1_fail,
    // length d_d1_info[2]); phase 4: the credit sections of what was finished here.
    int enqueue_d1(const int32_t *list, const int32_t *n_dev, int32_t cap, int32_t n_all, hipStream_t ks, int ph, int ztag) {
        const int nw = (cap + 63) / 64;
        vpr_launch_stat ls;
        memset(&ls, 0, sizeof(ls));
        ls.threads = 64; ls.cells_per_thread = 1; ls.n_units = cap;
        int rc = VPR_OK;
        if (ph == 1) {
            rc = timed(0, ls, ks, "k_prep_d1", [&] {
                hipLaunchKernelGGL(k_d1_hdr, dim3(nw), dim3(64), 0, ks, h->d_descs, list, n_dev, cap, h->d_outs, h->d_d1_hdr);
                hipLaunchKernelGGL(k_d1_scan, dim3(1), dim3(D1_SCAN_NT), 0, ks, h->d_d1_hdr, nw, h->d1_in_cap, h->d1_log_cap, h->d_d1_info);
                hipLaunchKernelGGL(k_prep_d1, dim3(nw), dim3(256), 0, ks, h->dB, h->d_descs, list, n_dev, cap, h->d_outs, h->d_d1_hdr, h->d_d1_in);
            });
            if (rc) return rc;
            rc = timed(1, ls, ks, "k_one_lane", [&] {
                hipLaunchKernelGGL(k_one_lane, dim3(nw), dim3(64), 0, ks, h->d_descs, list, n_dev, cap, h->d_d1_hdr, h->d_d1_in, h->d_d1_log,
                                   h->d_outs, reinterpret_cast<PathEnt *>(h->plan0.arena), (h->cfg.flags & VPR_CFG_KEEP_PATHS) ? 1 : 0, h->d_d1_info,
                                   std::max(1, h->lane_prio_rows / 4));
                hipLaunchKernelGGL(k_fwd_band_finish, dim3((cap + 255) / 256), dim3(256), 0, ks, list, cap, h->d_outs, D1_TAG, n_dev);
            });
            if (rc) return rc;
            ordered_fails(list, n_all, h->d_d1_fail, h->d_d1_info + 2, n_dev, ks);
        } else if (ph == 4) {
            ls.cells_per_thread = 3;
            rc = timed(3, ls, ks, "k_one_walk", [&] {
                hipLaunchKernelGGL(k_one_walk, dim3(nw), dim3(64), 0, ks, h->d_descs, list, n_dev, cap, h->d_d1_hdr, h->d_d1_in, h->d_d1_log,
                                   h->d_outs, reinterpret_cast<PathEnt *>(h->plan0.arena), (h->cfg.flags & VPR_CFG_KEEP_PATHS) ? 1 : 0, h->d_d1_info,
                                   std::max(1, h->lane_prio_rows / 4));
            });
            if (rc) return rc;
            rc = timed(3, ls, ks, "k_one_credit", [&] {
                hipLaunchKernelGGL(k_one_credit, dim3(nw), dim3(64), 0, ks, h->dB, h->d_descs, list, n_dev, cap, h->d_d1_hdr, h->d_d1_log,
                                   h->d_outs, h->d_secs, h->d_fp_table, h->d_jobs, h->d_njobs, h->jobs_cap, ztag);
            });
        }
        return rc;
    }

    // read a fail slot once its list is complete (copies ride on stream `ls`, never the null stream)
    int read_fails(int slot, int64_t fail_off, hipStream_t ls, std::vector<int32_t> &fails) {
        HIPCHK(h, x_event_sync(h, h->ev_slot[slot], SITE));      // the list and its length are in pinned host memory by then
        const int32_t nf = h->hp_cnt[slot];
        if (nf > 0) {
            const size_t f0 = fails.size();
            fails.insert(fails.end(), h->hp_fail + fail_off, h->hp_fail + fail_off + nf);
            if (h->debug && nf <= (getenv("VPR_DEBUG_REJECTS") ? atoi(getenv("VPR_DEBUG_REJECTS")) : 64)) {
                for (size_t k = f0; k < fails.size(); k++) {
                    const int32_t a = fails[k];
                    AlnOut o;
                    (void)hipMemcpyAsync(&o, h->d_outs + a, sizeof(o), hipMemcpyDeviceToHost, ls);
                    (void)x_sync(h, ls, SITE);
                    const AlnDesc &d = h->descs[a];
                    fprintf(stderr, "[vpr] level %d rejected sc %d aln %d: Lq %d Lr %d Lt %d  s %d exit_min %d dq %d dr %d\n",
                            int(h->level[a]), d.sc, d.aln, d.Lq, d.Lr, d.Lt, o.s, o.exit_min, o.dist_q, o.dist_r);
                }
            }
        }
        return VPR_OK;
    }

    // ---- retry ladders.  Every rejected alignment climbs one window level (16 -> 64 -> 256 -> 1024 -> dense)
    // until its exit test passes.  Two independent ladders (one fed by the long part of round 0, one by the
    // short part) each own a stream, a workspace, staging buffers and fail slots, so their rounds run beside
    // each other and beside the rest of round 0; the host only ever waits for a fail list it needs next.
    bool overlap(const LadderCtx &c) const { return !h->no_round_overlap && !h->no_strips && c.ls2 != nullptr && (&c - h->lad) < 2; }
    int lad_sync(LadderCtx &c) {        // everything the context has in flight (its side stream: the back halves of retry rounds)
        HIPCHK(h, x_sync(h, c.ls, SITE));
        if (overlap(c)) HIPCHK(h, x_sync(h, c.ls2, SITE));
        return VPR_OK;
    }
    int lad_flush(LadderCtx &c, std::vector<int32_t> &out) {
        for (const auto &pd : c.pending) {
            int rc = read_fails(pd.first, pd.second, c.ls, out);
            if (rc) return rc;
        }
        c.pending.clear();
        c.plans.clear();
        c.slot_cur = 0; c.fail_cur = 0;
        // (a retry ladder whose rounds overlap keeps handing out fresh workspace and staging: what the last rounds' backward
        // sweeps and walks still use stays untouched; lad_start waits for them when it runs out)
        if (!overlap(c)) { c.stage_cur = 0; c.arena_cur = 0; }
        return VPR_OK;
    }

    // (regions of the tie list buffer: [0, 1/2) round 0 -- tie_off / tie_cap --, [1/2, 3/4) and [3/4, 1) the two retry ladders.
    //  Until the end of round 2 ladder 1's region ran to 5/4 of the buffer: more than cap / 8 marked alignments from one
    //  retry round wrote behind it; found by the chain fuzzer once the pooled allocator had put other arrays there)
    int lad_tie_off(int k) { return h->tie_list_cap / 2 + k * (h->tie_list_cap / 4); }

    int lad_tie_cap(int k) { (void)k; return h->tie_list_cap / 4; }

    void ladder_collect(int k, const int32_t *list, int32_t n, hipStream_t ks) {
        hipLaunchKernelGGL(k_collect_ties_list, blocks(n), dim3(256), 0, ks, list, n, h->d_outs, h->d_tie_list + lad_tie_off(k),
                           h->d_tie_cnt + 5 + k, lad_tie_cap(k));
        hipLaunchKernelGGL(k_publish_ties, dim3(4), dim3(256), 0, ks, h->d_tie_list + lad_tie_off(k), h->d_tie_cnt + 5 + k,
                           h->hp_tie_list + lad_tie_off(k), h->hp_tie_cnt + 5 + k, lad_tie_cap(k));
        post_flag(9 + k, ks);
        lad_tie_wait[k] = true;
    }

    // a ladder that plans from the front of its workspace again overwrites the walks of its earlier rounds:
    // vpr_download_path then reports "not resident" instead of reading what replaced them
    void drop_resident(const LadderCtx &c) {
        for (auto &r : h->resident)
            if (r.second >= c.arena && r.second < c.arena + c.arena_bytes) r.second = nullptr;
    }

    // A larger workspace for ladder context c -- and for its sibling retry ladder: which of the two contexts a round runs
    // on depends on which is idle at that moment, so they grow together and one execute settles the sizes of both.  The
    // old blocks stay allocated until the batch is released (dev_alloc: the plans in flight point into them), so nothing
    // has to be waited for.
    // must: what one alignment of the round needs -- granted outside the ladders' half of the memory plan if it has to be
    bool lad_grow(LadderCtx &c, int64_t nb, int64_t must = 0) {
        const int k = int(&c - h->lad);
        for (int j = 0; j < 4; j++) {
            LadderCtx &g = h->lad[j];
            const bool sibling = &g != &c;
            if (sibling && !(k < 2 && j == 1 - k)) continue;
            if (g.arena_bytes >= nb) continue;
            // (inside the ladders' half of the batch's memory plan: the outgrown block stays allocated, so the new one counts whole)
            const int64_t need = sibling ? 0 : must;
            int64_t nb_g = std::max<int64_t>(std::min<int64_t>(nb, h->lad_budget - h->lad_bytes), need);
            if (nb_g < g.arena_bytes + g.arena_bytes / 2 && need <= g.arena_bytes) {
                if (!sibling) return false;
                continue;
            }
            uint8_t *na2 = nullptr;
            h->soft_alloc = true;
            int rc_grow = dev_alloc(h, &na2, size_t(nb_g) + 256);
            h->soft_alloc = false;
            if (rc_grow != VPR_OK && need > g.arena_bytes && need < nb_g) {      // what it must have, then
                h->err.clear();
                (void)hipGetLastError();
                nb_g = need;
                rc_grow = dev_alloc(h, &na2, size_t(nb_g) + 256);
            }
            if (rc_grow != VPR_OK) {
                h->err.clear();
                (void)hipGetLastError();
                if (!sibling) return false;
                continue;
            }
            h->lad_bytes += nb_g;
            g.arena = na2; g.arena_bytes = nb_g; g.arena_cur = 0;
        }
        return true;
    }

    // tie: the tie pass -- same level again, with the container-order replay between the forward and the backward sweep
    int lad_start(LadderCtx &c, std::vector<int32_t> &fails, std::vector<int32_t> &carry, bool tie = false) {
        if (fails.empty()) return VPR_OK;
        if (!tie) n_retry += int64_t(fails.size());
        std::sort(fails.begin(), fails.end());   // deterministic planning
        std::vector<int32_t> by_lv[LV_DENSE + 1];
        // (a tie round repeats the alignment's level; the zero-distance lane kernel never marks one, so an alignment still
        // listed at LV_Z was accepted by the in-place 16-cell round)
        // (Round 6 tried to skip the window levels an alignment's lengths rule out -- a truth hap longer or shorter than both planes
        // by more than a window holds the path's INS / DEL run -- and measured it on one box: sv_synth 390 - 396 ms with the skip
        // against 328 - 337 without, joint_synth 60 against 50: the narrow levels fail within microseconds, and what reaches the
        // dense level early runs there as a round of its own in front of the ladder's real dense round.  Taken out.)
        for (int32_t a : fails) by_lv[std::min<int>(tie ? std::max<int>(h->level[size_t(a)], LV_Q16) : h->level[size_t(a)] + 1, LV_DENSE)].push_back(a);
        if (h->debug)
            fprintf(stderr, "[vpr] retry round (ladder %d): %zu -> 16, %zu -> 64, %zu -> 256, %zu -> 1024, %zu -> dense\n",
                    int(&c - h->lad), by_lv[1].size(), by_lv[2].size(), by_lv[3].size(), by_lv[4].size(), by_lv[5].size());
        const size_t nf = fails.size();
        if (c.work_cap < c.stage_cur + nf || c.hp_cap < c.stage_cur + nf) {
            // The staging of this context (device work lists, page-locked descriptor source of k_stage) has run out: rounds
            // still in flight hold its front.  New, larger blocks -- for the sibling retry ladder too, see lad_grow -- instead
            // of waiting for those rounds; the old blocks are released with the batch.
            const int k = int(&c - h->lad);
            for (int j = 0; j < 4; j++) {
                LadderCtx &g = h->lad[j];
                if (&g != &c && !(k < 2 && j == 1 - k)) continue;
                const size_t cap = std::max<size_t>(2 * (c.stage_cur + nf), 1 << 14);
                if (g.work_cap >= cap && g.hp_cap >= cap && &g != &c) continue;
                void *pd = nullptr, *pw = nullptr;
                int32_t *dw_ = nullptr;
                int rc = dev_alloc(h, &dw_, cap);
                if (rc) return rc;
                { int rc_pin = pin_alloc(h, &pd, cap * sizeof(AlnDesc)); if (rc_pin) return rc_pin; }
                { int rc_pin = pin_alloc(h, &pw, cap * sizeof(int32_t)); if (rc_pin) return rc_pin; }
                g.d_work = dw_; g.work_cap = cap;
                g.hp_descs = static_cast<AlnDesc *>(pd); g.hp_work = static_cast<int32_t *>(pw); g.hp_cap = cap;
                g.stage_cur = 0;
            }
        }
        bool zero_slots = true;
        std::vector<std::function<int()>> later;      // what a retry round enqueues behind its flag (below)
        c.plans.reserve(c.plans.size() + LV_DENSE + 1);    // (the closures index c.plans; no reallocation while they are pending)
        if (c.arena_cur == 0) drop_resident(c);
        if (!tie) hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, c.ls, h->d_tie_cnt + 5 + int(&c - h->lad), 0);   // the round's tie count
        for (int lv = LV_Q16; lv <= LV_DENSE; lv++) {
            if (by_lv[lv].empty()) continue;
            c.plans.emplace_back();
            Plan &P = c.plans.back();
            const int tag_or = tie ? TIE_TAG_BIT : 0;
            trace("    level %d: %zu alignments, planning", lv, by_lv[lv].size());
            int rc = make_plan(h, by_lv[lv], lv, P, c.arena + c.arena_cur, c.arena_bytes - c.arena_cur, tag_or);
            trace("    planned");
            if (rc == VPR_OK && P.chunks.size() > 1 && c.arena_cur > 0) rc = VPR_ERR_NOMEM;   // retry with the whole workspace
            if (rc == VPR_ERR_NOMEM && c.arena_cur > 0) {
                { const int rs = lad_sync(c); if (rs) return rs; }
                c.arena_cur = 0;
                drop_resident(c);
                rc = make_plan(h, by_lv[lv], lv, P, c.arena, c.arena_bytes, tag_or);
            }
            if (rc == VPR_OK && P.chunks.size() > 1 && h->cfg.workspace_bytes <= 0) {
                // the plan needs several passes through the ladder's workspace (it starts small), i.e. its launches run one
                // after the other with a few alignments each -- at the dense level that is one workgroup per alignment on a
                // 256-CU device.  Grow the workspace to hold the plan (a ladder whose rounds overlap hands its workspace out
                // once: twice the plan leaves room for the next round without waiting for this one), up to the bound fixed at
                // vpr_create.
                const int64_t nb = std::min<int64_t>((overlap(c) ? 2 : 1) * P.total_need + (1 << 20), h->lad_arena_max);
                if (nb > c.arena_bytes + c.arena_bytes / 2 && lad_grow(c, nb))
                    rc = make_plan(h, by_lv[lv], lv, P, c.arena, c.arena_bytes, tag_or);
            }
            if (rc == VPR_ERR_NOMEM && h->cfg.workspace_bytes <= 0) {
                // one alignment does not fit the ladder's workspace: grow it to twice that need
                if (lad_grow(c, std::max<int64_t>(2 * h->last_need, 2 * c.arena_bytes), h->last_need + (1 << 20)))
                    rc = make_plan(h, by_lv[lv], lv, P, c.arena, c.arena_bytes, tag_or);
            }
            if (rc) return rc;
            if (P.chunks.size() == 1) c.arena_cur += round_up(P.arena_used, 256);
            else c.arena_cur = c.arena_bytes;    // multi-chunk plan: the whole workspace is in use
            // stage descriptors / work list (bump allocation: all plans of the round are in flight together)
            const size_t n = P.work.size();
            int32_t *dw = c.d_work + c.stage_cur;
            c.stage_cur += n;
            memcpy(c.hp_descs + (c.stage_cur - n), P.descs.data(), n * sizeof(AlnDesc));
            memcpy(c.hp_work + (c.stage_cur - n), P.work.data(), n * 4);
            hipLaunchKernelGGL(k_stage, blocks(int64_t(std::max<size_t>(n, LadderCtx::N_SLOTS))), dim3(256), 0, c.ls,
                               c.hp_descs + (c.stage_cur - n), c.hp_work + (c.stage_cur - n), int(n), h->d_descs, dw,
                               h->d_cnt + c.slot0, zero_slots ? LadderCtx::N_SLOTS : 0);
            zero_slots = false;
            h->dirty.insert(h->dirty.end(), P.work.begin(), P.work.end());
            trace("    staged");
            // A retry round whose plan is one workspace chunk: the forward sweeps (and fail lists) of all its plans first, then
            // the flag the host waits for, then the backward sweeps, tie lists and walks of what was accepted -- the host starts
            // the next round while those still run (on the ladder's other context when that is idle: the main loop).
            const bool defer = !tie && lv != LV_DENSE && P.chunks.size() == 1 && c.slot_cur + 2 <= LadderCtx::N_SLOTS && !h->no_round_overlap;
            const size_t pi_ = c.plans.size() - 1;
            if (lv == LV_DENSE) {
                // (several kernel classes: side by side on the class streams -- the largest alignment of the largest class is
                // the critical path of a batch of long alignments, it should not wait for the smaller classes)
                bool many = false;
                for (const Chunk &ch : P.chunks) many = many || ch.launches.size() > 1;
                if (!tie && !h->no_round_overlap) {
                    later.push_back([this, &c, pi_, dw, many, tag_or]() { return run_dense(c.plans[pi_], dw, overlap(c) ? c.ls2 : c.ls, !many, tag_or, true); });
                } else {
                    if ((rc = run_dense(P, dw, c.ls, !many || tag_or != 0, tag_or, !tie))) return rc;     // (a tie round's replays share one scratch)
                }
            } else if (defer) {
                const Chunk &ch = P.chunks[0];
                const int ci_k = int(&c - h->lad);
                for (int part = 0; part < 2; part++) {
                    const int32_t n_long = ch.n_long;
                    const int64_t off = part == 0 ? ch.work_off : ch.work_off + n_long;
                    const int32_t cnt = part == 0 ? n_long : ch.count - n_long;
                    if (cnt <= 0) continue;
                    const int slot = c.slot0 + c.slot_cur++;
                    const int64_t foff = c.fail_base + c.fail_cur;
                    const int64_t pc = ch.part_cells[part], pin = ch.part_in[part], pd = ch.part_dense[part];
                    if ((rc = enqueue_part(P, dw, off, cnt, lv, c.ls, slot, foff, part == 0, pc, pin, pd, -1, 1, nullptr, 0, tag_or))) return rc;
                    later.push_back([this, &c, pi_, dw, off, cnt, lv, slot, foff, part, pc, pin, pd, tag_or, ci_k]() {
                        const Plan &Pl = c.plans[pi_];
                        hipStream_t ks = overlap(c) ? c.ls2 : c.ls;
                        int r = enqueue_part(Pl, dw, off, cnt, lv, ks, slot, foff, part == 0, pc, pin, pd, -1, 2, nullptr, 0, tag_or);
                        if (r) return r;
                        ladder_collect(ci_k, dw + off, cnt, ks);
                        return enqueue_part(Pl, dw, off, cnt, lv, ks, slot, foff, part == 0, pc, pin, pd, -1, 4, nullptr, 0, tag_or);
                    });
                    c.pending.emplace_back(slot, foff);
                    c.fail_cur += cnt;
                }
            } else {
                for (const Chunk &ch : P.chunks) {
                    if (c.slot_cur + 2 > LadderCtx::N_SLOTS) {   // out of fail slots: drain what is in flight
                        std::vector<Plan> keep;
                        keep.swap(c.plans);
                        const int64_t ac = c.arena_cur;
                        const size_t sc_ = c.stage_cur;
                        if ((rc = lad_flush(c, carry))) return rc;
                        c.plans.swap(keep);
                        c.arena_cur = ac; c.stage_cur = sc_;
                        HIPCHK(h, hipMemsetAsync(h->d_cnt + c.slot0, 0, LadderCtx::N_SLOTS * 4, c.ls));
                    }
                    const int32_t n_long = ch.n_long;
                    if (n_long > 0) {
                        const int slot = c.slot0 + c.slot_cur++;
                        // tie round: (early replays,) forward sweep, replays + patch, then the rest; retry round: forward +
                        // backward sweep, the list of what that left to a tie round, walk + credit
                        for (int pi = 0; pi < 2; pi++) {
                            const int ph = tie ? (pi ? 6 : 1) : (pi ? 4 : 3);
                            if (tie && pi == 0 && (rc = tie_replay(P, ch.work_off, n_long, c.ls, true))) return rc;
                            if ((rc = enqueue_part(P, dw, ch.work_off, n_long, lv, c.ls, slot, c.fail_base + c.fail_cur, true,
                                                   ch.part_cells[0], ch.part_in[0], ch.part_dense[0], -1, ph, nullptr, 0, tag_or))) return rc;
                            if (tie && pi == 0 && ((rc = tie_replay(P, ch.work_off, n_long, c.ls, false)) || (rc = tie_patch(P, c.ls, LV_TAG[lv] | tag_or)))) return rc;
                            if (!tie && pi == 0) ladder_collect(int(&c - h->lad), dw + ch.work_off, n_long, c.ls);
                        }
                        c.pending.emplace_back(slot, c.fail_base + c.fail_cur);
                        c.fail_cur += n_long;
                    }
                    if (ch.count > n_long) {
                        const int slot = c.slot0 + c.slot_cur++;
                        for (int pi = 0; pi < 2; pi++) {
                            const int ph = tie ? (pi ? 6 : 1) : (pi ? 4 : 3);
                            if (tie && pi == 0 && (rc = tie_replay(P, ch.work_off + n_long, ch.count - n_long, c.ls, true))) return rc;
                            if ((rc = enqueue_part(P, dw, ch.work_off + n_long, ch.count - n_long, lv, c.ls, slot,
                                                   c.fail_base + c.fail_cur, false, ch.part_cells[1], ch.part_in[1], ch.part_dense[1], -1, ph, nullptr, 0, tag_or))) return rc;
                            if (tie && pi == 0 && ((rc = tie_replay(P, ch.work_off + n_long, ch.count - n_long, c.ls, false)) || (rc = tie_patch(P, c.ls, LV_TAG[lv] | tag_or)))) return rc;
                            if (!tie && pi == 0) ladder_collect(int(&c - h->lad), dw + ch.work_off + n_long, ch.count - n_long, c.ls);
                        }
                        c.pending.emplace_back(slot, c.fail_base + c.fail_cur);
                        c.fail_cur += ch.count - n_long;
                    }
                }
            }
            h->resident.emplace_back(P.work, P.arena);
        }
        post_flag(4 + int(&c - h->lad), c.ls);
        (void)hipStreamQuery(c.ls);     // flush
        if (!later.empty() && overlap(c)) {     // the back halves: on the side stream, behind the forward sweeps
            HIPCHK(h, hipEventRecord(c.ev2, c.ls));
            HIPCHK(h, hipStreamWaitEvent(c.ls2, c.ev2, 0));
        }
        for (auto &fn : later) { const int rc = fn(); if (rc) return rc; }
        if (!later.empty()) { (void)hipStreamQuery(c.ls); if (overlap(c)) (void)hipStreamQuery(c.ls2); }
        return VPR_OK;
    }

    // resident: the chunk of plan0 whose workspace is still intact (early rounds), or nullptr.  An alignment that was
    // accepted where plan0 placed it (also by the in-place 16-cell round) is replayed early, from that workspace.
    int tie_round(LadderCtx &LT, const int4 *lst_in, int32_t n, bool full, const Chunk *resident) {
        tie_ctx = &LT;
        tie_resident = full ? nullptr : resident;
        std::vector<int4> lst(lst_in, lst_in + n);
        std::sort(lst.begin(), lst.end(), [](const int4 &x, const int4 &y) { return x.x < y.x; });   // deterministic planning
        std::vector<int32_t> marked, carry;
        marked.reserve(size_t(n));
        tie_early.clear();
        for (const int4 &e : lst) {
            tie_s[e.x] = e.w;
            int lv = LV_DENSE;
            for (int k = 0; k < LV_DENSE; k++) if (LV_TAG[k] == e.y) lv = k;
            h->level[size_t(e.x)] = uint8_t(lv);
            marked.push_back(e.x);
            if (resident && !full && e.z > 0) {
                const int32_t pos = h->plan0_pos[size_t(e.x)];
                const AlnDesc &d0 = plan_desc(h, h->plan0, size_t(pos));
                const int dtag = d0.band_pad;
                // (the backward sweep over column strips leaves the forward flags of the cells it does not score, pr_strip.hip: an
                // early replay, which finds the consulted ties by their flag bytes, cannot read that workspace)
                const bool strips = d0.band_w == 0 && class_of(std::max(d0.Lq, d0.Lr)) >= STRIP_CLS && !h->no_strips;
                if (pos >= resident->work_off && pos < resident->work_off + resident->count && !strips &&
                    (dtag == e.y || (dtag == LV_TAG[LV_Z] && e.y == LV_TAG[LV_Q16])))
                    tie_early[e.x] = TieEarly{e.z, pos, &h->plan0, spec_set.count(e.x) ? 3 : 1};
            }
        }
        if (h->tie_jobs_cap < tie_job_cur + size_t(n)) {
            void *pj = nullptr;
            { int rc_pin = pin_alloc(h, &pj, size_t(n) * 4 * sizeof(TieJob)); if (rc_pin) return rc_pin; }    // (an outgrown block stays until the batch is released: a launch may still read it)
            h->hp_tie_jobs = static_cast<TieJob *>(pj);
            h->tie_jobs_cap = size_t(n) * 4;
            tie_job_cur = 0;
        }
        const size_t j0 = tie_job_cur;
        tie_full = full;
        if (h->debug) fprintf(stderr, "[vpr] tie round: %d alignments%s, first ids %d %d %d, last %d\n", n, full ? " (full logs)" : "", marked[0], marked[size_t(n) / 3], marked[size_t(n) / 2], marked.back());
        int rc_ = lad_start(LT, marked, carry, true);
        if (rc_) return rc_;
        if (!carry.empty()) return fail(h, VPR_ERR_STATE, "tie round: the re-run forward sweep rejected alignment %d", carry[0]);
        tie_job_blocks.emplace_back(h->hp_tie_jobs + j0, tie_job_cur - j0);
        tie_job_total += tie_job_cur - j0;
        return VPR_OK;
    }

    int tie_flush(LadderCtx &LT) {
        std::vector<int32_t> rejected;
        int rc_ = lad_flush(LT, rejected);
        if (rc_) return rc_;
        if (!rejected.empty()) return fail(h, VPR_ERR_STATE, "tie round: the re-run forward sweep rejected alignment %d", rejected[0]);
        return VPR_OK;
    }

    int spec_round(const int4 *lst_all, int32_t n_all) {
        // A replay costs one dependent pass per WAVE of the reference's expansion, i.e. per unit of distance: a head start pays for
        // the whole-genome batch's long chains (distance 8, nine waves), not for an alignment at distance 1 723 (a truth-only
        // 1.7 kb insertion inside a tandem repeat, joint_synth: 1 724 waves, 27 ms for a replay whose tie was never consulted --
        // and the step's tie round waits for the slowest speculative job).  Those wait for their backward sweep's verdict.
        std::vector<int4> kept;
        kept.reserve(size_t(n_all));
        for (int32_t k = 0; k < n_all; k++)
            if (lst_all[k].w <= SPEC_MAX_DIST) kept.push_back(lst_all[k]);
        const int4 *lst = kept.data();
        const int32_t n = int32_t(kept.size());
        if (n == 0) return VPR_OK;
        LadderCtx &LT = h->lad[2];
        tie_ctx = &LT;
        spec_plan = Plan();
        spec_plan.arena = P0.arena;
        tie_early.clear();
        for (int32_t k = 0; k < n; k++) {
            const int32_t a = lst[k].x;
            tie_s[a] = lst[k].w;
            spec_plan.work.push_back(a);
            spec_plan.descs.push_back(h->descs[size_t(a)]);
            tie_early[a] = TieEarly{0, h->plan0_pos[size_t(a)], &h->plan0, 2};
        }
        if (h->tie_jobs_cap < tie_job_cur + size_t(n)) {
            void *pj = nullptr;
            { int rc_pin = pin_alloc(h, &pj, size_t(n) * 4 * sizeof(TieJob)); if (rc_pin) return rc_pin; }
            h->hp_tie_jobs = static_cast<TieJob *>(pj);
            h->tie_jobs_cap = size_t(n) * 4;
            tie_job_cur = 0;
        }
        const size_t j0 = tie_job_cur;
        tie_full = false;
        if (h->debug) fprintf(stderr, "[vpr] speculative replays: %d long alignments\n", n);
        int rc_ = tie_replay(spec_plan, 0, n, LT.ls, true);
        if (rc_) return rc_;
        spec_slot = tie_patch_slot; spec_off = tie_patch_off; spec_cap = tie_patch_cap;
        tie_patch_slot = -1;
        HIPCHK(h, hipEventRecord(h->ev_spec, LT.ls2));
        (void)hipStreamQuery(LT.ls2);
        for (int32_t k = 0; k < n; k++) spec_set[lst[k].x] = 1;
        tie_early.clear();
        tie_job_blocks.emplace_back(h->hp_tie_jobs + j0, tie_job_cur - j0);
        tie_job_total += tie_job_cur - j0;
        return VPR_OK;
    }

    // part k's alignments left to a tie round -> pinned host memory; ev_tie[k] marks the list complete
    int collect_part(int k, const int32_t *list, int32_t n, hipStream_t ks) {
        hipLaunchKernelGGL(k_collect_ties_list, blocks(n), dim3(256), 0, ks, list, n, h->d_outs, h->d_tie_list + tie_off[k],
                           h->d_tie_cnt + 2 + k, tie_cap[k]);
        hipLaunchKernelGGL(k_publish_ties, dim3(4), dim3(256), 0, ks, h->d_tie_list + tie_off[k], h->d_tie_cnt + 2 + k,
                           h->hp_tie_list + tie_off[k], h->hp_tie_cnt + 2 + k, tie_cap[k]);
        post_flag(2 + k, ks);
        return VPR_OK;
    }

    // ---- round 0 of a windowed plan and everything that runs beside it
    int round0_windowed() {
        int rc = VPR_OK;
        // Per chunk of the round-0 plan: the short alignments (a throughput problem) and the long ones (latency
        // chains: rows are sequential) run on two streams; the ids rejected by the exit test are known right
        // after each forward sweep, and their retry ladders run beside the rest of the round; the alignments whose
        // backward sweep met a tied swap cell are known right after that sweep, and their tie rounds run beside it too.
        // HIP maps streams onto 4 hardware queues: round 0 uses two, the ladders and the tie rounds share the others
        LL.ls = h->cls_stream[2]; LS.ls = h->cls_stream[3];
        // (side streams for the back halves of the retry rounds: the class streams of the widest dense classes, which the
        // column strips leave unused)
        LL.ls2 = h->cls_stream[5]; LS.ls2 = h->cls_stream[6]; LL.ev2 = h->ev_side[0]; LS.ev2 = h->ev_side[1];
        LL.slot0 = 2; LS.slot0 = 2 + LadderCtx::N_SLOTS;
        for (size_t ci = 0; ci < P0.chunks.size(); ci++) {
            const Chunk &ch = P0.chunks[ci];
            const int32_t n_long = ch.n_long;
            // (nothing of an earlier chunk or execute is in flight: the retry ladders hand out their workspaces from the front)
            LL.arena_cur = LS.arena_cur = 0; LL.stage_cur = LS.stage_cur = 0;
            const int64_t rbase = na_ + na_ / 16 + 256;                    // start of the retry rounds' fail region
            LL.fail_base = rbase; LS.fail_base = rbase + n_long;
            HIPCHK(h, hipMemsetAsync(h->d_cnt, 0, 8, st));
            if (ci == 0 && h->d_d1_info) HIPCHK(h, hipMemsetAsync(h->d_d1_info, 0, 64, st));
            HIPCHK(h, hipMemsetAsync(h->d_tie_cnt, 0, 32, st));
            HIPCHK(h, hipMemsetAsync(h->d_tie_ndec, 0, TIE_DEC_SLOTS * 4, st));
            // the previous chunk has joined: its decision lists (slots of d_tie_ndec, regions of d_tie_dec) are free again
            tie_dec_slot = 0; tie_dec_cur = 0; tie_patch_slot = -1; spec_slot = -1;
            HIPCHK(h, hipEventRecord(h->ev_fork, st));
            HIPCHK(h, hipStreamWaitEvent(s_long, h->ev_fork, 0));
            HIPCHK(h, hipStreamWaitEvent(s_short, h->ev_fork, 0));
            HIPCHK(h, hipStreamWaitEvent(LL.ls, h->ev_fork, 0));
            HIPCHK(h, hipStreamWaitEvent(LS.ls, h->ev_fork, 0));
            for (int k = 0; k < 4; k++) HIPCHK(h, hipStreamWaitEvent(h->tie_stream[k], h->ev_fork, 0));
            for (int k = 2; k < 4; k++)          // preset the replay scratches beside round 0 (see LadderCtx::tie_clean)
                for (int e = 0; e < 2; e++) {
                    LadderCtx &c = h->lad[k];
                    c.tie_clean[e] = 0;
                    if (ci == 0 && c.tie_scratch[e] && c.tie_first[e] > 0 && c.tie_first[e] <= c.tie_scratch_bytes[e]) {
                        HIPCHK(h, hipMemsetAsync(c.tie_scratch[e], 0xff, size_t(c.tie_first[e]), e ? c.ls2 : c.ls));
                        c.tie_clean[e] = c.tie_first[e];
                    }
                    c.tie_first_seen[e] = false;
                }
            // Round 0 of the short part.  At LV_Z, what the zero-distance sweep rejects (every alignment with s > 0)
            // re-runs *in place* with the general 16-cell kernels on the same stream, phase by phase behind the
            // zero-distance kernels: same layout and workspace slots, the device-built fail list is the work list and
            // its length stays on the device, so the host plans and copies nothing.  (A separate stream would not
            // help: the bulk kernels' millions of workgroups starve a concurrent launch until they drain.)
            const int32_t n_short = ch.count - n_long;
            bool wait_spec = false;
            spec_set.clear(); spec_slot = -1;
            const int SLOT_IP = LS.slot0 + LadderCtx::N_SLOTS - 1;          // fail slot of the in-place round
            const int64_t foff_ip = rbase + na_ - n_short;                 // tail of the ladders' fail region
            const int32_t cap_ip = std::min<int32_t>(n_short, std::max<int32_t>(4096, (n_short / 4 + 3) & ~3));
            const bool inplace = P0.lv == LV_Z && n_short > 0;
            const int32_t *short_list = P0.d_work + ch.work_off + n_long;
            if (n_short > 0 && !inplace) {
                for (int ph = 1; ph <= 4; ph <<= 1) {   // forward sweep, the fail list, backward sweep, the tie list, walk + credit
                    if ((rc = enqueue_part(P0, P0.d_work, ch.work_off + n_long, n_short, P0.lv, s_short, 1, n_long, false,
                                           ch.part_cells[1], ch.part_in[1], ch.part_dense[1], -1, ph))) return rc;
                    if (ph == 1) post_flag(1, s_short);
                    if (ph == 2 && (rc = collect_part(1, short_list, n_short, s_short))) return rc;
                }
            }
            if (inplace) {
                zl_wave0 = h->zl_wave0[ci];
                zl_rows = ch.part_rows[1];
                const int32_t *n_dev = h->d_cnt + 1;
                const int ztag = LV_TAG[LV_Z];
                HIPCHK(h, hipMemsetAsync(h->d_cnt + SLOT_IP, 0, 4, s_short));
                // Between the two: the distance-1 lane level (pr_d1.hip) on the zero level's fail list; what it leaves is a second
                // device-built list, the in-place round's.
                const bool d1 = h->d_d1_hdr != nullptr && cap_ip <= h->d1_wave_cap * 64;
                const int32_t n_all1 = n_short + n_short / 16 + 64;
                if (d1) HIPCHK(h, hipMemsetAsync(h->d_d1_info + 2, 0, 4, s_short));
                // The credit walks of what the two lane kernels finished only need those kernels: on a side stream beside the in-place
                // 16-cell round instead of behind it (the short part is one of the step's three chains)
                hipStream_t s_cred = h->side_credit ? h->cls_stream[7] : s_short;       // (a dense class's stream: idle in a windowed round 0)
                for (int ph = 1; ph <= 4; ph <<= 1) {
                    if (ph != 4 || s_cred == s_short) {
                        if ((rc = enqueue_part(P0, P0.d_work, ch.work_off + n_long, n_short, LV_Z, s_short, 1, n_long, false,
                                               ch.part_cells[1], ch.part_in[1], ch.part_dense[1], -1, ph))) return rc;
                        if (d1 && (rc = enqueue_d1(h->d_fail + n_long, n_dev, cap_ip, n_all1, s_short, ph, ztag))) return rc;
                    }
                    if (ph == 1 && s_cred != s_short) {
                        // (forking right behind the zero level instead -- its walk beside the distance-1 level -- slows `k_prep_d1` from
                        // 0.7 to 1.8 ms and the step by 0.5: the walk's 62 000 waves take the slots the chain's next launch wants)
                        HIPCHK(h, hipEventRecord(h->ev_cred[0], s_short));
                        HIPCHK(h, hipStreamWaitEvent(s_cred, h->ev_cred[0], 0));
                        if ((rc = enqueue_part(P0, P0.d_work, ch.work_off + n_long, n_short, LV_Z, s_cred, 1, n_long, false,
                                               ch.part_cells[1], ch.part_in[1], ch.part_dense[1], -1, 4))) return rc;
                        if (d1 && (rc = enqueue_d1(h->d_fail + n_long, n_dev, cap_ip, n_all1, s_cred, 4, ztag))) return rc;
                        HIPCHK(h, hipEventRecord(h->ev_cred[1], s_cred));
                    }
                    if (ph == 4 && s_cred != s_short) HIPCHK(h, hipStreamWaitEvent(s_short, h->ev_cred[1], 0));
                    // entries past cap_ip (more than a quarter of the part rejected) stay rejected and go to the ladder
                    if ((rc = d1 ? enqueue_part(P0, h->d_d1_fail, 0, cap_ip, LV_Q16, s_short, SLOT_IP, foff_ip, false, 0, 0, 0,
                                                ztag, ph, h->d_d1_info + 2, std::min<int32_t>(h->d1_fail_cap, n_all1 + n_all1 / 64 + 64))
                                 : enqueue_part(P0, h->d_fail, n_long, cap_ip, LV_Q16, s_short, SLOT_IP, foff_ip, false, 0, 0, 0,
                                                ztag, ph, n_dev, n_all1))) return rc;
                    if (ph == 1) post_flag(1, s_short);
                    if (ph == 2 && (rc = collect_part(1, short_list, n_short, s_short))) return rc;
                }
            }
            if (n_long > 0) {
                const int lv = P0.lv <= LV_Q16 ? LONG_LV : P0.lv;
                for (int ph = 1; ph <= 4; ph <<= 1) {
                    fwd_save_delta = (ph == 1 && lv == LV_C1 && h->d_save) ? int64_t(reinterpret_cast<intptr_t>(h->d_save) - reinterpret_cast<intptr_t>(P0.arena)) : 0;
                    rc = enqueue_part(P0, P0.d_work, ch.work_off, n_long, lv, s_long, 0, 0, true, ch.part_cells[0],
                                      ch.part_in[0], ch.part_dense[0], -1, ph);
                    fwd_save_delta = 0;
                    if (rc) return rc;
                    if (ph == 1) {
                        post_flag(0, s_long);
                        if (lv == LV_C1) {
                            hipLaunchKernelGGL(k_collect_spec, blocks(n_long), dim3(256), 0, s_long, P0.d_work + ch.work_off, n_long, h->d_outs,
                                               LV_TAG[LV_C1], h->d_tie_list + tie_off[2], h->d_tie_cnt + 4, tie_cap[2]);
                            hipLaunchKernelGGL(k_publish_ties, dim3(4), dim3(256), 0, s_long, h->d_tie_list + tie_off[2], h->d_tie_cnt + 4,
                                               h->hp_tie_list + tie_off[2], h->hp_tie_cnt + 4, tie_cap[2]);
                            post_flag(8, s_long);
                            wait_spec = true;
                        }
                    }
                    if (ph == 2 && (rc = collect_part(0, P0.d_work + ch.work_off, n_long, s_long))) return rc;
                }
            }
            // (the runtime submits a stream's trailing marker lazily: without these flushes the fork event on the main
            // stream, on which everything above waits, is only submitted when the host next blocks on something)
            lapx("round 0 enqueued");
            if (ci == 0) phase(1);
            // ---- the host serves whatever is ready: a fail list of round 0 starts a ladder, a finished ladder round
            // starts the next, a published tie list starts a tie round; it never blocks on one while another is ready
            std::vector<int32_t> fails, carry[2];
            int64_t idle_polls = 0;
            auto idle_since = std::chrono::steady_clock::now();
            bool wait_fail[2] = {n_long > 0, n_short > 0}, wait_tie[2] = {n_long > 0, n_short > 0};
            while (wait_fail[0] || wait_fail[1] || wait_tie[0] || wait_tie[1] || wait_spec || lad_tie_wait[0] || lad_tie_wait[1] || !LL.pending.empty() || !LS.pending.empty() ||
                   !h->lad[2].pending.empty() || !h->lad[3].pending.empty()) {
                bool progressed = false;
                if (idle_polls == 0) idle_since = std::chrono::steady_clock::now();
                if (wait_spec && flag_up(8)) {
                    trace("flag 8 (speculative list), %d candidates", h->hp_tie_cnt[4]);
                    const int32_t n = std::min(h->hp_tie_cnt[4], tie_cap[2]);
                    // (a speculative replay pays off as a head start for a few long chains; when thousands of alignments carry
                    // tied cells most of them are never consulted, and the round waits for the backward sweep's marks instead)
                    if (n > 0 && n <= 1024 && (rc = spec_round(h->hp_tie_list + tie_off[2], n))) return rc;
                    lapx("speculative list -> replays");
                    wait_spec = false;
                    progressed = true;
                }
                for (int k = 0; k < 2; k++) {
                    LadderCtx &c = h->lad[k];
                    if (lad_tie_wait[k] && flag_up(9 + k)) {       // (before the ladder's next round reuses the region)
                        trace("flag %d (ladder %d tie list), %d marked", 9 + k, k, h->hp_tie_cnt[5 + k]);
                        const int32_t n = std::min(h->hp_tie_cnt[5 + k], lad_tie_cap(k));
                        if (n > 0) {
                            // (behind whatever the tie ladder's stream is still busy with: running this round on the ladder's side
                            // stream instead saved 3 ms of a lone step but showed a rare wrong walk in repeated executes -- removed)
                            LadderCtx &T = h->lad[2 + k];
                            if ((rc = tie_round(T, h->hp_tie_list + lad_tie_off(k), n, false, nullptr))) return rc;
                        }
                        lad_tie_wait[k] = false;
                        progressed = true;
                    }
                    if (wait_fail[k] && flag_up(k)) {
                        trace("flag %d (round 0 fail list, part %d)", k, k);
                        fails.clear();
                        if (k == 1 && inplace) {
                            if ((rc = read_fails(SLOT_IP, foff_ip, c.ls, fails))) return rc;
                            for (int32_t a : fails) h->level[size_t(a)] = uint8_t(LV_Q16);
                            n_retry += h->hp_cnt[1];   // rejected by the zero-distance sweep (published before the in-place
                                                       // round's own list, same stream; the count includes the list's -1 padding)
                        } else {
                            if ((rc = read_fails(k, k == 0 ? 0 : n_long, c.ls, fails))) return rc;
                        }
                        trace("  %zu rejected -> ladder %d", fails.size(), k);
                        if ((rc = lad_start(c, fails, carry[k]))) return rc;
                        trace("  ladder %d round enqueued", k);
                        lapx(k ? "short fail list -> ladder" : "long fail list -> ladder");
                        wait_fail[k] = false;
                        progressed = true;
                    } else if (!wait_fail[k] && !c.pending.empty() && flag_up(4 + k)) {
                        // The round's fail lists are complete (its backward sweeps and walks may still be running: lad_start).
                        // The next round goes to the other retry ladder's context when that is idle -- own stream, own
                        // workspace: it then runs beside the rest of this round instead of behind it -- else to this one.
                        // A context whose last tie list the host has not taken yet is not reused (the list would be overwritten).
                        const int o = 1 - k;
                        int tk = -1;
                        if (!h->no_round_overlap && h->lad[o].pending.empty() && !wait_fail[o] && !lad_tie_wait[o] && carry[o].empty()) tk = o;
                        else if (!lad_tie_wait[k]) tk = k;
                        if (tk >= 0) {
                            trace("flag %d (ladder %d round done)", 4 + k, k);
                            fails.clear();
                            fails.swap(carry[k]);
                            if ((rc = lad_flush(c, fails))) return rc;
                            trace("  %zu rejected -> ladder %d", fails.size(), tk);
                            if ((rc = lad_start(h->lad[tk], fails, carry[tk]))) return rc;
                            trace("  ladder %d round enqueued", tk);
                            progressed = true;
                        }
                    }
                    if (wait_tie[k] && flag_up(2 + k)) {
                        const int32_t n = std::min(h->hp_tie_cnt[2 + k], tie_cap[k]);
                        trace("flag %d (round 0 tie list, part %d), %d marked", 2 + k, k, n);
                        if (n > 0 && (rc = tie_round(h->lad[2 + k], h->hp_tie_list + tie_off[k], n, false, &ch))) return rc;
                        lapx(k ? "short tie list -> tie round" : "long tie list -> tie round");
                        wait_tie[k] = false;
                        progressed = true;
                    }
                }
                for (int k = 0; k < 2; k++)
                    if (!wait_tie[k] && !lad_tie_wait[k] && !h->lad[2 + k].pending.empty() && flag_up(6 + k)) {
                        trace("flag %d (tie ladder %d round done)", 6 + k, k);
                        if ((rc = tie_flush(h->lad[2 + k]))) return rc;
                        progressed = true;
                    }
                // (no busy wait: a spinning host thread can exhaust the process's CPU quota, which stalls the runtime's threads)
                if (progressed) idle_polls = 0;
                else {
                    if ((rc = idle_check(++idle_polls, idle_since))) return rc;
                    std::this_thread::sleep_for(std::chrono::microseconds(20));
                }
            }
            lapx("ladders and tie rounds drained");
            phase(2);
            // join: the next chunk reuses the arena
            HIPCHK(h, hipEventRecord(h->ev_join[0], s_long));
            HIPCHK(h, hipEventRecord(h->ev_join[1], s_short));
            HIPCHK(h, hipEventRecord(h->ev_join[2], LL.ls));
            HIPCHK(h, hipEventRecord(h->ev_join[3], LS.ls));
            HIPCHK(h, hipEventRecord(h->ev_join[4], h->lad[2].ls));
            HIPCHK(h, hipEventRecord(h->ev_join[5], h->lad[3].ls));
            HIPCHK(h, hipEventRecord(h->ev_join[6], LL.ls2));
            HIPCHK(h, hipEventRecord(h->ev_join[7], LS.ls2));
            for (int k = 0; k < 8; k++) HIPCHK(h, hipStreamWaitEvent(st, h->ev_join[k], 0));
            if (ci + 1 < P0.chunks.size()) HIPCHK(h, x_sync(h, st, SITE));
            if (ci + 1 == P0.chunks.size()) { h->res0_off = ch.work_off; h->res0_cnt = ch.count; }
        }
        return VPR_OK;
    }

    int final_tie_pass() {
        int rc = VPR_OK;
        // ---- final tie pass: whatever is still marked (ties met by the retry ladders' rounds, lists that outgrew their
        // buffer, replays whose capped FIFO logs overflowed: those get worst-case logs now)
        {
            const int na = int(h->descs.size());
            for (int iter = 0; na > 0; iter++) {
                HIPCHK(h, hipMemsetAsync(h->d_tie_cnt, 0, 8, st));
                hipLaunchKernelGGL(k_collect_ties, blocks(na), dim3(256), 0, st, h->d_outs, na, h->d_tie_list, h->d_tie_cnt, h->tie_list_cap);
                int32_t n_mark = 0;
                HIPCHK(h, hipMemcpyAsync(&n_mark, h->d_tie_cnt, 4, hipMemcpyDeviceToHost, st));
                // (the number of deferred edit distances rides along: when nothing is marked it is final, and K4 needs no wait of its own)
                HIPCHK(h, hipMemcpyAsync(&n_jobs_pre, h->d_njobs, 4, hipMemcpyDeviceToHost, st));
                HIPCHK(h, x_sync(h, st, SITE));
                n_jobs_final = (n_mark == 0);
                if (n_mark == 0) break;
                if (iter >= 3) return fail(h, VPR_ERR_STATE, "tie pass: %d alignments still marked after %d attempts", n_mark, iter);
                const int32_t n = std::min(n_mark, h->tie_list_cap);
                std::vector<int4> lst;
                lst.resize(size_t(n));
                HIPCHK(h, hipMemcpy(lst.data(), h->d_tie_list, size_t(n) * sizeof(int4), hipMemcpyDeviceToHost));
                if (h->debug) fprintf(stderr, "[vpr] final tie pass %d: %d alignments marked\n", iter, n_mark);
                if ((rc = tie_round(h->lad[2], lst.data(), n, iter > 0, nullptr))) return rc;
                if ((rc = tie_flush(h->lad[2]))) return rc;
                HIPCHK(h, x_sync(h, h->lad[2].ls, SITE));
            }
        }
        return rc;
    }

    void debug_replays() {
        if (h->debug && tie_job_total > 0) {   // the slowest replays of the execute
            std::vector<TieJob> js;
            for (const auto &blk : tie_job_blocks) js.insert(js.end(), blk.first, blk.first + blk.second);
            std::sort(js.begin(), js.end(), [](const TieJob &x, const TieJob &y) { return x.dbg_us > y.dbg_us; });
            int64_t tot_us = 0;
            for (const TieJob &J : js) tot_us += J.dbg_us;
            size_t nf1 = 0, nf2 = 0;
            for (const TieJob &J : js) { nf1 += J.pad == 1; nf2 += J.pad == 2; }
            fprintf(stderr, "[vpr] tie replay: %zu jobs, %.1f ms summed; gave up: %zu (logs), %zu (stamp grid)\n", js.size(), tot_us / 1000.0, nf1, nf2);
            for (const TieJob &J : js)
                if (J.pad == 2 && nf2-- < 4) {
                    const AlnDesc &d = h->descs[size_t(J.a)];
                    fprintf(stderr, "[vpr]   grid too small: sc %d aln %d Lq %d Lr %d Lt %d, diagonals [%d, +%d) [%d, +%d), spans q (%d, %d) t (%d, %d)\n", d.sc, d.aln, d.Lq, d.Lr, d.Lt,
                            J.dlo[0], J.dn[0], J.dlo[1], J.dn[1], h->hp_dspan[d.qs][d.sc].x, h->hp_dspan[d.qs][d.sc].y, h->hp_dspan[d.ts][d.sc].x, h->hp_dspan[d.ts][d.sc].y);
                    fprintf(stderr, "[vpr]     first cell outside: plane %d position %d row %d (wave %d), mode %d\n", J.dbg_oob[0], J.dbg_oob[1], J.dbg_oob[2], J.dbg_oob[3], J.mode);
                }
            for (size_t k = 0; k < js.size() && k < 8; k++) {
                const AlnDesc &d = h->descs[size_t(js[k].a)];
                fprintf(stderr, "[vpr]   sc %d aln %d Lq %d Lr %d Lt %d level %d: %d us, %d waves, %d BFS steps, %d cells; mode %d, consulted %d, decided %d (last in wave %d)\n", d.sc, d.aln,
                        d.Lq, d.Lr, d.Lt, int(h->level[size_t(js[k].a)]), js[k].dbg_us, js[k].dbg_waves, js[k].dbg_steps, js[k].dbg_cells,
                        js[k].mode, js[k].n_used, js[k].dbg_nres, js[k].dbg_lastw);
                fprintf(stderr, "[vpr]     us: BFS %d, patch %d, order A %d B %d suffix %d C %d, seeding %d, setup %d\n", js[k].dbg_t[0], js[k].dbg_t[1],
                        js[k].dbg_t[2], js[k].dbg_t[3], js[k].dbg_t[4], js[k].dbg_t[5], js[k].dbg_t[6], js[k].dbg_t[7]);
                fprintf(stderr, "[vpr]     narrow steps %d: %d not in one row, %d shrinking or growing, %d not reproduced, %d look-ahead steps committing %d levels\n",
                        js[k].dbg_la[0], js[k].dbg_la[1], js[k].dbg_la[2], js[k].dbg_la[3], js[k].dbg_la[4], js[k].dbg_la[5]);
            }
        }
    }

    // (VPR_DEBUG) where the alignments ended up: per final level, how many, their dense cells, and their distances
    void debug_levels() {
        if (!h->debug || h->descs.empty()) return;
        std::vector<AlnOut> outs(h->descs.size());
        if (hipMemcpy(outs.data(), h->d_outs, outs.size() * sizeof(AlnOut), hipMemcpyDeviceToHost) != hipSuccess) return;
        static const int32_t edge[6] = {0, 4, 16, 64, 256, 1024};
        for (int lv = 0; lv <= LV_DENSE; lv++) {
            int64_t n[7] = {0, 0, 0, 0, 0, 0, 0};
            double cells[7] = {0, 0, 0, 0, 0, 0, 0};
            for (size_t a = 0; a < outs.size(); a++) {
                if (h->level[a] != lv) continue;
                const AlnDesc &d = h->descs[a];
                int b = 0;
                while (b < 6 && outs[a].s > edge[b]) b++;
                n[b]++;
                cells[b] += double(d.Lq + d.Lr) * d.Lt;
            }
            int64_t tot = 0;
            for (int b = 0; b < 7; b++) tot += n[b];
            if (!tot) continue;
            fprintf(stderr, "[vpr] level %d: %lld alignments; by distance (<=0, <=4, <=16, <=64, <=256, <=1024, more): ", lv, (long long)tot);
            for (int b = 0; b < 7; b++) fprintf(stderr, "%lld (%.2e cells)%s", (long long)n[b], cells[b], b < 6 ? ", " : "\n");
        }
    }

    // K4: deferred section edit distances
    int deferred_edit_distances() {
        int rc = VPR_OK;
        // K4: deferred section edit distances
        int32_t n_jobs = n_jobs_pre;
        if (!n_jobs_final) {
            HIPCHK(h, hipMemcpyAsync(&n_jobs, h->d_njobs, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(h, x_sync(h, st, SITE));
        }
        n_jobs = std::min(n_jobs, h->jobs_cap);
        // the bit-parallel kernel (pr_ed.hip: a wavefront per section, no host pass over the jobs) unless the batch holds a string
        // its LDS bit planes cannot carry (VPR_ED_WF / VPR_ED_DIAG: the older kernels, which stay as that fallback)
        if (h->ed_max_len < 0) {
            int64_t mx = 0;
            const size_t nsc = h->descs.size() / 4;
            for (size_t sc = 0; sc < nsc; sc++) {
                int32_t Lq, Lr, Lt;
                h->descs.lens(sc * 4, Lq, Lr, Lt);          // (ref and truth segments are what a section compares)
                mx = std::max<int64_t>(mx, std::max(Lr, Lt));
                h->descs.lens(sc * 4 + 1, Lq, Lr, Lt);
                mx = std::max<int64_t>(mx, Lt);
            }
            h->ed_max_len = mx;
        }
        if (n_jobs > 0 && h->ed_max_len <= EDB_MAX_TEXT && !getenv("VPR_ED_WF") && !getenv("VPR_ED_DIAG")) {
            vpr_launch_stat es_;
            memset(&es_, 0, sizeof(es_));
            es_.threads = 64; es_.n_units = n_jobs;
            return timed(4, es_, st, "k_ed_bits", [&] {
                hipLaunchKernelGGL(k_ed_bits, dim3(n_jobs), dim3(64), 0, st, h->dB, h->d_descs, h->d_jobs, n_jobs, h->d_secs);
            });
        }
        if (n_jobs > 0) {
            std::vector<EdJob> jobs(n_jobs);
            HIPCHK(h, hipMemcpy(jobs.data(), h->d_jobs, size_t(n_jobs) * sizeof(EdJob), hipMemcpyDeviceToHost));
            int64_t stride = 0, max_short = 0, max_sum = 0;
            for (const EdJob &j : jobs) {
                stride = std::max<int64_t>(stride, std::max(j.ref_len, j.tru_len) + 1);
                max_short = std::max<int64_t>(max_short, std::min(j.ref_len, j.tru_len));
                max_sum = std::max<int64_t>(max_sum, int64_t(j.ref_len) + j.tru_len);
            }
            // wavefront kernel when the largest section's furthest-reaching rows fit LDS; else the anti-diagonal kernel when its
            // diagonals do (16-bit distances); else the row sweep through global memory
            const int64_t max_long = stride - 1;
            const size_t lds_wf = size_t(2 * (2 * max_long + 3) * 2 + max_sum + 16);
            const int64_t pitch = (max_short + 2 + 7) & ~int64_t(7);
            const size_t lds_diag = size_t(3 * pitch * 2 + max_sum + 16);
            if (lds_wf <= 150 * 1024 && max_long < 29000 && !getenv("VPR_ED_DIAG")) {     // (VPR_ED_WF lands here)
                // size classes by the longer string (x 4 from class to class), largest first
                void *ps = nullptr, *qs = nullptr;
                { int rc_pin = exec_pin(h, &ps, size_t(n_jobs) * 4); if (rc_pin) return rc_pin; }
                if ((rc = exec_alloc(h, &qs, size_t(n_jobs) * 4))) return rc;
                int32_t *h_sel = static_cast<int32_t *>(ps), *d_sel = static_cast<int32_t *>(qs);
                std::vector<int32_t> order(static_cast<size_t>(n_jobs));
                for (int32_t k = 0; k < n_jobs; k++) order[size_t(k)] = k;
                auto longer = [&](int32_t k) { return std::max(jobs[size_t(k)].ref_len, jobs[size_t(k)].tru_len); };
                std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return longer(x) > longer(y); });
                std::copy(order.begin(), order.end(), h_sel);
                HIPCHK(h, hipMemcpyAsync(d_sel, h_sel, size_t(n_jobs) * 4, hipMemcpyHostToDevice, st));
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ed_wf), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_wf));
                HIPCHK(h, hipEventRecord(h->ev_fork, st));
                int n_cls = 0;
                for (int32_t k0 = 0; k0 < n_jobs;) {
                    hipStream_t ks = h->cls_stream[n_cls % N_CLASSES];      // (the classes side by side)
                    HIPCHK(h, hipStreamWaitEvent(ks, h->ev_fork, 0));
                    const int64_t top = longer(order[size_t(k0)]);
                    int32_t k1 = k0;
                    int64_t sum = 0;
                    while (k1 < n_jobs && longer(order[size_t(k1)]) * 4 > top) {
                        sum = std::max<int64_t>(sum, int64_t(jobs[size_t(order[size_t(k1)])].ref_len) + jobs[size_t(order[size_t(k1)])].tru_len);
                        k1++;
                    }
                    const size_t lds = size_t(2 * (2 * top + 3) * 2 + sum + 16);
                    vpr_launch_stat es_;
                    memset(&es_, 0, sizeof(es_));
                    es_.threads = EDW_NT; es_.n_units = k1 - k0;
                    rc = timed(4, es_, ks, "k_ed_wf", [&] {
                        hipLaunchKernelGGL(k_ed_wf, dim3(k1 - k0), dim3(EDW_NT), lds, ks, h->dB, h->d_descs, h->d_jobs, d_sel + k0, k1 - k0,
                                           h->d_secs, int(top));
                    });
                    if (rc) return rc;
                    HIPCHK(h, hipEventRecord(h->ev_join[n_cls % N_CLASSES], ks));
                    HIPCHK(h, hipStreamWaitEvent(st, h->ev_join[n_cls % N_CLASSES], 0));
                    n_cls++;
                    k0 = k1;
                }
            } else if (lds_diag <= 150 * 1024 && max_sum < 65000) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ed_diag), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_diag));
                vpr_launch_stat es_;
                memset(&es_, 0, sizeof(es_));
                es_.threads = ED_NT; es_.n_units = n_jobs;
                rc = timed(4, es_, st, "k_ed_diag", [&] {
                    hipLaunchKernelGGL(k_ed_diag, dim3(n_jobs), dim3(ED_NT), lds_diag, st, h->dB, h->d_descs, h->d_jobs, n_jobs,
                                       h->d_secs, int(max_short));
                });
                if (rc) return rc;
            } else {
            stride = round_up(stride, 16);
            const int64_t max_ints = int64_t(1) << 28;   // 1 GiB of scratch per slice
            const int32_t per = int32_t(std::max<int64_t>(1, std::min<int64_t>(n_jobs, max_ints / stride)));
            if (h->ed_scratch_ints < size_t(per * stride)) {
                int32_t *p;
                if ((rc = dev_alloc(h, &p, size_t(per * stride)))) return rc;
                h->d_ed_scratch = p;
                h->ed_scratch_ints = size_t(per * stride);
            }
            for (int32_t j0 = 0; j0 < n_jobs; j0 += per) {
                const int32_t cnt = std::min(per, n_jobs - j0);
                vpr_launch_stat es_;
                memset(&es_, 0, sizeof(es_));
                es_.threads = 64; es_.n_units = cnt;
                rc = timed(4, es_, st, "k_ed", [&] {
                    hipLaunchKernelGGL(k_ed, dim3(cnt), dim3(64), 0, st, h->dB, h->d_descs, h->d_jobs + j0, cnt,
                                       h->d_secs, h->d_ed_scratch, stride);
                });
                if (rc) return rc;
            }
            }
        }
        return rc;
    }

    // K5 (per-variant results, phase, tally) and the call's timing record
    int finish() {
        int rc = VPR_OK;
        // K5: per-variant results, phase, tally
        {
            const int na = int(h->descs.size());
            vpr_launch_stat fs_;
            memset(&fs_, 0, sizeof(fs_));
            fs_.threads = 128; fs_.n_units = na;
            rc = timed(5, fs_, st, "k_finalize+k_phase_tally", [&] {
                if (na && h->n_aliased) hipLaunchKernelGGL(k_alias_outs, blocks(int64_t(na)), dim3(256), 0, st, h->d_alias, h->n_sc, h->d_outs);
                if (na) hipLaunchKernelGGL(k_finalize, dim3((na + 127) / 128), dim3(128), 0, st, h->d_descs, na, h->d_outs,
                                           h->d_secs, h->d_fp_table, h->dR, h->dB.sc_limit, h->d_alias);
                if (h->n_sc) hipLaunchKernelGGL(k_phase_tally, dim3((h->n_sc + 127) / 128), dim3(128), 0, st, h->d_descs,
                                                h->n_sc, h->dR);
            });
            if (rc) return rc;
        }
        if (h->d_d1_info) HIPCHK(h, hipMemcpyAsync(h->hp_tie_cnt + 24, h->d_d1_info, 32, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipEventRecord(t1, st));
        phase(4);
        HIPCHK(h, hipStreamSynchronize(st));       // (the one wait every execute has: not counted as "blocked")
        phase(5);
        lapx("done");
        HIPCHK(h, hipGetLastError());
        (void)need_err_check;      // (an alignment neither the strips nor one workgroup can take carries VPR_ST_ERR_LIMIT: k_strip_plan)
        float ms = 0;
        (void)hipEventElapsedTime(&ms, t0, t1);
        h->timing.ms_total = ms;
        h->timing.ms_fwd = h->timing.ms_bwd = h->timing.ms_walk = h->timing.ms_ed = h->timing.ms_tie = 0;
        for (auto &e : h->events) {
            float m = 0;
            (void)hipEventElapsedTime(&m, e.a, e.b);
            e.st.ms = m;
            if (e.kind == 1) h->timing.ms_fwd += m;
            else if (e.kind == 2) h->timing.ms_bwd += m;
            else if (e.kind == 3) h->timing.ms_walk += m;
            else if (e.kind == 4) h->timing.ms_ed += m;
            else if (e.kind == 6) h->timing.ms_tie += m;
        }
        h->timing.n_fwd_launches = n_fwd;
        h->timing.cells_touched = cells_touched;
        h->timing.n_band_retries = n_retry;
        h->timing.n_tie_replays = n_tie_jobs;
        h->timing.n_alignments_computed = int64_t(h->descs.size()) - h->n_aliased;
        h->timing.n_lane1_seen = h->d_d1_info ? h->hp_tie_cnt[24 + 3] : 0;
        h->timing.n_lane1_finished = h->d_d1_info ? h->hp_tie_cnt[24 + 4] : 0;
        h->timing.n_lane1_waves_dropped = h->d_d1_info ? h->hp_tie_cnt[24 + 1] : 0;
#ifdef D1_CLOCKS
        if (h->debug && h->d_d1_info) {
            int32_t ck[8];
            (void)hipMemcpy(ck, h->d_d1_info + 8, 32, hipMemcpyDeviceToHost);
            fprintf(stderr, "[vpr] k_one_lane passes, longest wave (us): %.1f %.1f %.1f %.1f; pass 1 wave-rows %d of which four slots %d; pass 3 %d / %d\n", ck[0] * 0.01, ck[1] * 0.01,
                    ck[2] * 0.01, ck[3] * 0.01, ck[4], ck[5], ck[6], ck[7]);
        }
#endif
        if (h->debug && h->d_d1_info) fprintf(stderr, "[vpr] distance-1 lane level: %d waves, %d dropped, %d left; %d seen, %d finished (pass 1 %d, pass 2 %d, end cell %d)\n",
                                              h->hp_tie_cnt[24], h->hp_tie_cnt[25], h->hp_tie_cnt[26], h->hp_tie_cnt[27], h->hp_tie_cnt[28], h->hp_tie_cnt[29], h->hp_tie_cnt[30], h->hp_tie_cnt[31]);
        h->timing.ms_wall = phase_ms[5];
        for (int k = 0; k < 6; k++) h->timing.ms_wall_phase[k] = phase_ms[k];
        h->timing.ms_host_alloc = h->hs.ms_alloc;
        h->timing.ms_host_blocked = h->hs.ms_sync;
        h->timing.n_device_allocs = h->hs.n_dev_alloc;
        h->timing.n_device_frees = h->hs.n_dev_free;
        h->timing.n_host_allocs = h->hs.n_pin_alloc;
        if (h->stall_log && phase_ms[5] > stall_dump_ms) {      // the device's timeline of a slow execute, and what the host did when
            std::vector<std::tuple<float, float, const char *, int>> tl;
            for (auto &e : h->events) {
                float a = 0, d = 0;
                (void)hipEventElapsedTime(&a, t0, e.a);
                (void)hipEventElapsedTime(&d, e.a, e.b);
                tl.emplace_back(a, d, e.st.kernel, e.st.n_units);
            }
            std::sort(tl.begin(), tl.end());
            for (auto &x : tl) fprintf(stderr, "[vpr]   %p dev %9.2f ms +%8.2f  %-24s %d\n", (void *)h, std::get<0>(x), std::get<1>(x), std::get<2>(x), std::get<3>(x));
            for (auto &x : trace_) fprintf(stderr, "[vpr]   %p host %9.2f ms  %s\n", (void *)h, x.first, x.second.c_str());
        }
        if (h->stall_log)
            fprintf(stderr, "[vpr] execute %p: wall %.1f ms (reset %.1f, round 0 enqueued %.1f, drained %.1f, ties %.1f, K4/K5 enqueued %.1f), kernels %.1f ms, "
                            "allocator %.1f ms (%lld hipMalloc, %lld hipFree, %lld hipHostMalloc), blocked %.1f ms\n", (void *)h, phase_ms[5], phase_ms[0],
                    phase_ms[1], phase_ms[2], phase_ms[3], phase_ms[4], ms, h->hs.ms_alloc, (long long)h->hs.n_dev_alloc, (long long)h->hs.n_dev_free,
                    (long long)h->hs.n_pin_alloc, h->hs.ms_sync);
        return VPR_OK;
    }

    int run() {
        HIPCHK(h, hipSetDevice(h->cfg.device));
        h->events.clear();
        h->ev_used = 0;
        {
            InitArgs IA;
            IA.outs = h->d_outs; IA.na = int64_t(h->descs.size());
            int64_t top = std::max<int64_t>(IA.na, 8);
            for (int q = 0; q < 4; q++) {
                IA.fp[q] = h->d_fp[q]; IA.n_fp[q] = h->n_var[q >> 1];
                IA.n_var[q] = h->n_var[q];
                for (int w = 0; w < 2; w++) IA.v[q][w] = h->dR.v[q][w];
                top = std::max(top, std::max(IA.n_fp[q], IA.n_var[q]));
            }
            IA.tally = h->dR.tally; IA.njobs = h->d_njobs; IA.cnt = h->d_cnt; IA.n_cnt = 2;
            hipLaunchKernelGGL(k_init_execute, blocks(top), dim3(256), 0, st, IA);
        }
        if (!h->dirty.empty()) {   // restore the round-0 descriptors and levels a previous execute's retry rounds replaced
            // (only those: a few thousand of a whole-genome batch's millions -- rewriting all of them was 0.5 GB of traffic per
            // execute, 0.4 ms alone and 1.5 ms beside another batch's kernels, on the host's way to the first launch)
            const size_t nd = h->dirty.size();
            bool some = nd * 8 < h->plan0.work.size() && h->plan0_pos.size() == h->descs.size();
            for (size_t k = 0; k < nd && some; k++) some = h->plan0_pos[size_t(h->dirty[k])] >= 0;
            if (some) {
                void *hp = nullptr, *dp = nullptr;
                { const int rc = exec_pin(h, &hp, nd * 4); if (rc) return rc; }
                { const int rc = exec_alloc(h, &dp, nd * 4); if (rc) return rc; }
                int32_t *hpos = static_cast<int32_t *>(hp);
                for (size_t k = 0; k < nd; k++) hpos[k] = h->plan0_pos[size_t(h->dirty[k])];
                HIPCHK(h, hipMemcpyAsync(dp, hp, nd * 4, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(k_scatter_some, blocks(int64_t(nd)), dim3(256), 0, st, h->plan0.d_descs, static_cast<const int32_t *>(dp),
                                   int(nd), h->d_descs);
            } else
            hipLaunchKernelGGL(k_scatter_descs, blocks(int64_t(h->plan0.work.size())), dim3(256), 0, st, h->plan0.d_descs,
                               int(h->plan0.work.size()), h->d_descs);
            for (int32_t a : h->dirty) h->level[size_t(a)] = h->level0[size_t(a)];     // (every level change goes through a plan)
            h->dirty.clear();
        }
        // (measured: without one blocking call here the runtime does not start this call's submissions for 0.1 - 2 s when the
        // host goes straight to polling memory below, e.g. right behind another library's work on the device)
        HIPCHK(h, x_sync(h, h->stream, SITE));
        wall0 = std::chrono::steady_clock::now();
        phase(0);
        HIPCHK(h, hipEventCreate(&t0));
        HIPCHK(h, hipEventCreate(&t1));
        HIPCHK(h, hipEventRecord(t0, st));
        for (int k = 0; k < 2; k++) {     // tie ladder k: rounds of the long / short part of round 0 (the final pass uses 0)
            LadderCtx &c = h->lad[2 + k];
            c.ls = h->tie_stream[2 * k]; c.ls2 = h->tie_stream[2 * k + 1]; c.ev2 = h->ev_tie2[k];
            c.slot0 = 2 + (2 + k) * LadderCtx::N_SLOTS;
            c.fail_base = (2 + k) * int64_t(h->descs.size()) + int64_t(h->descs.size()) / 16 + 256;
        }
        int rc = VPR_OK;
        h->resident.clear();
        h->res0_cnt = 0;
        if (h->cfg.band_mode == 0) {
            if ((rc = run_dense(h->plan0, h->plan0.d_work, st, false))) return rc;
            if (!h->plan0.chunks.empty()) {
                const Chunk &c = h->plan0.chunks.back();
                h->res0_off = c.work_off; h->res0_cnt = c.count;
            }
        } else if ((rc = round0_windowed())) {
            return rc;
        }
        if ((rc = final_tie_pass())) return rc;
        debug_replays();
        debug_levels();
        lapx("final tie pass done");
        phase(3);
        if ((rc = deferred_edit_distances())) return rc;
        return finish();
    }
};

}  // namespace

extern "C" {

int vpr_execute(vpr_handle *h) {
    if (!h) return VPR_ERR_ARG;
    if (!h->uploaded) return fail(h, VPR_ERR_STATE, "vpr_execute before vpr_upload");
    for (auto &b : h->exec_blks) b.used = false;
    for (auto &b : h->exec_pins) b.used = false;
    h->hs = vpr_handle::HostStat();
    Exec x(h);
    const int rc = x.run();
    if (x.t0) (void)hipEventDestroy(x.t0);
    if (x.t1) (void)hipEventDestroy(x.t1);
    if (rc == VPR_OK) h->executed = true;
    return rc;
}

int32_t vpr_test_pool_workers(int32_t cpu_quota) { return int32_t(ParPool::pool_workers(unsigned(std::max(cpu_quota, 1)))); }

}  // extern "C"
