// wfa_cluster.hip -- biWFA dependency clustering on the GPU (include/vcfdist_cluster.h: vcl_wfa_cluster).
//
// wf_swg_cluster (src/cluster.cpp:954-1263) grows every cluster of one haplotype until its left / right "reach" -- how
// far an alignment no worse than the cluster's own alignment score can wander along the reference -- stops touching its
// neighbours.  The alignments are independent per cluster, so one wavefront takes one (cluster, side) job:
//   k_wfa_align   score of query(cluster applied to ref) vs ref        (wf_swg_align,      src/dist.cpp:1510-1652)
//   k_wfa_reach   farthest ref index reachable within that score       (wf_swg_max_reach,  src/dist.cpp:2150-2333)
// with the 64 lanes striding over the diagonals of a wavefront row; rows of previous scores live in a ring in a
// per-job scratch region whose exact size the host knows before the launch (string lengths are arithmetic).  The
// reference's "iterative doubling" of the reference window becomes rounds of launches: a job whose reach hits the
// window edge is re-issued with a window twice as large.  The merge passes between iterations are the same cheap
// sequential sweeps as in the distance clustering and stay on the host.  No CPU fallback for the alignments.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/vcfdist_cluster.h"

namespace {

enum { M_SUB = 0, M_INS = 1, M_DEL = 2, MATS = 3 };

struct DevHap {
    int32_t n;
    const int32_t *pos;
    const uint8_t *type;
    const int32_t *ref_len, *alt_len;
    const int64_t *alt_off;
    const uint8_t *pool;
    const uint8_t *ctg;
    int32_t ctg_len;
};

struct WfaJob {
    int32_t v_beg, v_end;          // variants of the cluster
    int32_t q_beg, q_end;          // generate_str region [q_beg, q_end)
    int32_t r_beg, r_len;          // reference substring
    int32_t q_len;                 // length of the generated query
    int32_t main_diag, main_diag_start, max_score;
    int32_t reverse;
    int32_t out;                   // result slot
    int64_t scratch;               // int offset of the job's scratch (strings first, then the offs ring)
};

extern __shared__ int32_t wfa_lds[];   // LDS variant: the job's strings and ring (one wave per workgroup)

// A job's scratch: its two strings and the ring of wavefront rows.  LDS = true keeps it in LDS (jobs that fit:
// almost all of them), LDS = false in a global region read and written with cache-bypassing (volatile) accesses.
// A single wave owns the region, so ordering is program order plus this fence.
template <bool LDS>
struct Scratch {
    volatile int32_t *g;      // global base (LDS = false)
    int str_words;            // words taken by the two strings
    int q_words;
    __device__ __forceinline__ int ld(int i) const { if (LDS) return wfa_lds[str_words + i]; return g[str_words + i]; }
    __device__ __forceinline__ void st(int i, int v) const { if (LDS) wfa_lds[str_words + i] = v; else g[str_words + i] = v; }
    __device__ __forceinline__ uint8_t q(int i) const {
        if (LDS) return reinterpret_cast<const uint8_t *>(wfa_lds)[i];
        return reinterpret_cast<volatile const uint8_t *>(g)[i];
    }
    __device__ __forceinline__ uint8_t t(int i) const {
        if (LDS) return reinterpret_cast<const uint8_t *>(wfa_lds)[q_words * 4 + i];
        return reinterpret_cast<volatile const uint8_t *>(g)[q_words * 4 + i];
    }
    __device__ __forceinline__ void setq(int i, uint8_t v) const {
        if (LDS) reinterpret_cast<uint8_t *>(wfa_lds)[i] = v; else reinterpret_cast<volatile uint8_t *>(g)[i] = v;
    }
    __device__ __forceinline__ void sett(int i, uint8_t v) const {
        if (LDS) reinterpret_cast<uint8_t *>(wfa_lds)[q_words * 4 + i] = v; else reinterpret_cast<volatile uint8_t *>(g)[q_words * 4 + i] = v;
    }
    __device__ __forceinline__ void sync() const {
        if (LDS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        else { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_s_waitcnt(0); }
    }
};

// generate_str (dist.cpp:81-136) into dst; returns the length.  One lane writes a segment element each.
template <bool LDS>
__device__ int gen_str(const DevHap &H, int beg_idx, int end_idx, int beg_pos, int end_pos, const Scratch<LDS> &S, int lane) {
    int var_idx = beg_idx, len = 0;
    while (var_idx < H.n && H.pos[var_idx] < beg_pos) var_idx++;
    for (int ref_pos = beg_pos; ref_pos < end_pos;) {
        if (var_idx < end_idx && ref_pos == H.pos[var_idx]) {
            const int t = H.type[var_idx];
            if (t == 2 || t == 1) {   // INS / SUB: the ALT allele
                const int al = H.alt_len[var_idx];
                const uint8_t *src = H.pool + H.alt_off[var_idx];
                for (int k = lane; k < al; k += 64) S.setq(len + k, src[k]);
                len += al;
                if (t == 1) ref_pos++;
            } else if (t == 3) {
                ref_pos += H.ref_len[var_idx];
            }
            var_idx++;
        } else {
            const int ref_end = (var_idx < end_idx) ? min(end_pos, H.pos[var_idx]) : end_pos;
            if (ref_end < ref_pos) break;   // overlapping variants: the reference ERRORs
            const int n = ref_end - ref_pos;
            for (int k = lane; k < n; k += 64) S.setq(len + k, H.ctg[ref_pos + k]);
            len += n;
            ref_pos = ref_end;
        }
    }
    return len;
}

// strings of a job: query (generated) and truth (reference substring), optionally reversed
template <bool LDS>
__device__ void load_strings(const DevHap &H, const WfaJob &J, const Scratch<LDS> &S, int lane) {
    gen_str<LDS>(H, J.v_beg, J.v_end, J.q_beg, J.q_end, S, lane);
    for (int k = lane; k < J.r_len; k += 64) S.sett(k, H.ctg[J.r_beg + k]);
    S.sync();
    if (J.reverse) {   // std::reverse of both strings
        for (int k = lane; k < J.q_len / 2; k += 64) { const uint8_t a = S.q(k), b = S.q(J.q_len - 1 - k); S.setq(k, b); S.setq(J.q_len - 1 - k, a); }
        for (int k = lane; k < J.r_len / 2; k += 64) { const uint8_t a = S.t(k), b = S.t(J.r_len - 1 - k); S.sett(k, b); S.sett(J.r_len - 1 - k, a); }
        S.sync();
    }
}

// wf_swg_align, dist.cpp:1510-1652: only the score is needed; rows older than max(x, o+e) are never read, so a ring
// of `scores` freshly initialised rows replaces the reference's ever-growing vectors.
template <bool LDS>
__global__ void __launch_bounds__(64) k_wfa_align(DevHap H, const WfaJob *__restrict__ jobs, int n_jobs,
                                                  int32_t *__restrict__ scratch_all, int32_t *__restrict__ out,
                                                  int x, int o, int e) {
    if (int(blockIdx.x) >= n_jobs) return;
    const WfaJob J = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    const int query_len = J.q_len, truth_len = J.r_len, mat_len = query_len + truth_len - 1;
    const int scores = max(x, o + e) + 1;
    const int y = mat_len, z = y * scores;
    Scratch<LDS> S;
    S.g = scratch_all + J.scratch;
    S.q_words = (query_len + 3) >> 2;
    S.str_words = S.q_words + ((truth_len + 3) >> 2);
    load_strings<LDS>(H, J, S, lane);
    for (int k = lane; k < MATS * z; k += 64) S.st(k, -2);
    S.sync();
    int s = 0, s2 = 0;
    if (lane == 0) S.st(M_SUB * z + query_len - 1, -1);
    S.sync();
    while (true) {
        for (int m = M_INS; m < MATS; m++)
            for (int d = lane; d < mat_len; d += 64) {
                const int off = S.ld(m * z + s2 * y + d), diag = d + 1 - query_len;
                if (off >= 0 && off < query_len && diag + off >= 0 && diag + off < truth_len && off >= S.ld(M_SUB * z + s2 * y + d))
                    S.st(M_SUB * z + s2 * y + d, off);
            }
        S.sync();
        bool done = false;
        for (int d0 = 0; d0 < mat_len && !done; d0 += 64) {
            const int d = d0 + lane;
            bool fin = false;
            if (d < mat_len) {
                int off = S.ld(M_SUB * z + s2 * y + d);
                const int diag = d + 1 - query_len;
                while (off != -2 && diag + off >= -1 && off < query_len - 1 && diag + off < truth_len - 1) {
                    if (S.q(off + 1) == S.t(diag + off + 1)) off++;
                    else break;
                }
                S.st(M_SUB * z + s2 * y + d, off);
                fin = (off == query_len - 1 && off + diag == truth_len - 1);
            }
            done = __any(fin);
        }
        if (done) break;
        S.sync();
        s++; s2++;
        if (s2 == scores) s2 = 0;
        for (int m = 0; m < MATS; m++)
            for (int d = lane; d < mat_len; d += 64) S.st(m * z + s2 * y + d, -2);
        S.sync();
        auto row = [&](int back) { int r = s2 - back; if (r < 0) r += scores; return r; };
        for (int d = lane; d < mat_len; d += 64) {
            const int diag = d + 1 - query_len;
            int vsub = -2, vdel = -2, vins = -2;
            if (s - x >= 0) {
                const int p = S.ld(M_SUB * z + row(x) * y + d);
                if (p != -2 && p + 1 < query_len && diag + p + 1 < truth_len && p + 1 >= vsub) vsub = p + 1;
            }
            if (s - (o + e) >= 0 && d > 0) {
                const int p = S.ld(M_SUB * z + row(o + e) * y + d - 1);
                if (p != -2 && diag + p < truth_len && p >= vdel) vdel = p;
            }
            if (s - (o + e) >= 0 && d < mat_len - 1) {
                const int p = S.ld(M_SUB * z + row(o + e) * y + d + 1);
                if (p != -2 && p + 1 < query_len && diag + p + 1 < truth_len && diag + p + 1 >= 0 && p + 1 >= vins) vins = p + 1;
            }
            if (s - e >= 0 && d > 0) {
                const int p = S.ld(M_DEL * z + row(e) * y + d - 1);
                if (p != -2 && diag + p < truth_len && p >= vdel) vdel = p;
            }
            if (s - e >= 0 && d < mat_len - 1) {
                const int p = S.ld(M_INS * z + row(e) * y + d + 1);
                if (p != -2 && p + 1 < query_len && diag + p + 1 < truth_len && diag + p + 1 >= 0 && p + 1 >= vins) vins = p + 1;
            }
            S.st(M_SUB * z + s2 * y + d, vsub);
            S.st(M_DEL * z + s2 * y + d, vdel);
            S.st(M_INS * z + s2 * y + d, vins);
        }
        S.sync();
    }
    if (lane == 0) out[J.out] = s;
}

// wf_swg_max_reach, dist.cpp:2150-2333.  The ring keeps the reference's exact contents: the SUB row of a reused ring
// slot is NOT reset (only INS / DEL are, :2215-2219), and the final maximum runs over every slot of the ring.
template <bool LDS>
__global__ void __launch_bounds__(64) k_wfa_reach(DevHap H, const WfaJob *__restrict__ jobs, int n_jobs,
                                                  int32_t *__restrict__ scratch_all, int32_t *__restrict__ out,
                                                  int x, int o, int e) {
    if (int(blockIdx.x) >= n_jobs) return;
    const WfaJob J = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    const int query_len = J.q_len, truth_len = J.r_len, mat_len = query_len + truth_len - 1;
    const int main_diag = J.main_diag, main_diag_off = J.main_diag_start - J.main_diag, max_score = J.max_score;
    const bool reverse = J.reverse != 0;
    const int scores = max(x, o + e) + 1;
    const int y = mat_len, z = y * scores;
    Scratch<LDS> S;
    S.g = scratch_all + J.scratch;
    S.q_words = (query_len + 3) >> 2;
    S.str_words = S.q_words + ((truth_len + 3) >> 2);
    load_strings<LDS>(H, J, S, lane);
    for (int k = lane; k < MATS * z; k += 64) S.st(k, -2);
    S.sync();
    int s = 0, s2 = 0;
    if (lane == 0) S.st(M_SUB * z + query_len - 1, -1);
    S.sync();
    int result = INT_MIN;
    while (true) {
        if (!reverse) {
            for (int m = M_INS; m < MATS; m++)
                for (int d = lane; d < mat_len; d += 64) {
                    const int off = S.ld(m * z + s2 * y + d), diag = d + 1 - query_len;
                    if (off >= 0 && off < query_len && diag + off >= 0 && diag + off < truth_len && off >= S.ld(M_SUB * z + s2 * y + d))
                        S.st(M_SUB * z + s2 * y + d, off);
                }
            S.sync();
        }
        // extension; the reference returns at the first diagonal (ascending d) that reaches the last column of the
        // truth or the end of the query
        for (int d0 = 0; d0 < mat_len && result == INT_MIN; d0 += 64) {
            const int d = d0 + lane;
            int hit = INT_MIN;
            if (d < mat_len) {
                int off = S.ld(M_SUB * z + s2 * y + d);
                const int diag = d + 1 - query_len;
                while ((diag != main_diag || off + 1 < main_diag_off) && off != -2 && diag + off >= -1 &&
                       off < query_len - 1 && diag + off < truth_len - 1) {
                    if (S.q(off + 1) == S.t(diag + off + 1)) off++;
                    else break;
                }
                S.st(M_SUB * z + s2 * y + d, off);
                if (off + diag == truth_len - 1) hit = truth_len - 1;
                else if (off == query_len - 1 && off + diag >= 0 && off + diag < truth_len - 1) hit = off + diag;
            }
            const unsigned long long mh = __ballot(hit != INT_MIN);
            if (mh) result = __shfl(hit, __builtin_ctzll(mh));
        }
        if (result != INT_MIN) break;
        if (s == max_score) break;
        S.sync();
        s++; s2++;
        if (s2 == scores) s2 = 0;
        for (int m = M_INS; m < MATS; m++)
            for (int d = lane; d < mat_len; d += 64) S.st(m * z + s2 * y + d, -2);
        S.sync();
        auto row = [&](int back) { int r = s2 - back; if (r < 0) r += scores; return r; };
        for (int d = lane; d < mat_len; d += 64) {
            const int diag = d + 1 - query_len;
            int vsub = S.ld(M_SUB * z + s2 * y + d);      // stale value of the reused ring slot
            int vdel = -2, vins = -2;
            if (s - x >= 0) {
                const int p = S.ld(M_SUB * z + row(x) * y + d);
                if (p != -2 && p + 1 < query_len && diag + p + 1 < truth_len && p + 1 >= vsub) vsub = p + 1;
            }
            {
                const int back = reverse ? e : (o + e);
                if (s - back >= 0 && d > 0) {
                    const int p = S.ld(M_SUB * z + row(back) * y + d - 1);
                    if (p != -2 && diag + p < truth_len && p >= vdel) vdel = p;
                }
                if (s - back >= 0 && d < mat_len - 1) {
                    const int p = S.ld(M_SUB * z + row(back) * y + d + 1);
                    if (p != -2 && p + 1 < query_len && diag + p + 1 < truth_len && diag + p + 1 >= 0 && p + 1 >= vins) vins = p + 1;
                }
            }
            if (reverse && s - o >= 0) {
                for (int m = M_INS; m < MATS; m++) {
                    const int p = S.ld(m * z + row(o) * y + d);
                    if (p >= 0 && p < query_len && diag + p >= 0 && diag + p < truth_len && p > vsub) vsub = p;
                }
            }
            if (s - e >= 0 && d > 0) {
                const int p = S.ld(M_DEL * z + row(e) * y + d - 1);
                if (p != -2 && diag + p < truth_len && p >= vdel) vdel = p;
            }
            if (s - e >= 0 && d < mat_len - 1) {
                const int p = S.ld(M_INS * z + row(e) * y + d + 1);
                if (p != -2 && p + 1 < query_len && diag + p + 1 < truth_len && diag + p + 1 >= 0 && p + 1 >= vins) vins = p + 1;
            }
            S.st(M_SUB * z + s2 * y + d, vsub);
            S.st(M_DEL * z + s2 * y + d, vdel);
            S.st(M_INS * z + s2 * y + d, vins);
        }
        S.sync();
    }
    if (result == INT_MIN) {   // max reach over every slot of the ring, dist.cpp:2318-2331
        S.sync();
        int mr = 0;
        for (int k = lane; k < MATS * z; k += 64) {
            const int off = S.ld(k);
            const int d = k % y, diag = d + 1 - query_len;
            if (off >= 0 && off < query_len && diag + off >= 0 && diag + off < truth_len) mr = max(mr, diag + off);
        }
        for (int sft = 32; sft > 0; sft >>= 1) mr = max(mr, __shfl_xor(mr, sft));
        result = mr;
    }
    if (lane == 0) out[J.out] = result;
}

// host mirror of generate_str's length arithmetic
int gen_len(const vcl_hap &h, int beg_idx, int end_idx, int beg_pos, int end_pos) {
    int var_idx = beg_idx, len = 0;
    while (var_idx < h.n_var && h.pos[var_idx] < beg_pos) var_idx++;
    for (int ref_pos = beg_pos; ref_pos < end_pos;) {
        if (var_idx < end_idx && ref_pos == h.pos[var_idx]) {
            const int t = h.type[var_idx];
            if (t == 2) len += h.alt_len[var_idx];
            else if (t == 3) ref_pos += h.ref_len[var_idx];
            else if (t == 1) { len += h.alt_len[var_idx]; ref_pos++; }
            var_idx++;
        } else {
            const int ref_end = (var_idx < end_idx) ? std::min(end_pos, h.pos[var_idx]) : end_pos;
            if (ref_end < ref_pos) break;
            len += ref_end - ref_pos;
            ref_pos = ref_end;
        }
    }
    return len;
}

template <typename T>
T *dev_copy(const T *src, size_t n, std::vector<void *> &allocs) {
    void *p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
    allocs.push_back(p);
    if (n && hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return static_cast<T *>(p);
}

}  // namespace

extern "C" int vcl_wfa_cluster(const vcl_hap_seq *hs, const uint8_t *ctg_seq, int32_t ctg_len, int32_t sub, int32_t open,
                               int32_t extend, int32_t max_cluster_itrs, int32_t reach_min_gap, int32_t device,
                               vcl_clusters **out, vcl_wfa_stats *stats) {
    if (!hs || !out || !ctg_seq || ctg_len <= 0 || sub <= 0 || open < 0 || extend <= 0) return VCL_ERR_ARG;
    *out = nullptr;
    const vcl_hap &h = hs->cols;
    const int n = h.n_var;
    for (int v = 0; v < n; v++) {
        if (h.type[v] < 1 || h.type[v] > 3) return VCL_ERR_TYPE;
        if (v && h.pos[v] < h.pos[v - 1]) return VCL_ERR_ARG;
    }
    vcl_wfa_stats st;
    memset(&st, 0, sizeof(st));
    std::vector<int> left_reach, right_reach, prev_clusters;
    std::vector<void *> allocs;
    hipEvent_t ev_done[2] = {nullptr, nullptr};      // destroyed by cleanup() on every return path
    auto cleanup = [&] { for (void *p : allocs) (void)hipFree(p); for (hipEvent_t e : ev_done) if (e) (void)hipEventDestroy(e); };
    if (n) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess)
            return VCL_ERR_DEVICE;   // no CPU fallback
        int64_t pool_len = 0;
        for (int v = 0; v < n; v++) pool_len = std::max<int64_t>(pool_len, hs->alt_off[v] + h.alt_len[v]);
        DevHap H;
        H.n = n; H.ctg_len = ctg_len;
        H.pos = dev_copy(h.pos, size_t(n), allocs);
        H.type = dev_copy(h.type, size_t(n), allocs);
        H.ref_len = dev_copy(h.ref_len, size_t(n), allocs);
        H.alt_len = dev_copy(h.alt_len, size_t(n), allocs);
        H.alt_off = dev_copy(hs->alt_off, size_t(n), allocs);
        H.pool = dev_copy(hs->pool, size_t(pool_len), allocs);
        H.ctg = dev_copy(ctg_seq, size_t(ctg_len), allocs);
        if (!H.pos || !H.type || !H.ref_len || !H.alt_len || !H.alt_off || !H.pool || !H.ctg) { cleanup(); return VCL_ERR_DEVICE; }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            if (e0) (void)hipEventDestroy(e0);
            cleanup();
            return VCL_ERR_DEVICE;
        }
        ev_done[0] = e0; ev_done[1] = e1;
        const int scores = std::max(sub, open + extend) + 1;
        // one launch of `kern` over `jobs` (scratch offsets assigned here); results into res[job.out]
        int32_t *d_scratch = nullptr; size_t scratch_cap = 0;
        WfaJob *d_jobs = nullptr; size_t jobs_cap = 0;
        int32_t *d_res = nullptr; size_t res_cap = 0;
        auto run = [&](std::vector<WfaJob> &jobs, bool align, std::vector<int32_t> &res) -> bool {
            if (jobs.empty()) return true;
            int64_t total = 0;
            for (WfaJob &j : jobs) {
                j.scratch = total;
                const int64_t mat_len = int64_t(j.q_len) + j.r_len - 1;
                total += ((j.q_len + 3) >> 2) + ((j.r_len + 3) >> 2) + MATS * scores * mat_len + 16;
            }
            if (size_t(total) > scratch_cap) {
                auto drop = [&](void *q) { if (q) { allocs.erase(std::remove(allocs.begin(), allocs.end(), q), allocs.end()); (void)hipFree(q); } };
                (void)hipDeviceSynchronize();     // (an outgrown buffer is released at once, not at the end of the call)
                drop(d_scratch); d_scratch = nullptr;
                if (hipMalloc(reinterpret_cast<void **>(&d_scratch), size_t(total) * 4 * 3 / 2) != hipSuccess) return false;
                allocs.push_back(d_scratch);
                scratch_cap = size_t(total) * 3 / 2;
            }
            if (jobs.size() > jobs_cap) {
                if (hipMalloc(reinterpret_cast<void **>(&d_jobs), jobs.size() * 2 * sizeof(WfaJob)) != hipSuccess) return false;
                allocs.push_back(d_jobs);
                jobs_cap = jobs.size() * 2;
            }
            if (res.size() > res_cap) {
                if (hipMalloc(reinterpret_cast<void **>(&d_res), res.size() * 2 * 4) != hipSuccess) return false;
                allocs.push_back(d_res);
                res_cap = res.size() * 2;
            }
            // jobs whose strings + ring fit in LDS run the LDS variant, bucketed by size (the dynamic LDS size of a
            // launch bounds the waves per CU); the rest uses the global scratch
            const int64_t LIM[4] = {8 << 10, 16 << 10, 32 << 10, 64 << 10};
            auto need_bytes = [&](const WfaJob &j) {
                return 4 * (int64_t((j.q_len + 3) >> 2) + ((j.r_len + 3) >> 2) + MATS * scores * (int64_t(j.q_len) + j.r_len - 1));
            };
            auto bucket_of = [&](const WfaJob &j) { const int64_t nb = need_bytes(j); for (int b = 0; b < 4; b++) if (nb <= LIM[b]) return b; return 4; };
            std::stable_sort(jobs.begin(), jobs.end(), [&](const WfaJob &a, const WfaJob &b) { return bucket_of(a) < bucket_of(b); });
            if (hipMemcpy(d_jobs, jobs.data(), jobs.size() * sizeof(WfaJob), hipMemcpyHostToDevice) != hipSuccess) return false;
            (void)hipEventRecord(e0, nullptr);
            for (size_t lo = 0; lo < jobs.size();) {
                const int b = bucket_of(jobs[lo]);
                size_t hi = lo;
                while (hi < jobs.size() && bucket_of(jobs[hi]) == b) hi++;
                const unsigned cnt = unsigned(hi - lo);
                const size_t shm = b < 4 ? size_t(LIM[b]) : 0;
                if (align) {
                    if (b < 4) hipLaunchKernelGGL(k_wfa_align<true>, dim3(cnt), dim3(64), shm, nullptr, H, d_jobs + lo, int(cnt), d_scratch, d_res, sub, open, extend);
                    else hipLaunchKernelGGL(k_wfa_align<false>, dim3(cnt), dim3(64), 0, nullptr, H, d_jobs + lo, int(cnt), d_scratch, d_res, sub, open, extend);
                } else {
                    if (b < 4) hipLaunchKernelGGL(k_wfa_reach<true>, dim3(cnt), dim3(64), shm, nullptr, H, d_jobs + lo, int(cnt), d_scratch, d_res, sub, open, extend);
                    else hipLaunchKernelGGL(k_wfa_reach<false>, dim3(cnt), dim3(64), 0, nullptr, H, d_jobs + lo, int(cnt), d_scratch, d_res, sub, open, extend);
                }
                lo = hi;
            }
            (void)hipEventRecord(e1, nullptr);
            if (hipMemcpy(res.data(), d_res, res.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            st.ms_device += ms;
            return hipGetLastError() == hipSuccess;
        };

        prev_clusters.resize(size_t(n) + 1);
        for (int i = 0; i <= n; i++) prev_clusters[size_t(i)] = i;
        std::vector<char> prev_active(size_t(n) + 1, 1);
        left_reach.assign(size_t(n) + 1, 0);
        right_reach.assign(size_t(n) + 1, 0);
        int iter = 0;
        while (std::find(prev_active.begin(), prev_active.end(), 1) != prev_active.end()) {
            iter++;
            if (iter > max_cluster_itrs) break;
            st.iterations = iter;
            const std::vector<int> &cl = prev_clusters;
            const size_t nc = cl.size();
            left_reach[nc - 1] = INT_MAX;
            right_reach[nc - 1] = INT_MAX;
            std::vector<int32_t> act;
            for (size_t c = 0; c + 1 < nc; c++)
                if (prev_active[c]) act.push_back(int32_t(c));
            // ---- scores of the active clusters (cluster.cpp:1035-1045)
            std::vector<WfaJob> jobs;
            std::vector<int32_t> score(act.size(), 0);
            for (size_t k = 0; k < act.size(); k++) {
                const int c = act[k], bi = cl[size_t(c)], ei = cl[size_t(c) + 1];
                WfaJob j;
                memset(&j, 0, sizeof(j));
                j.v_beg = bi; j.v_end = ei;
                j.q_beg = std::max(0, h.pos[bi] - 1);
                j.q_end = std::min(ctg_len, h.pos[ei - 1] + h.rlen[ei - 1] + 1);
                j.r_beg = j.q_beg; j.r_len = j.q_end - j.q_beg;
                j.q_len = gen_len(h, bi, ei, j.q_beg, j.q_end);
                j.out = int32_t(k);
                jobs.push_back(j);
            }
            if (!run(jobs, true, score)) { cleanup(); return VCL_ERR_DEVICE; }
            st.align_calls += int64_t(jobs.size());
            // ---- reaches with iterative doubling (cluster.cpp:1048-1161): a round per window size
            struct Side { int c, k, reverse, beg_pos, end_pos, main_diag, main_diag_start, ref_len, reach; bool done; };
            std::vector<Side> sides;
            for (size_t k = 0; k < act.size(); k++) {
                const int c = act[k], bi = cl[size_t(c)], ei = cl[size_t(c) + 1];
                const int beg_pos = h.pos[bi] - 1, end_pos = h.pos[ei - 1] + h.rlen[ei - 1] + 1;
                int main_diag = 0;
                for (int v = bi; v < ei; v++) main_diag += h.ref_len[v] - h.alt_len[v];
                sides.push_back(Side{c, int(k), 1, beg_pos, end_pos, main_diag, end_pos - h.pos[bi], end_pos - beg_pos, 0, false});
                sides.push_back(Side{c, int(k), 0, beg_pos, end_pos, main_diag, h.pos[ei - 1] + h.rlen[ei - 1] - beg_pos,
                                     end_pos - beg_pos, 0, false});
            }
            for (size_t pending = sides.size(); pending > 0;) {
                jobs.clear();
                std::vector<size_t> idx;
                for (size_t q = 0; q < sides.size(); q++) {
                    Side &S = sides[q];
                    if (S.done) continue;
                    const int bi = cl[size_t(S.c)], ei = cl[size_t(S.c) + 1], sc = score[size_t(S.k)];
                    S.ref_len *= 2;
                    WfaJob j;
                    memset(&j, 0, sizeof(j));
                    j.v_beg = bi; j.v_end = ei;
                    if (S.reverse) {
                        S.beg_pos = std::max(0, S.end_pos - S.ref_len - std::abs(S.main_diag) - sc / extend - 3);
                        j.r_beg = std::max(0, S.end_pos - S.ref_len);
                        j.r_len = std::min(S.ref_len, ctg_len - j.r_beg);                      // std::string::substr clamp
                    } else {
                        S.end_pos = std::min(ctg_len, S.beg_pos + S.ref_len + std::abs(S.main_diag) + sc / extend + 3);
                        j.r_beg = S.beg_pos;
                        j.r_len = std::min(std::min(S.ref_len, S.end_pos - S.beg_pos), ctg_len - j.r_beg);
                    }
                    j.q_beg = S.beg_pos; j.q_end = S.end_pos;
                    j.q_len = gen_len(h, bi, ei, j.q_beg, j.q_end);
                    j.main_diag = S.main_diag; j.main_diag_start = S.main_diag_start; j.max_score = sc;
                    j.reverse = S.reverse;
                    j.out = int32_t(jobs.size());
                    jobs.push_back(j);
                    idx.push_back(q);
                }
                std::vector<int32_t> reach(jobs.size(), 0);
                if (!run(jobs, false, reach)) { cleanup(); return VCL_ERR_DEVICE; }
                st.reach_calls += int64_t(jobs.size());
                for (size_t w = 0; w < idx.size(); w++) {
                    Side &S = sides[idx[w]];
                    S.reach = reach[w];
                    const bool edge = S.reverse ? (S.beg_pos == 0) : (S.end_pos == ctg_len);
                    if (S.reach != S.ref_len - 1 || edge) { S.done = true; pending--; }
                }
            }
            for (const Side &S : sides) {
                if (S.reverse) left_reach[size_t(S.c)] = S.end_pos - S.reach;
                else right_reach[size_t(S.c)] = S.beg_pos + S.reach + 1;
            }
            // ---- merge passes (cluster.cpp:1173-1232), host
            struct Group { int first, lo, hi; bool active; };
            std::vector<Group> g1, g2;
            for (size_t c = 0; c < nc;) {
                Group g{cl[c], left_reach[c], right_reach[c], false};
                size_t w = c + 1;
                while (w < nc && g.hi + reach_min_gap >= left_reach[w]) {
                    g.hi = std::max(g.hi, right_reach[w]);
                    g.lo = std::min(g.lo, left_reach[w]);
                    w++;
                }
                g.active = w - c > 1;
                g1.push_back(g);
                c = w;
            }
            for (int64_t k = int64_t(g1.size()) - 1; k >= 0;) {
                Group g = g1[size_t(k)];
                while (k > 0 && g.lo <= g1[size_t(k - 1)].hi + reach_min_gap) {
                    k--;
                    g.lo = std::min(g.lo, g1[size_t(k)].lo);
                    g.hi = std::max(g.hi, g1[size_t(k)].hi);
                    g.first = g1[size_t(k)].first;
                    g.active = true;
                }
                g2.push_back(g);
                k--;
            }
            prev_clusters.clear(); prev_active.clear(); left_reach.clear(); right_reach.clear();
            for (auto it = g2.rbegin(); it != g2.rend(); ++it) {
                prev_clusters.push_back(it->first);
                prev_active.push_back(it->active ? 1 : 0);
                left_reach.push_back(it->lo);
                right_reach.push_back(it->hi);
            }
        }
        cleanup();
    }
    vcl_clusters *c = static_cast<vcl_clusters *>(calloc(1, sizeof(vcl_clusters)));
    if (!c) return VCL_ERR_ARG;
    const size_t m = prev_clusters.size();
    c->n = m ? int32_t(m) - 1 : 0;
    c->var_beg = static_cast<int32_t *>(malloc((m + 1) * 4));
    c->left_reach = static_cast<int32_t *>(malloc((m + 1) * 4));
    c->right_reach = static_cast<int32_t *>(malloc((m + 1) * 4));
    for (size_t k = 0; k < m; k++) {
        c->var_beg[k] = prev_clusters[k];
        c->left_reach[k] = left_reach[k];
        c->right_reach[k] = right_reach[k];
    }
    *out = c;
    if (stats) *stats = st;
    return VCL_OK;
}
