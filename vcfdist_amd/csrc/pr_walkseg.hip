// pr_walkseg.hip -- the forward walk of get_prec_recall_path_sync (dist.cpp:865-998) over the striped 64-cell layout, in
// parallel over SEGMENTS of truth rows.
//
// A walk is a chain of dependent steps: one wavefront needs ~0.6 us per truth row (k_walk_rows), 6 ms for the 9 288-row
// alignment that bounds a whole-genome batch, while the rest of the device idles.  But where the walk goes from a cell
// depends only on the path_ptr bytes, not on how it got there.  So:
//   1. k_wseg_map: for every segment of WSEG_S stripes (128 truth rows) and EVERY cell of the segment's first row that the
//      walk could enter (2 planes x 64 window columns, one thread each), follow the walk to the segment's end: the cell of
//      the next segment's first row it enters, the move that enters it, and the number of path entries on the way.
//   2. k_wseg_compose: one wavefront per alignment chains the segments' maps from the start cell: entry state and path
//      index of every segment (one table lookup per 128 rows).
//   3. k_wseg_emit: every segment is walked again from its now known entry by the row-sweep code of k_walk_rows
//      (walk_rows_range, pr_band.hip), which writes its path entries at their final place.
// Same path, entry for entry, as the sequential walk; the 9 288-row alignment takes ~0.2 ms instead of 6.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcfdist_pr.h"
#include "pr_device.h"

#define WSEG_S 16                       // stripes per segment
#define WSEG_ROWS (WSEG_S * FS_K)       // truth rows per segment
#define WSEG_COLS 256                   // columns of each plane whose pointers are staged per segment
#define WSEG_DEAD 0xffffffffu           // the walk leaves the window / finds no pointer from this entry
#define WSEG_END 0xfffffffeu            // the walk reaches the alignment's end cell inside the segment

struct WsegTables {
    int32_t *counter;       // segment slots handed out
    int32_t *seg_base;      // [launch position] first slot of the alignment, -1: not walked by this launch
    int2 *owner;            // [slot] {launch position, segment}
    uint2 *map;             // [slot][128]  x: exit cell (column | plane << 31) or WSEG_DEAD / WSEG_END; y: entries | move << 24
    int4 *entry;            // [slot] {entry cell, entering move, path index, valid}
    int32_t cap;
};

__device__ __forceinline__ bool wseg_active(const AlnDesc &d, const AlnOut &O, int tag) { return O.band_ok == tag && d.band_pad == tag; }

__global__ void k_wseg_plan(const AlnDesc *__restrict__ descs, const int32_t *__restrict__ work, int n_work,
                            const AlnOut *__restrict__ outs, int tag, WsegTables T) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_work) return;
    const int a = work[k];
    int base = -1;
    if (a >= 0) {
        const AlnDesc &d = descs[a];
        if (wseg_active(d, outs[a], tag)) {
            const int n_stripes = (d.Lt + FS_K - 1) / FS_K;
            const int n_seg = (n_stripes + WSEG_S - 1) / WSEG_S;
            base = atomicAdd(T.counter, n_seg);
            if (base + n_seg > T.cap) base = -1;      // (cannot happen: the host sized the tables by the rows of the launch)
            else for (int j = 0; j < n_seg; j++) T.owner[base + j] = make_int2(k, j);
        }
    }
    T.seg_base[k] = base;
}

__global__ void __launch_bounds__(128) k_wseg_map(DevBatch B, const AlnDesc *__restrict__ descs, const int32_t *__restrict__ work,
                                                  const uint8_t *__restrict__ ws, const int32_t *__restrict__ blo_all, WsegTables T) {
    const int slot = blockIdx.x;
    if (slot >= *T.counter) return;
    const int2 own = T.owner[slot];
    const int a = work[own.x];
    const AlnDesc d = descs[a];
    const int tid = threadIdx.x;
    const int Lq = d.Lq, Lr = d.Lr, Lt = d.Lt;
    const int Lp[2] = {Lq, Lr};
    const int pitch[2] = {d.pitch[0], d.pitch[1]};
    const int s0 = own.y * WSEG_S, t0 = s0 * FS_K;
    const int nr = min(WSEG_ROWS, Lt - t0);
    const int32_t *blo = blo_all + d.blo_off;
    const int2 *wq_ = B.wk_q[d.qs] + d.q_off, *wr_ = B.wk_r[d.qs] + d.r_off;
    __shared__ __align__(16) uint8_t rows[2][WSEG_ROWS * FS_W];
    __shared__ int32_t cols[2][WSEG_COLS];
    __shared__ int32_t los[2][WSEG_S];
    // ---- stage: stripe origins, the segment's path_ptr rows (contiguous per plane), the pointers of the columns around them
    if (tid < 2 * WSEG_S) {
        const int p = tid / WSEG_S, j = tid % WSEG_S;
        const int t = min(t0 + j * FS_K, Lt - 1);
        los[p][j] = blo[p * Lt + t];
    }
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const uint8_t *src = ws + d.mat_off[p] + size_t(t0) * pitch[p];
        const int nbytes = nr * pitch[p];
        for (int k = tid * 16; k < nbytes; k += 128 * 16)
            *reinterpret_cast<uint4 *>(&rows[p][k]) = *reinterpret_cast<const uint4 *>(src + k);
    }
    const int cb[2] = {blo[t0], blo[Lt + t0]};
    for (int k = tid; k < 2 * WSEG_COLS; k += 128) {
        const int p = k / WSEG_COLS, x = cb[p] + k % WSEG_COLS;
        cols[p][k % WSEG_COLS] = (x < Lp[p]) ? (p == 0 ? wq_[x].x : wr_[x].x) : 0;
    }
    __syncthreads();
    auto other = [&](int p, int x) -> int {      // q2r (p = 0) / r2q (p = 1) of column x
        const int k = x - cb[p];
        if (k >= 0 && k < WSEG_COLS) return cols[p][k];
        return p == 0 ? wq_[x].x : wr_[x].x;
    };
    // ---- every possible entry cell of the first row: plane = wave, window column = lane
    int hi = tid >> 6, e = los[hi][0] + (tid & 63);
    uint32_t out = WSEG_DEAD;
    int count = 0, mv = 0;
    bool alive = true;
    for (int r = 0; r < nr && alive; r++) {
        const int lo_h = los[hi][r >> 3];
        int el = e - lo_h;
        if (el < 0 || el > 63 || e >= Lp[hi]) { alive = false; break; }
        const uint8_t *row = &rows[hi][r * pitch[hi]];
        int pc = el < pitch[hi] ? int(row[el]) : 0;
        // the run of INS-only cells that starts at the entry cell
        while ((pc & F_INS) && !(pc & (F_MAT | F_SUB)) && !(hi == 1 && (pc & F_SWP))) {
            e++; el++; count++;
            if (el > 63 || e >= Lp[hi]) { alive = false; break; }
            pc = el < pitch[hi] ? int(row[el]) : 0;
        }
        if (!alive) break;
        count++;
        if (t0 + r == Lt - 1) {                    // the walk ends at the end cell of its plane
            if (e == Lp[hi] - 1) out = WSEG_END;
            alive = false;
            break;
        }
        const int p = pc & 31;
        if (hi == 1 && (p & F_SWP)) { mv = F_SWP; e = other(1, e) + 1; hi = 0; }
        else if (p & F_MAT) { mv = F_MAT; e = e + 1; }
        else if (p & F_SUB) { mv = F_SUB; e = e + 1; }
        else if (p & F_DEL) { mv = F_DEL; }
        else if (hi == 0 && (p & F_SWP)) { mv = F_SWP; e = other(0, e) + 1; hi = 1; }
        else { alive = false; break; }
    }
    if (alive) out = uint32_t(e) | (uint32_t(hi) << 31);      // entered the next segment's first row
    T.map[size_t(slot) * 128 + tid] = make_uint2(out, uint32_t(count) | (uint32_t(mv) << 24));
}

__global__ void __launch_bounds__(64) k_wseg_compose(const AlnDesc *__restrict__ descs, const int32_t *__restrict__ work, int n_work,
                                                     const int32_t *__restrict__ blo_all, AlnOut *__restrict__ outs, WsegTables T) {
    const int k = blockIdx.x;
    if (k >= n_work) return;
    const int base = T.seg_base[k];
    if (base < 0) return;
    const int a = work[k];
    const AlnDesc &d = descs[a];
    AlnOut &O = outs[a];
    const int Lt = d.Lt;
    const int32_t *blo = blo_all + d.blo_off;
    const int n_stripes = (Lt + FS_K - 1) / FS_K, n_seg = (n_stripes + WSEG_S - 1) / WSEG_S;
    int hi = O.beg_plane, e = 0, mv = 0;
    int64_t n = 0;
    uint32_t status = 0;
    bool done = false;
    int j = 0;
    int bq = 0, br = 0;      // window origins of the first rows of 64 segments (lane l <-> segment c0 + l)
    for (; j < n_seg; j++) {
        if ((j & 63) == 0) {
            const int jj = j + int(threadIdx.x);
            bq = br = 0;
            if (jj < n_seg) { bq = blo[jj * WSEG_ROWS]; br = blo[Lt + jj * WSEG_ROWS]; }
        }
        if (threadIdx.x == 0) T.entry[base + j] = make_int4(int(uint32_t(e) | (uint32_t(hi) << 31)), mv, int(n), 1);
        const int el = e - (hi ? __builtin_amdgcn_readlane(br, j & 63) : __builtin_amdgcn_readlane(bq, j & 63));
        if (el < 0 || el > 63) { status |= VPR_ST_ERR_NO_PTR; break; }
        const uint2 m = T.map[size_t(base + j) * 128 + hi * 64 + el];
        n += int(m.y & 0xffffffu);
        if (m.x == WSEG_DEAD) { status |= VPR_ST_ERR_NO_PTR; break; }      // (the segment's own walk reports where)
        if (n > d.path_cap) { status |= VPR_ST_ERR_LIMIT; break; }
        if (m.x == WSEG_END) { done = true; j++; break; }
        e = int(m.x & 0x7fffffffu); hi = int(m.x >> 31); mv = int(m.y >> 24);
    }
    if (threadIdx.x == 0) {
        for (int q = j + (done || !status ? 0 : 1); q < n_seg; q++) T.entry[base + q] = make_int4(0, 0, 0, 0);   // behind a failure: not walked
        O.path_len = int32_t(min<int64_t>(n, int64_t(d.path_cap)));
        if (!done && !status) status |= VPR_ST_ERR_NO_PTR;
        if (status) { O.n_sec = 0; atomicOr(&O.status, status); }
    }
}

__global__ void __launch_bounds__(64) k_wseg_emit(DevBatch B, const AlnDesc *__restrict__ descs, const int32_t *__restrict__ work,
                                                  const uint8_t *__restrict__ ws, const int32_t *__restrict__ blo_all,
                                                  AlnOut *__restrict__ outs, PathEnt *__restrict__ paths, WsegTables T) {
    const int slot = blockIdx.x;
    if (slot >= *T.counter) return;
    const int4 en = T.entry[slot];
    if (!en.w) return;
    const int2 own = T.owner[slot];
    const int a = work[own.x];
    const AlnDesc d = descs[a];
    const int n_stripes = (d.Lt + FS_K - 1) / FS_K;
    const int s0 = own.y * WSEG_S;
    walk_rows_range(B, d, outs[a], ws, blo_all, paths, s0, min(s0 + WSEG_S, n_stripes), int(uint32_t(en.x) >> 31), en.x & 0x7fffffff, en.y,
                    int64_t(en.z), false);
}
