// pr_fwdpar.hip -- the forward sweep of the striped 64-cell level (calc_prec_recall_aln, dist.cpp:251-443) in parallel over
// BLOCKS of truth rows.
//
// The rows of a dynamic-programming sweep are sequential: one wavefront needs ~0.67 us per truth row (k_fwd_stripe), 6 - 9 ms
// for the 9 288-row alignment that bounds a whole-genome batch.  But the edit-distance recurrence is a linear map in the
// (min, +) semiring, and products of such maps lose rank: a few dozen rows below ANY starting row, the D row of the window is
// the true one up to an additive constant (every cheap path has funnelled through the same cells), and the flag bytes -- which
// only compare neighbouring D values -- are the true ones exactly.  Hence (Maleki, Musuvathi, Mytkowicz: "Parallelizing dynamic
// programming through rank convergence", PPoPP 2014):
//   1. k_fwdp_block<1>: every block of FP_S stripes (128 rows) sweeps from a guessed row (the alignment's first block from the
//      real one), writes its flags and records the D row at the end of every stripe: the TENTATIVE run.
//   2. k_fwdp_block<2>: every block but the first sweeps again, now from the row the block before it recorded at its end, and
//      compares its D row with the tentative one at the end of every stripe.  Once the two differ by one constant over all cells
//      of both planes, the rest of the tentative run is exact up to that constant: the block stops (normally after one stripe).
//   3. k_fwdp_finish: one wavefront per alignment chains the constants, takes the exit test's minimum and the end cells' distances
//      in the right frames and writes what k_fwd_stripe writes.  A block whose runs did not meet by its end (its recorded end row
//      was wrong, so the block behind it started wrong) sends the alignment to k_fwd_stripe, launched behind for exactly those.
// Exactness does not rest on the guess: a block's flags are kept only where its fix-up run, started from the row before, has
// produced them or has proved the tentative ones equal.  The induction starts at the first block, which is exact.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vcfdist_pr.h"
#include "pr_device.h"

#define FP_S 64                         // stripes per block (a multiple of 8: a block starts at a multiple of 64 rows); 512 rows: the runs meet after ~60

struct FwdParTables {
    int32_t *counter;       // [0] stripe slots, [1] block slots handed out
    int32_t *st_base;       // [launch position] first stripe slot of the alignment, -1: not swept by this launch
    int32_t *bl_base;       // [launch position] first block slot
    int2 *owner;            // [block slot] {launch position, block}
    int32_t *snap;          // [stripe slot][128]
    int2 *acc;              // [stripe slot]
    int4 *blk;              // [block slot]
    int4 *endc;             // [launch position]
    int32_t *fallback;      // [launch position] 1: the alignment goes to k_fwd_stripe
    int32_t cap_stripes, cap_blocks;
};

__global__ void k_fwdp_plan(const AlnDesc *__restrict__ descs, const int32_t *__restrict__ work, int n_work, FwdParTables T) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_work) return;
    const int a = work[k];
    int sb = -1, bb = -1;
    if (a >= 0) {
        const int n_stripes = (descs[a].Lt + FS_K - 1) / FS_K, n_blk = (n_stripes + FP_S - 1) / FP_S;
        sb = atomicAdd(T.counter, n_stripes);
        bb = atomicAdd(T.counter + 1, n_blk);
        if (sb + n_stripes > T.cap_stripes || bb + n_blk > T.cap_blocks) { sb = -1; bb = -1; }   // (cannot happen: sized by the launch's rows)
        else for (int j = 0; j < n_blk; j++) T.owner[bb + j] = make_int2(k, j);
    }
    T.st_base[k] = sb;
    T.bl_base[k] = bb;
    T.fallback[k] = (a >= 0 && sb < 0) ? 1 : 0;
}

template <int MODE>
__global__ void __launch_bounds__(64) k_fwdp_block(DevBatch B, const AlnDesc *__restrict__ descs, const int32_t *__restrict__ work,
                                                   uint8_t *__restrict__ ws, int32_t *__restrict__ blo_all, AlnOut *__restrict__ outs,
                                                   FwdParTables T) {
    const int slot = blockIdx.x;
    if (slot >= T.counter[1]) return;
    const int2 own = T.owner[slot];
    if (MODE == 2 && own.y == 0) return;            // the first block is exact as it is
    __builtin_amdgcn_s_setprio(2);
    const int a = work[own.x];
    const AlnDesc d = descs[a];
    const int n_stripes = (d.Lt + FS_K - 1) / FS_K;
    const int sb = T.st_base[own.x];
    FwdParOut FP;
    FP.snap = T.snap + size_t(sb) * 128;
    FP.acc = T.acc + sb;
    FP.blk = T.blk + slot;
    FP.endc = T.endc + own.x;
    fwd_stripe_range<MODE>(B, d, a, ws, blo_all, outs, own.y * FP_S, min(own.y * FP_S + FP_S, n_stripes), FP);
}

__global__ void __launch_bounds__(64) k_fwdp_finish(const AlnDesc *__restrict__ descs, const int32_t *__restrict__ work, int n_work,
                                                    AlnOut *__restrict__ outs, FwdParTables T) {
    const int k = blockIdx.x;
    if (k >= n_work) return;
    const int sb = T.st_base[k], bb = T.bl_base[k];
    if (sb < 0) return;
    const int a = work[k];
    const int lane = threadIdx.x;
    const int Lt = descs[a].Lt;
    const int n_stripes = (Lt + FS_K - 1) / FS_K, n_blk = (n_stripes + FP_S - 1) / FP_S;
    // frames: a run's D values are the true ones plus a constant -- alpha_b for block b's fix-up run (the frame of the row it
    // started from: the tentative frame of the block before), beta_b for its tentative run; where they met, fix-up - tentative =
    // delta_b, so beta_b = alpha_b - delta_b.  Block 0 is exact: beta_0 = 0.  Stripes of a block up to the meeting stripe were
    // last written by the fix-up run, the rest by the tentative run.
    int em = D_INF, mt = D_INF;
    int beta_prev = 0;
    bool fail = false;
    int alpha_last = 0, beta_last = 0, met_last = -1;
    for (int b = 0; b < n_blk; b++) {
        int alpha = 0, beta = 0, met = -1;      // met: last stripe of the block (local) in the fix-up frame
        if (b > 0) {
            const int4 rec = T.blk[bb + b];
            alpha = beta_prev;
            if (lane == 0) {      // (diagnostics: VPR_FWDP_STATS)
                atomicAdd(T.counter + 2, 1);
                if (rec.x >= 0) { atomicAdd(T.counter + 3, 1); atomicAdd(T.counter + 4, rec.x); }
                else if (b != n_blk - 1) atomicAdd(T.counter + 5, 1);
            }
            if (rec.x >= 0) { met = rec.x; beta = alpha - rec.y; }
            else if (b == n_blk - 1) { met = FP_S - 1; beta = alpha; }       // the last block may run to its end: all fix-up frame
            else { fail = true; break; }
        }
        // this block's stripes: lanes over stripes
        const int s0 = b * FP_S, ns = min(FP_S, n_stripes - s0);
        if (lane < ns) {
            const int2 ac = T.acc[sb + s0 + lane];
            const int c = (b == 0) ? 0 : (lane <= met ? alpha : beta);
            if (ac.x < D_INF / 2) em = min(em, ac.x - c);
            if (ac.y < D_INF / 2) mt = min(mt, ac.y - c);
        }
        beta_prev = beta;
        alpha_last = alpha; beta_last = beta; met_last = met;
    }
    if (fail) { if (lane == 0) { T.fallback[k] = 1; atomicAdd(T.counter + 6, 1); } return; }
    wave_prefix_min2(em, mt);
    if (lane == 63) {
        const int4 ec = T.endc[k];
        int dq, dr;
        if (n_blk == 1) { dq = ec.x; dr = ec.y; }
        else if (T.blk[bb + n_blk - 1].x < 0) { dq = ec.z < D_INF / 2 ? ec.z - alpha_last : D_INF; dr = ec.w < D_INF / 2 ? ec.w - alpha_last : D_INF; }
        else { dq = ec.x < D_INF / 2 ? ec.x - beta_last : D_INF; dr = ec.y < D_INF / 2 ? ec.y - beta_last : D_INF; }
        (void)met_last;
        outs[a].dist_q = dq;
        outs[a].dist_r = dr;
        outs[a].exit_min = em;
        outs[a].path_len = (mt < D_INF) ? mt + 1 : 0;      // (see k_fwd_stripe)
    }
}
