// pr_scan.h -- wave-level scans over DPP (no LDS traffic): shared by the dense kernels (pr_kernels.hip, pr_strip.hip) and the
// window kernels (pr_band.hip, pr_wide.hip).
#ifndef PR_SCAN_H_
#define PR_SCAN_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---------------------------------------------------------------------------
// DPP wave scans (gfx9 row_shr / row_bcast / wave_shr): ~12 VALU ops instead of six ds_bpermute hops
// ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
// Two independent inclusive prefix-min scans over the wave, interleaved, as 12 v_min_i32_dpp: with
// bound_ctrl off a lane whose DPP source does not exist is simply disabled and keeps its value.  hipcc emits
// v_mov_dpp + s_nop + v_min for the builtin form (3x the instructions), and it cannot see the DPP read
// inside an asm statement, so the two wait states a DPP read needs after a VALU write of the same register
// are spelled out here (the other scan's step + one s_nop).
__device__ __forceinline__ void wave_prefix_min2(int &a, int &b) {
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
        "s_nop 0\n\t"
        "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_min_i32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_min_i32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b));
}
__device__ __forceinline__ int wave_shr1(int x, int fill) { return dpp_mov<0x138, 0xf>(fill, x); }   // lane i <- lane i-1

#endif
