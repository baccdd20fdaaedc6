// Access-width microbenchmark (kernel tuning aid): a wave streams through private blocks the way k_zero_lane does -- three passes, 56 B per
// lane-row, 4- and 8-byte accesses ("narrow") or the logs and the truth words in 16-byte pieces ("wide").  hipcc --offload-arch=gfx950 -O3.
// MI355X, 62 500 waves x 44 rows: narrow 5.9 TB/s, wide 6.0 TB/s: the memory system is not what bounds the lane kernel (3.1 TB/s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
typedef __attribute__((ext_vector_type(2))) unsigned int u2;
#define OOB 0xfffffff0u
// layout per wave: in: Q rows, R rows, T rows (256 B each); log: flags 512 B/row, ppw 256 B/row, steps 512 B/row
template <int WIDE>
__global__ void __launch_bounds__(64, 6) k(const uint32_t *__restrict__ in, uint4 *__restrict__ log, int rows, uint32_t *__restrict__ sink) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(in + size_t(w) * 64 * 3 * rows), 0, 256 * 3 * rows, 0x00020000);
    const auto rlog = __builtin_amdgcn_make_buffer_rsrc(log + size_t(w) * 80 * rows, 0, 1280 * rows, 0x00020000);
    const uint32_t l4 = lane << 2, l8 = lane << 3, l16 = lane << 4;
    const uint32_t posR = uint32_t(rows) << 8, posT = uint32_t(rows) << 9, logP = uint32_t(rows) << 9, logS = logP + (uint32_t(rows) << 8);
    uint32_t acc = 0;
    // pass 1: forward
    uint32_t x = 0, z = 0;
    u4 tq = {0, 0, 0, 0}; u2 keep = {0, 0};
    for (int t = 0; t < rows; t++) {
        uint32_t tw;
        if (WIDE) { if ((t & 3) == 0) tq = __builtin_amdgcn_raw_buffer_load_b128(rin, posT + (uint32_t(t >> 2) << 10) + l16, 0, 0); tw = tq[t & 3]; }
        else tw = __builtin_amdgcn_raw_buffer_load_b32(rin, posT + (uint32_t(t) << 8) + l4, 0, 0);
        const uint32_t qw = __builtin_amdgcn_raw_buffer_load_b32(rin, (x << 8) + l4, 0, 0);
        const uint32_t rw = __builtin_amdgcn_raw_buffer_load_b32(rin, posR + (z << 8) + l4, 0, 0);
        acc += tw ^ qw ^ rw;
        x = min(x + 1 + (qw & 0u), uint32_t(rows - 1)); z = min(z + 1 + (rw & 0u), uint32_t(rows - 1));      // (the next addresses depend on the loads)
        u2 e; e.x = acc; e.y = qw;
        if (WIDE) {
            if (t & 1) { u4 p; p.x = keep.x; p.y = keep.y; p.z = e.x; p.w = e.y; __builtin_amdgcn_raw_buffer_store_b128(p, rlog, (uint32_t(t >> 1) << 10) + l16, 0, 0); }
            else keep = e;
        } else __builtin_amdgcn_raw_buffer_store_b64(e, rlog, (uint32_t(t) << 9) + l8, 0, 0);
    }
    // pass 2: backward over the flags, writes ppw
    u4 pq = {0, 0, 0, 0}, fq = {0, 0, 0, 0};
    for (int t = rows - 1; t >= 0; t--) {
        u2 e;
        if (WIDE) { if ((t & 1) == 1 || t == rows - 1) fq = __builtin_amdgcn_raw_buffer_load_b128(rlog, (uint32_t(t >> 1) << 10) + l16, 0, 0); e.x = (t & 1) ? fq.z : fq.x; e.y = (t & 1) ? fq.w : fq.y; }
        else e = __builtin_amdgcn_raw_buffer_load_b64(rlog, (uint32_t(t) << 9) + l8, 0, 0);
        acc = acc * 3 + e.x + e.y;
        if (WIDE) { pq[t & 3] = acc; if ((t & 3) == 0) __builtin_amdgcn_raw_buffer_store_b128(pq, rlog, logP + (uint32_t(t >> 2) << 10) + l16, 0, 0); }
        else __builtin_amdgcn_raw_buffer_store_b32(acc, rlog, logP + (uint32_t(t) << 8) + l4, 0, 0);
    }
    // pass 3: walk
    x = 0;
    for (int t = 0; t < rows; t++) {
        uint32_t pw, tw;
        if (WIDE) {
            if ((t & 3) == 0) { pq = __builtin_amdgcn_raw_buffer_load_b128(rlog, logP + (uint32_t(t >> 2) << 10) + l16, 0, 0); tq = __builtin_amdgcn_raw_buffer_load_b128(rin, posT + (uint32_t(t >> 2) << 10) + l16, 0, 0); }
            pw = pq[t & 3]; tw = tq[t & 3];
        } else {
            pw = __builtin_amdgcn_raw_buffer_load_b32(rlog, logP + (uint32_t(t) << 8) + l4, 0, 0);
            tw = __builtin_amdgcn_raw_buffer_load_b32(rin, posT + (uint32_t(t) << 8) + l4, 0, 0);
        }
        const uint32_t cw = __builtin_amdgcn_raw_buffer_load_b32(rin, (x << 8) + l4, 0, 0);
        x = min(x + 1 + (cw & 0u), uint32_t(rows - 1));
        acc += pw ^ tw ^ cw;
        u2 e; e.x = acc; e.y = cw;
        if (WIDE) {
            if (t & 1) { u4 p; p.x = keep.x; p.y = keep.y; p.z = e.x; p.w = e.y; __builtin_amdgcn_raw_buffer_store_b128(p, rlog, logS + (uint32_t(t >> 1) << 10) + l16, 0, 0); }
            else keep = e;
        } else __builtin_amdgcn_raw_buffer_store_b64(e, rlog, logS + (uint32_t(t) << 9) + l8, 0, 0);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char **argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 44, waves = argc > 2 ? atoi(argv[2]) : 62500;
    uint32_t *in, *sink; uint4 *log;
    hipMalloc(&in, size_t(waves) * 64 * 3 * rows * 4); hipMalloc(&log, size_t(waves) * 80 * rows * 16); hipMalloc(&sink, 64);
    hipMemset(in, 1, size_t(waves) * 64 * 3 * rows * 4); hipMemset(log, 0, size_t(waves) * 80 * rows * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; mode++)
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            if (mode) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, in, log, rows, sink);
            else hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, in, log, rows, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double bytes = double(waves) * 64 * rows * 56.0;
            printf("%s rows %d waves %d: %.3f ms, %.2f TB/s (56 B per lane-row)\n", mode ? "wide  " : "narrow", rows, waves, ms, bytes / ms * 1e-9);
        }
    return 0;
}
