// pr_plan.hip -- the device side of planning a round-0 batch (replaces the host's counting sort and prefix sums of
// sort_superclusters' stand-in, make_plan in pr_api.hip; reference: cluster.cpp:42-122 sizes and orders the work units on the
// host, one supercluster at a time).  Own translation unit: rocPRIM's device-wide radix sort and scan instantiate dozens of
// kernels, which the planner's file need not recompile.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "pr_plan.h"

// stable, descending by key: (keys, vals) -> (keys_out, vals_out).  tmp == nullptr: only *tmp_bytes is set.
int vplan_sort_pairs_desc(void *tmp, size_t *tmp_bytes, const uint16_t *keys, uint16_t *keys_out, const int32_t *vals, int32_t *vals_out,
                          size_t n, hipStream_t st) {
    return int(rocprim::radix_sort_pairs_desc(tmp, *tmp_bytes, keys, keys_out, vals, vals_out, n, 0u, 16u, st));
}

// exclusive prefix sums of uint32 values (workspace offsets in 128-byte units: the caller checked that the total fits)
int vplan_exclusive_scan_u32(void *tmp, size_t *tmp_bytes, const uint32_t *in, uint32_t *out, size_t n, hipStream_t st) {
    return int(rocprim::exclusive_scan(tmp, *tmp_bytes, in, out, uint32_t(0), n, rocprim::plus<uint32_t>(), st));
}
