// pr_plan.h -- rocPRIM-backed primitives of the device-side planner (pr_plan.hip), called from pr_api.hip
#ifndef PR_PLAN_H_
#define PR_PLAN_H_
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
int vplan_sort_pairs_desc(void *tmp, size_t *tmp_bytes, const uint16_t *keys, uint16_t *keys_out, const int32_t *vals, int32_t *vals_out,
                          size_t n, hipStream_t st);
int vplan_exclusive_scan_u32(void *tmp, size_t *tmp_bytes, const uint32_t *in, uint32_t *out, size_t n, hipStream_t st);
#endif
