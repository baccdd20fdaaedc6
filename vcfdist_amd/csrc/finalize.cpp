// finalize.cpp -- store_phase as a plain host function for callers that hold four
// distances (the device path evaluates the same expression in k_finalize), and the version string.
#include <cstdint>

#include "../../include/vcfdist_pr.h"

extern "C" {

// store_phase, src/dist.cpp:449-475: float division compared against the double
// threshold, in this order of tests.
int32_t vpr_store_phase(const int32_t s[4], double phase_threshold,
                        int32_t *orig_phase_dist, int32_t *swap_phase_dist) {
    const int orig = s[0] + s[3];   // QUERY1_TRUTH1 + QUERY2_TRUTH2
    const int swp = s[2] + s[1];    // QUERY2_TRUTH1 + QUERY1_TRUTH2
    int phase = VPR_PHASE_NONE;
    if (orig != swp) {
        if (orig == 0) phase = VPR_PHASE_ORIG;
        else if (swp == 0) phase = VPR_PHASE_SWAP;
        else if (1 - float(swp) / orig > phase_threshold) phase = VPR_PHASE_SWAP;
        else if (1 - float(orig) / swp > phase_threshold) phase = VPR_PHASE_ORIG;
    }
    if (orig_phase_dist) *orig_phase_dist = orig;
    if (swap_phase_dist) *swap_phase_dist = swp;
    return phase;
}

const char *vpr_version(void) { return "vcfdist_amd 0.1 (gfx950)"; }

}  // extern "C"
