// Picks the prefill kernel generation.  The software-pipelined kernel (prefill_kernel.hip) is the
// product path; the baseline (prefill_baseline.hip) stays in the library as an in-process A/B
// reference for tools/prefill_ab.py and can be forced with SFA_PREFILL_IMPL=0.
#include <cstdlib>

#include "prefill_common.h"

namespace sfa {

int launch_prefill(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream) {
    static const int impl = [] {
        const char *e = std::getenv("SFA_PREFILL_IMPL");
        return e ? std::atoi(e) : 1;
    }();
    const char *e = std::getenv("SFA_PREFILL_IMPL_DYNAMIC");      // A/B harness only: re-read every call
    const int which = e ? std::atoi(e) : impl;
    if (which == 0) return launch_prefill_baseline(p, dtype, head_dim, causal, stream);
    if (which >= 2) return launch_prefill_variant(which, p, dtype, head_dim, causal, stream);
    return launch_prefill_main(p, dtype, head_dim, causal, stream);
}

}  // namespace sfa
