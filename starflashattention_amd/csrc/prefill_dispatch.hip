// Picks the prefill kernel generation.  v1 (software-pipelined) is the product path; v0 (the
// straightforward one-tile-at-a-time kernel) stays in the library as an in-process A/B baseline
// for tools/prefill_ab.py and can be forced with SFA_PREFILL_IMPL=0.
#include <cstdlib>

#include "prefill_common.h"

namespace sfa {

int launch_prefill(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream) {
    static const int impl = [] {
        const char *e = std::getenv("SFA_PREFILL_IMPL");
        return e ? std::atoi(e) : 5;
    }();
    const char *e = std::getenv("SFA_PREFILL_IMPL_DYNAMIC");      // A/B harness only: re-read every call
    const int which = e ? std::atoi(e) : impl;
    if (which >= 100) return launch_prefill_ablation(p, which - 100, dtype, head_dim, causal, stream);
    if (which == 0) return launch_prefill_v0(p, dtype, head_dim, causal, stream);
    if (which == 1) return launch_prefill_v1(p, dtype, head_dim, causal, stream);
    if (which == 2) return launch_prefill_v2(p, dtype, head_dim, causal, stream);
    if (which == 5) return launch_prefill_v5(p, dtype, head_dim, causal, stream);
    if (which == 6) return launch_prefill_v6(p, dtype, head_dim, causal, stream);
    if (which == 4) return launch_prefill_v4(p, dtype, head_dim, causal, stream);
    return launch_prefill_v3(p, dtype, head_dim, causal, stream);
}

}  // namespace sfa
