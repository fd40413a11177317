// Picks the prefill kernel.
//   auto (default): the 256-row software-pipelined kernel (prefill_kernel.hip) whenever the problem
//     has enough of its workgroups (a pair of 256-row q-tiles each) for about half the 256 CUs; smaller
//     problems take the 128-row geometry (prefill_kernel_bm128.hip: four times the workgroups).
//     Each geometry comes in two numeric flavours: exact scale (scores = fp32 QK^T times the scale in
//     fp32) and prescaled Q (Q * scale * log2 e rounded to 16 bit once per q-tile, the scale pass
//     gone from the inner loop: +5 %, score error growing with the logits).  Exact is the default;
//     the prescaled flavour runs only for callers that set sfa_prefill_args.fast_scale and want no
//     log-sum-exp.  Within a flavour the two geometries are bit-identical.
//   SFA_PREFILL_IMPL=0 / 1 / 20 force the baseline generation / the 256-row / the 128-row kernel
//   (flavour by the rule above); 3 / 10 and 21 / 22 force prescaled / exact of the 256- and 128-row
//   kernels; 2 and 4 are diagnostic variants (tools/prefill_ab.py, tools/prefill_stamps.py).
#include <cstdlib>

#include "prefill_common.h"

namespace sfa {

int launch_prefill(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream) {
    static const int impl = [] {
        const char *e = std::getenv("SFA_PREFILL_IMPL");
        return e ? std::atoi(e) : -1;
    }();
    const char *e = std::getenv("SFA_PREFILL_IMPL_DYNAMIC");      // tests / A-B harness: re-read every call
    int which = e ? std::atoi(e) : impl;
    if (which < 0) {
        // the 256-row kernel runs one workgroup per PAIR of q-tiles.  Measured crossover
        // (tools/prefill_small_grids.sh): causal, 64 pair-workgroups 128-row +18..39 %, 128: -10..+7 %,
        // 192+: 256-row +15 %; full attention (pairing balances nothing there), 128: 128-row +39 %, 192: -3 %
        const long long wgs = (long long)p.B * p.Hq * (((p.Sq + 255) / 256 + 1) / 2);
        which = wgs < (causal ? 128 : 192) ? 20 : 1;
    }
    if (which == 0) return launch_prefill_baseline(p, dtype, head_dim, causal, stream);
    if (which >= 30 && which <= 32) return launch_prefill_x16(p, dtype, head_dim, causal, stream, which - 30);
    if (which >= 20 && which <= 22) return launch_prefill_bm128(p, dtype, head_dim, causal, stream, which - 20);
    if (which >= 2) return launch_prefill_variant(which, p, dtype, head_dim, causal, stream);
    return launch_prefill_main(p, dtype, head_dim, causal, stream);
}

}  // namespace sfa
