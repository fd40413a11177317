// Picks the prefill kernel.
//   head_dim 256: prefill_w4d_kernel.hip (the 4-wave persistent structure, 32 query rows per wave); prefill_d256_kernel.hip
//     (compiler-scheduled) only when a head's rows do not fit 32-bit buffer descriptors, or forced with prefill_impl 61.
//   auto (default), head_dim 128:
//     the 4-wave persistent kernel (prefill_w4_kernel.hip: one wave per SIMD, 64 query rows per wave,
//     O^T in the accumulator file, K/V by LDS-DMA, 256 persistent workgroups) whenever the problem
//     has enough 256-row q-tiles to feed the 256 CUs (and, under the causal mask, rows long enough to amortise
//     its per-q-tile fixed costs: the rule and its measurements are in launch_prefill below);
//   otherwise:
//     the 8-wave 256-row software-pipelined kernel (prefill_kernel.hip) whenever the problem has enough of
//     its workgroups (a pair of 256-row q-tiles each) for about half the 256 CUs; smaller problems take the
//     128-row geometry (prefill_kernel_bm128.hip: four times the workgroups).
//   Each geometry comes in two numeric flavours: exact scale (scores = fp32 QK^T times the scale in
//   fp32) and prescaled Q (Q * scale * log2 e rounded to 16 bit once per q-tile, the scale pass
//   gone from the inner loop: +5 %, score error growing with the logits).  Exact is the default;
//   the prescaled flavour runs only for callers that set sfa_prefill_args.fast_scale and want no
//   log-sum-exp.  Within a flavour the geometries agree to fp32 summation order.
//   Forced choices, through sfa_debug_set("prefill_impl", n) only (tests, tools/ -- the launch path reads
//   no environment variable):  1 / 20 / 40 force the 8-wave 256-row / the 128-row / the 4-wave kernel
//   (flavour by the rule above); 3 / 10, 21 / 22 and 41 / 42 force prescaled / exact of the three;
//   2 and 4 are diagnostic variants of the 8-wave kernel (tools/prefill_ab.py, tools/prefill_stamps.py).
//   0 (the baseline generation) and 30..32 (the 16x16x32-MFMA generation) exist only in the A/B build
//   of the library (build.py build_lib(variants=True), -DSFA_WITH_VARIANTS): they are never an auto choice.
#include "prefill_common.h"

namespace sfa {

int launch_prefill(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream) {
    int which = g_knobs.prefill_impl.load(std::memory_order_relaxed);
    auto ran = [](int id) { g_knobs.last_prefill_kernel.store(id, std::memory_order_relaxed); };
    const int flavour = p.fast_scale ? 1 : 2;       // what "by policy" resolves to (1 prescaled, 2 exact)
    if (head_dim == 256) {      // the persistent kernel; 61 forces the compiler-scheduled one (tests, A/B)
        if (which != 61 && prefill_w4d_serves(p)) { ran(60); return launch_prefill_w4d(p, dtype, causal, stream); }
        ran(61);
        return launch_prefill_d256(p, dtype, causal, stream);
    }
    if (which < 0) {
        const long long nq = (p.Sq + 255) / 256;
        const long long qtiles = (long long)p.B * p.Hq * nq;
        // Measured crossover of the 4-wave persistent kernel against the best of the other two (tools/prefill_crossover.sh,
        // round 3's kernel, profiles/r03_prefill_crossover.txt).  Full attention: from 256 q-tiles on (+10..25 %; 128: -3 %),
        // whatever the key count (8 x 32 x 4096 queries against 64 .. 1024 keys: +16..25 %).  Under the causal mask its
        // per-q-tile fixed costs weigh more on short rows: 16+ q-tiles per head from 256 q-tiles on (+2..7 %), 8 per head from
        // 1024 (512: -2 %, 2048: +8 %), 4 per head (seqlen 1024) from 2048 (+2 %; 4096: +11 %).  Fewer keys than queries
        // under the (bottom-right aligned) causal mask leaves q-tiles with few or no keys -- 8 x 32 x 4096 against 1024 keys
        // -20 %, against 2048 -6 % -- so those go to the other kernels.
        const bool w4_pays = !causal ? qtiles >= kW4MinTiles
                           : p.Sk < p.Sq ? false
                           : nq >= 16 ? qtiles >= 256 : nq >= 8 ? qtiles >= 1024 : nq >= 4 ? qtiles >= 2048 : false;
        const bool w4 = w4_pays && prefill_w4_serves(p, head_dim);
        if (w4) {
            which = 40;
        } else {
            // the 8-wave kernel runs one workgroup per PAIR of q-tiles.  Measured crossover
            // (tools/prefill_small_grids.sh): causal, 64 pair-workgroups 128-row +18..39 %, 128: -10..+7 %,
            // 192+: 256-row +15 %; full attention (pairing balances nothing there), 128: 128-row +39 %, 192: -3 %
            const long long wgs = (long long)p.B * p.Hq * ((nq + 1) / 2);
            which = wgs < (causal ? 128 : 192) ? 20 : 1;
        }
    }
#ifdef SFA_WITH_VARIANTS
    // the stamping / event-log builds (A/B library only) write up to 4 x 512 u64 into the caller's lse buffer instead of the lse
    {
        const int f = which >= 80 ? which - 80 : which - 40;
        const bool stamps = which >= 80 ? (which < 120 && (f == 4 || (f >= 16 && f <= 19) || f >= 21))     // round 2's kernel
                                        : (which == 43 || which == 44);
        if (stamps && (!p.lse || (long long)p.B * p.Hq * p.Sq * (long long)sizeof(float) < 4 * 512 * 8))
            return fail(SFA_ERR_BAD_SHAPE, "prefill_impl %d is a stamping build: it needs an lse buffer of at least 16 KiB", which);
    }
    if (which == 0) return launch_prefill_baseline(p, dtype, head_dim, causal, stream);
    if (which >= 30 && which <= 32) return launch_prefill_x16(p, dtype, head_dim, causal, stream, which - 30);
    if (which >= 80 && which <= 119) return launch_prefill_w4r2(p, dtype, head_dim, causal, stream, which - 80);
#else
    if (which == 0 || (which >= 30 && which <= 32) || (which >= 80 && which <= 119))
        return fail(SFA_ERR_BAD_SHAPE, "prefill_impl %d needs the A/B build of the library (build_lib(variants=True))", which);
#endif
    if (which >= 40 && which <= 44) { ran(which == 40 ? 40 + flavour : which); return launch_prefill_w4(p, dtype, head_dim, causal, stream, which - 40); }
    if (which >= 20 && which <= 22) { ran(which == 20 ? 20 + flavour : which); return launch_prefill_bm128(p, dtype, head_dim, causal, stream, which - 20); }
    if (which >= 2) { ran(which); return launch_prefill_variant(which, p, dtype, head_dim, causal, stream); }
    ran(p.fast_scale ? 3 : 1);
    return launch_prefill_main(p, dtype, head_dim, causal, stream);
}

}  // namespace sfa
