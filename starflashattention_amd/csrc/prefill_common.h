// Shared pieces of the gfx950 prefill kernels: tile geometry, the XOR-swizzled LDS images of the
// baseline generation and the block -> (head, q-tile) map.  Design notes: prefill_kernel.hip.
#pragma once
#include "sfa_device.h"
#include "sfa_host.h"

namespace sfa {
namespace prefill {

constexpr int kBM = 256;      // query rows per workgroup
constexpr int kBN = 64;       // keys per tile
constexpr int kThreads = 512;

__device__ __forceinline__ float ninf() { return -__builtin_huge_valf(); }

// byte offset of 16-byte chunk `ch` of row `row` inside a [kBN][D] 16-bit LDS tile
template <int D>
__device__ __forceinline__ int k_off(int row, int ch) {
    if (D == 128) return 256 * row + 16 * (ch ^ (row & 15));
    return 128 * row + 16 * (ch ^ ((row >> 1) & 7));
}
template <int D>
__device__ __forceinline__ int v_off(int row, int ch) {
    if (D == 128) return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
    return 128 * row + 16 * (ch ^ (((row >> 1) & 1) << 2));
}

typedef __attribute__((address_space(3))) i16x4 lds_i16x4;


// XCD-aware decode of blockIdx.x: XCD x (= bid % 8 under round-robin dispatch; a speed hint only)
// owns heads [x*bh_per_xcd, (x+1)*bh_per_xcd) and walks each head's q-tiles heaviest-first.
struct BlockCoord { int bh, qt; };
__device__ __forceinline__ BlockCoord block_coord(const PrefillKernelParams &p) {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    BlockCoord c;
    c.bh = xcd * p.bh_per_xcd + slot / p.nq_tiles;
    c.qt = p.nq_tiles - 1 - (slot % p.nq_tiles);
    return c;
}

}  // namespace prefill

// the product kernel and the baseline generation kept for A/B runs; launch_prefill picks one
int launch_prefill_main(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream);
int launch_prefill_variant(int which, const PrefillKernelParams &p, int dtype, int head_dim, bool causal,
                           hipStream_t stream);
int launch_prefill_bm128(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream,
                         int force = 0);
int launch_prefill_x16(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream,
                       int force = 0);
int launch_prefill_baseline(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream);
// the 4-wave persistent kernel (prefill_w4_kernel.hip); force: 0 = flavour by policy, 1 = prescaled, 2 = exact
int launch_prefill_w4(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream,
                      int force = 0);
// whether one head's Q / K / V rows fit the 32-bit buffer descriptors of the 4-wave kernel (else: the 8-wave kernel)
bool prefill_w4_serves(const PrefillKernelParams &p, int head_dim);
// the flavours of the 4-wave kernel that live in translation units of their own (prefill_w4_kernel_p1..3.hip; bf16 exact
// scale is in prefill_w4_kernel.hip itself): called by launch_prefill_w4 only
int launch_prefill_w4_fp16_exact(const PrefillKernelParams &p, bool causal, hipStream_t stream);
int launch_prefill_w4_fp16_prescaled(const PrefillKernelParams &p, bool causal, hipStream_t stream);
int launch_prefill_w4_bf16_prescaled(const PrefillKernelParams &p, bool causal, hipStream_t stream);
// round 2's generation of the 4-wave kernel (q-tiles scored and finished outside the pipeline) and its stamping /
// ablation builds: the A/B library only (prefill_w4r2_kernel.hip)
int launch_prefill_w4r2(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream,
                        int force = 0);
// head_dim 256: the one-wave-per-SIMD persistent kernel (prefill_w4d_kernel.hip) wherever one head's rows fit its
// 32-bit buffer descriptors, the compiler-scheduled kernel (prefill_d256_kernel.hip) otherwise
int launch_prefill_w4d(const PrefillKernelParams &p, int dtype, bool causal, hipStream_t stream);
bool prefill_w4d_serves(const PrefillKernelParams &p);
int launch_prefill_d256(const PrefillKernelParams &p, int dtype, bool causal, hipStream_t stream);
// q-tiles (256 rows) a full-attention problem must have before the auto rule picks the persistent kernel: one per CU
constexpr long long kW4MinTiles = 256;

}  // namespace sfa
