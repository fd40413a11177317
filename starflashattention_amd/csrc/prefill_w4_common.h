// Building blocks shared by the one-wave-per-SIMD prefill kernels (prefill_w4_kernel.hip: head_dim 128, 64 query
// rows per wave; prefill_w4d_kernel.hip: head_dim 256, 32 rows per wave): the asm MFMA wrappers with their operands
// pinned to a register file, the hazard fences, the LDS-DMA piece, the key mask, compile-time loops and the row-max
// stage.  What each is for is explained at the top of prefill_w4_kernel.hip.
#pragma once
#include <utility>

#include "prefill_core.h"

namespace sfa {
namespace w4c {

using namespace prefill;

constexpr int kThreadsW4 = 256;     // 4 waves, one per SIMD
constexpr float kThr = 8.0f;        // lazy-rescale threshold (log2 units)

typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) const u32x4 lds_cu4;
__device__ __forceinline__ u32x4 lds_read16(const lds_char *p) { return *reinterpret_cast<lds_cu4 *>(p); }

// ---- MFMA wrappers: operands by register file ---------------------------------------
// s (VGPR) = k (VGPR) . q (AGPR) [+ s]
template <class Tr>
__device__ __forceinline__ void mfma_qk_first(f32x16 &s, typename Tr::mfma_vec k, typename Tr::mfma_vec q) {
    if constexpr (Tr::id == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(k), "a"(q));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(s) : "v"(k), "a"(q));
}
template <class Tr>
__device__ __forceinline__ void mfma_qk_first_c(f32x16 &s, typename Tr::mfma_vec k, typename Tr::mfma_vec q, const f32x16 &c) {
    // C operand = a VALU-written register tuple: two wait states in front (hazard (2))
    if constexpr (Tr::id == 1) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s) : "v"(k), "a"(q), "v"(c));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s) : "v"(k), "a"(q), "v"(c));
}
template <class Tr>
__device__ __forceinline__ void mfma_qk(f32x16 &s, typename Tr::mfma_vec k, typename Tr::mfma_vec q) {
    if constexpr (Tr::id == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q));
}
// o (AGPR) += v (VGPR) . p (VGPR);  NOP: the operands may have been written by the VALU just before
template <class Tr, bool NOP>
__device__ __forceinline__ void mfma_pv(f32x16 &o, typename Tr::mfma_vec v, typename Tr::mfma_vec pfrag) {
    if constexpr (Tr::id == 1) {
        if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pfrag));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pfrag));
    } else {
        if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pfrag));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pfrag));
    }
}
// Let every MFMA issued so far drain before the VALU touches a result (hazard (1)): 2 x 16 wait states,
// tied to the values so neither side of the fence can be scheduled across it.
__device__ __forceinline__ void settle(f32x16 &x) { asm volatile("s_nop 15\n\ts_nop 15" : "+v"(x)); }
__device__ __forceinline__ void settle_acc(f32x16 &x) { asm volatile("s_nop 15\n\ts_nop 15" : "+a"(x)); }

// One 1-KiB LDS-DMA piece: 64 lanes x 16 B from `srd`[voff + soff] to LDS[lds .. lds + 1024).
// M0 is written in the statement that uses it (hipcc does not preserve it around asm).
__device__ __forceinline__ void dma_piece(unsigned lds, unsigned voff, u32x4s srd, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

// s[r] = -inf where key kbase + (r&3) + 8*(r>>2) + 4*h2 lies beyond `lim`: one subtract per call, then a compare
// against a literal and a select per register (prefill_core.h's mask_half builds sixteen key indices first, and
// hipcc hoists those out of the rare branch into the MFMA gaps).
__device__ __forceinline__ void mask_keys(f32x16 &s, int kbase, int h2, int lim) {
    asm volatile("" : "+s"(kbase));                     // (opaque: hipcc otherwise hoists the subtract out of the rare branch, into every half-step)
    const int room = lim - kbase - 4 * h2;              // keys with offset <= room stay
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if ((r & 3) + 8 * (r >> 2) > room) s[r] = ninf();
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>) -- every register index in
// the gap program must be a constant (a runtime-indexed f32x16 goes to scratch)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// m = max(m, a, b), in place among the MFMAs (and without the canonicalising v_max hipcc puts in front of fmaxf)
__device__ __forceinline__ void st_max3(float &m, const float &a, const float &b) {
    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(a), "v"(b));
}
// the first pair of a block: no -inf to start from
__device__ __forceinline__ void st_max2(float &m, const float &a, const float &b) {
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
}

}  // namespace w4c
}  // namespace sfa
