// Device-side helpers shared by the gfx950 kernels: 16-bit dtype traits, wave64
// reductions (DPP inside a row, v_permlane32_swap across the two 32-lane halves),
// and the MFMA fragment types.  gfx950 only -- no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfa {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

template <typename To, typename From>
__device__ __forceinline__ To bitcast(const From &f) {
    static_assert(sizeof(To) == sizeof(From), "size mismatch");
    return __builtin_bit_cast(To, f);
}

// ---- 16-bit storage types ---------------------------------------------------------
struct Fp16 {
    using mfma_vec = f16x8;
    static constexpr int id = 0;
    static constexpr uint32_t ones2 = 0x3C003C00u;      // (1.0, 1.0)
    static __device__ __forceinline__ float to_f32(uint16_t b) { return (float)bitcast<_Float16>(b); }
    static __device__ __forceinline__ uint16_t from_f32(float f) { return bitcast<uint16_t>((_Float16)f); }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        f16x2 v = {(_Float16)lo, (_Float16)hi};            // round-to-nearest-even each
        return bitcast<uint32_t>(v);
    }
    static __device__ __forceinline__ float lo_f32(uint32_t w) { return (float)bitcast<f16x2>(w)[0]; }
    static __device__ __forceinline__ float hi_f32(uint32_t w) { return (float)bitcast<f16x2>(w)[1]; }
    // acc + a.lo*b.lo + a.hi*b.hi  (v_dot2c_f32_f16)
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
        return __builtin_amdgcn_fdot2(bitcast<f16x2>(a), bitcast<f16x2>(b), acc, false);
    }
    static __device__ __forceinline__ f32x16 mfma32(mfma_vec a, mfma_vec b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

struct Bf16 {
    using mfma_vec = bf16x8;
    static constexpr int id = 1;
    static constexpr uint32_t ones2 = 0x3F803F80u;      // (1.0, 1.0)
    static __device__ __forceinline__ float to_f32(uint16_t b) { return bitcast<float>((uint32_t)b << 16); }
    static __device__ __forceinline__ uint16_t from_f32(float f) { return bitcast<uint16_t>((__bf16)f); }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        bf16x2 v = {(__bf16)lo, (__bf16)hi};               // v_cvt_pk_bf16_f32 (RNE, NaN-safe)
        return bitcast<uint32_t>(v);
    }
    static __device__ __forceinline__ float lo_f32(uint32_t w) { return bitcast<float>(w << 16); }
    static __device__ __forceinline__ float hi_f32(uint32_t w) { return bitcast<float>(w & 0xffff0000u); }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
        return __builtin_amdgcn_fdot2_f32_bf16(bitcast<bf16x2>(a), bitcast<bf16x2>(b), acc, false);
    }
    static __device__ __forceinline__ f32x16 mfma32(mfma_vec a, mfma_vec b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

// ---- cross-lane -------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return bitcast<float>(__builtin_amdgcn_update_dpp(0, bitcast<int>(x), CTRL, 0xf, 0xf, true));
}
// Sum over aligned groups of LANES (4, 8 or 16) lanes; every lane gets the total.
// quad_perm[1,0,3,2]=0xB1, quad_perm[2,3,0,1]=0x4E, row_half_mirror=0x141, row_mirror=0x140.
// v_permlane16_swap of a value with itself: a = {row0, row0, row2, row2}, b = {row1, row1, row3, row3}
// (rows of 16 lanes).  The result elements are copied to scalars before the bit cast:
// __builtin_bit_cast(float, r[1]) on the vector element itself reads element 0 (clang 19 / ROCm 7.2).
struct RowPair { float a, b; };
__device__ __forceinline__ RowPair row_pair(float x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const uint32_t r0 = r[0], r1 = r[1];
    return RowPair{__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1)};
}
__device__ __forceinline__ float row_pair_sum(float x) { const RowPair r = row_pair(x); return r.a + r.b; }
__device__ __forceinline__ float row_pair_max(float x) { const RowPair r = row_pair(x); return fmaxf(r.a, r.b); }

template <int LANES>
__device__ __forceinline__ float group_sum(float x) {
    x += dpp_mov<0xB1>(x);
    x += dpp_mov<0x4E>(x);
    if (LANES >= 8) x += dpp_mov<0x141>(x);
    if (LANES >= 16) x += dpp_mov<0x140>(x);
    if (LANES >= 32) x = row_pair_sum(x);   // across the two 16-lane rows of a 32-lane half
    return x;
}
// Value held by the same lane of the other 32-lane half (lane ^ 32).
__device__ __forceinline__ float other_half(float x) {
    uint32_t u = bitcast<uint32_t>(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // r[0] = {lo, lo}, r[1] = {hi, hi}  (lanes 0-31 | 32-63)
    return (threadIdx.x & 32) ? bitcast<float>(r[0]) : bitcast<float>(r[1]);
}
__device__ __forceinline__ float half_max(float x) {          // max over {lane, lane^32}
    uint32_t u = bitcast<uint32_t>(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(bitcast<float>(r[0]), bitcast<float>(r[1]));
}
__device__ __forceinline__ float half_sum(float x) {
    uint32_t u = bitcast<uint32_t>(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return bitcast<float>(r[0]) + bitcast<float>(r[1]);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

}  // namespace sfa
