// Device helpers shared by the decode kernels (decode_kernel.hip, decode_gqa_kernel.hip).
#pragma once
#include "sfa_device.h"
#include "sfa_host.h"

namespace sfa {
namespace decode {

constexpr int kDecodeWaves = 4;

__device__ __forceinline__ float neg_inf() { return -__builtin_huge_valf(); }

template <class Tr>
__device__ __forceinline__ float dot8(const uint4 &a, const uint4 &b) {
    float s = Tr::dot2(a.x, b.x, 0.0f);
    s = Tr::dot2(a.y, b.y, s);
    s = Tr::dot2(a.z, b.z, s);
    s = Tr::dot2(a.w, b.w, s);
    return s;
}

template <class Tr>
__device__ __forceinline__ void unpack8(const uint4 &v, float (&x)[8]) {
    x[0] = Tr::lo_f32(v.x); x[1] = Tr::hi_f32(v.x);
    x[2] = Tr::lo_f32(v.y); x[3] = Tr::hi_f32(v.y);
    x[4] = Tr::lo_f32(v.z); x[5] = Tr::hi_f32(v.z);
    x[6] = Tr::lo_f32(v.w); x[7] = Tr::hi_f32(v.w);
}

template <class Tr>
__device__ __forceinline__ uint4 pack8(const float (&x)[8]) {
    return make_uint4(Tr::pack2(x[0], x[1]), Tr::pack2(x[2], x[3]),
                      Tr::pack2(x[4], x[5]), Tr::pack2(x[6], x[7]));
}

// 16-byte cache-row load; NT = non-temporal (the cache rows are read exactly once per step)
template <bool NT>
__device__ __forceinline__ uint4 ld16(const uint16_t *p) {
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    if (NT) {
        const u32x4v v = __builtin_nontemporal_load(reinterpret_cast<const u32x4v *>(p));
        return make_uint4(v[0], v[1], v[2], v[3]);
    }
    return *reinterpret_cast<const uint4 *>(p);
}

// Which sticky status bit a sequence raises before anything is touched: 1 = seq_len[b] outside
// [0, memory_max_len); 2 = (paged caches) the block_table entry of the page the new token would be
// appended to lies outside the pool -- storing through a substituted page would corrupt another
// sequence.  0 = fine.  Wave-uniform; every workgroup of batch b computes the same value.
template <bool PAGED>
__device__ __forceinline__ int reject_code(const DecodeKernelParams &p, int b, int pos) {
    if (pos < 0 || pos >= p.M) return 1;
    if (PAGED) {
        const int pg = p.block_table[(long long)b * p.table_stride + (pos >> p.page_shift)];
        if ((unsigned)pg >= (unsigned)p.num_pages) return 2;
    }
    return 0;
}

// Running softmax state of one lane group: max (log2 units), sum, and this lane's 8 output dims.
struct Stream {
    float m, l, acc[8];
    __device__ __forceinline__ void init() {
        m = neg_inf(); l = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    }
    // fold another stream (m2, l2, acc2) into this one
    __device__ __forceinline__ void merge(float m2, float l2, const float (&acc2)[8]) {
        const float mn = fmaxf(m, m2);
        const float ms = (mn == neg_inf()) ? 0.f : mn;
        const float a1 = fast_exp2(m - ms), a2 = fast_exp2(m2 - ms);
        l = l * a1 + l2 * a2;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = acc[j] * a1 + acc2[j] * a2;
        m = mn;
    }
};

}  // namespace decode
}  // namespace sfa
