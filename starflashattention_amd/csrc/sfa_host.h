// Host-side glue between the C ABI (include/star_flash_attn.h) and the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <atomic>

#include "../../include/star_flash_attn.h"

namespace sfa {

// thread-local error text for sfa_last_error()
void set_error(const char *fmt, ...);
int fail(int status, const char *fmt, ...);

// Workspace layout of sfa_decode: [0,256) status block, then fp32 partial outputs
// [B,H,S,D], then float2 (m, l) [B,H,S].
constexpr size_t kStatusBytes = 256;

struct DecodeKernelParams {
    const uint16_t *qkv;
    const uint16_t *q_bias, *k_bias, *v_bias;
    uint16_t *o;
    const int32_t *seq_len;
    uint16_t *k_cache, *v_cache;
    const uint16_t *cos_tab, *sin_tab;
    float *part_o;          // [B,H,S,D]   un-normalised partial outputs
    float2 *part_ml;        // [B,H,S]     (running max in log2 units, running sum)
    int32_t *status;        // sticky error word
    int B, M, H, L, layer, rot_dim, num_splits;
    int Hkv;                // kv heads (== H unless grouped queries)
    long long qkv_stride;   // elements between batches of qkv
    long long kv_row_stride, kv_head_stride;    // elements between cache rows / heads of one (b, layer)
    const int32_t *block_table;  // paged caches: [B, table_stride] page numbers, else nullptr
    int page_shift, table_stride, num_pages;    // page_size = 1 << page_shift
    long long page_stride;       // elements between pages of a pool
    float scale_log2;       // softmax scale * log2(e)
};

struct PrefillKernelParams {
    const uint16_t *q, *k, *v;
    uint16_t *o;
    float *lse;
    int B, Hq, Hkv, Sq, Sk;
    long long qs[3], ks[3], vs[3], os[3];   // {batch, head, seq} strides (elements)
    float scale_log2;
    int nq_tiles;           // workgroup slots per (batch, head) -- set by each kernel's launcher
    int pairs_per_wg;       // prefill_kernel.hip: balanced q-tile pairs one workgroup walks (1 or 2)
    int bh_per_xcd;         // ceil(B*Hq / 8)
    int fast_scale;         // caller allows the prescaled-Q flavour (only used when lse == nullptr)
};

int launch_decode(const DecodeKernelParams &p, int dtype, int head_dim, hipStream_t stream);
int launch_decode_gqa(const DecodeKernelParams &p, int dtype, int head_dim, hipStream_t stream);
int launch_decode_gqa_mfma(const DecodeKernelParams &p, int dtype, int head_dim, hipStream_t stream);
int launch_prefill(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream);
int launch_prefill_no_keys(const PrefillKernelParams &p, int head_dim, hipStream_t stream);
int launch_rotary_table(void *cos_t, void *sin_t, int max_seq_len, int rot_dim, int dtype, hipStream_t stream);
int launch_fill16(void *arr, uint16_t bits, size_t n, hipStream_t stream);

int check_launch(const char *what);

// Test / A-B knobs, set through sfa_debug_set() (include/star_flash_attn.h) and nothing else: the launch
// paths read no environment variable.  -1 = the library's own choice.
struct DebugKnobs {
    std::atomic<int> prefill_impl{-1};      // prefill_dispatch.hip: which prefill kernel generation
    std::atomic<int> prefill_pairs{-1};     // prefill_kernel.hip: balanced q-tile pairs per workgroup (1 or 2)
    std::atomic<int> decode_nt{-1};         // decode kernels: 0 / 1 force default / non-temporal cache loads
    std::atomic<int> decode_gqa_mfma{-1};   // decode_gqa_kernel.hip: 0 forces the VALU grouped-query kernel
    std::atomic<int> bm128_one_wg{-1};      // prefill_kernel_bm128.hip: 1 = one workgroup per CU (diagnostic)
    std::atomic<int> last_prefill_kernel{-1};   // written by launch_prefill: what ran last (sfa_debug_get)
};
extern DebugKnobs g_knobs;

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: one of these per kernel
// instantiation (a function-local static) remembers which devices of the process already have it.
struct DynLdsAttr {
    std::atomic<unsigned long long> done{0};
    // SFA_OK, or SFA_ERR_LAUNCH with the HIP error text
    int ensure(const void *func, int bytes, const char *what) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return fail(SFA_ERR_LAUNCH, "%s: hipGetDevice failed", what);
        const unsigned long long bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_relaxed) & bit) return SFA_OK;
        const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess)
            return fail(SFA_ERR_LAUNCH, "%s: cannot raise dynamic LDS to %d bytes on device %d: %s", what, bytes, dev,
                        hipGetErrorString(e));
        done.fetch_or(bit, std::memory_order_relaxed);
        return SFA_OK;
    }
};

}  // namespace sfa
