// Small helpers the reference's C++ harness calls (src/flash_attn.h:9-11):
//   rotary cos/sin LUT   (intended semantics of rotary_table_kernel, flash_attn.cu:512-538)
//   16-bit array fill    (init_half_array_kernel, flash_attn.cu:493-510)
#include "sfa_device.h"
#include "sfa_host.h"

namespace sfa {
namespace {

// table[pos * (rot/2) + j] = cos/sin(pos * 10000^(-2j/rot)), fp32 math, rounded to 16 bit.
// (The reference strides rows by `rot` and overruns its rot/2 * max_seq_len allocation.)
template <class Tr>
__global__ void __launch_bounds__(256)
rotary_table_kernel(uint16_t *cos_t, uint16_t *sin_t, int max_seq_len, int half_rot, int rot) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)max_seq_len * half_rot) return;
    const int pos = (int)(i / half_rot), j = (int)(i % half_rot);
    const float inv_freq = 1.0f / powf(10000.0f, (float)(2 * j) / (float)rot);
    float s, c;
    sincosf((float)pos * inv_freq, &s, &c);
    cos_t[i] = Tr::from_f32(c);
    sin_t[i] = Tr::from_f32(s);
}

// 8 elements (16 B) per thread, grid-stride; scalar tail.
__global__ void __launch_bounds__(256)
fill16_kernel(uint16_t *arr, uint16_t bits, size_t n) {
    const size_t nvec = n / 8;
    const uint32_t w = (uint32_t)bits | ((uint32_t)bits << 16);
    const uint4 v = make_uint4(w, w, w, w);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(arr) & 15) == 0) {
        for (size_t i = gid; i < nvec; i += stride) reinterpret_cast<uint4 *>(arr)[i] = v;
        for (size_t i = nvec * 8 + gid; i < n; i += stride) arr[i] = bits;
    } else {
        for (size_t i = gid; i < n; i += stride) arr[i] = bits;
    }
}

// Prefill with no keys at all: every row is empty -> O = 0, lse = -inf (include/star_flash_attn.h).
// One thread per 16 bytes of O; rows are addressed through the caller's strides.
__global__ void __launch_bounds__(256)
prefill_no_keys_kernel(const PrefillKernelParams p, int head_dim) {
    const int cpr = head_dim / 8;
    const long long n = (long long)p.B * p.Hq * p.Sq * cpr;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ch = (int)(i % cpr);
    const long long row = i / cpr;
    const int s = (int)(row % p.Sq);
    const long long bh = row / p.Sq;
    const int h = (int)(bh % p.Hq);
    const long long b = bh / p.Hq;
    *reinterpret_cast<uint4 *>(p.o + b * p.os[0] + h * p.os[1] + s * p.os[2] + 8 * ch) = make_uint4(0, 0, 0, 0);
    if (p.lse && ch == 0) p.lse[row] = -__builtin_huge_valf();
}

}  // namespace

int launch_prefill_no_keys(const PrefillKernelParams &p, int head_dim, hipStream_t stream) {
    const long long n = (long long)p.B * p.Hq * p.Sq * (head_dim / 8);
    if (n == 0) return SFA_OK;
    hipLaunchKernelGGL(prefill_no_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, head_dim);
    return check_launch("prefill_no_keys_kernel");
}

int launch_rotary_table(void *cos_t, void *sin_t, int max_seq_len, int rot_dim, int dtype, hipStream_t stream) {
    const int half = rot_dim / 2;
    const long long n = (long long)max_seq_len * half;
    if (n == 0) return SFA_OK;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == SFA_DTYPE_FP16) {
        hipLaunchKernelGGL(rotary_table_kernel<Fp16>, grid, block, 0, stream, (uint16_t *)cos_t,
                           (uint16_t *)sin_t, max_seq_len, half, rot_dim);
    } else if (dtype == SFA_DTYPE_BF16) {
        hipLaunchKernelGGL(rotary_table_kernel<Bf16>, grid, block, 0, stream, (uint16_t *)cos_t,
                           (uint16_t *)sin_t, max_seq_len, half, rot_dim);
    } else {
        return fail(SFA_ERR_BAD_DTYPE, "sfa_compute_rotary_table: dtype %d", dtype);
    }
    return check_launch("rotary_table_kernel");
}

int launch_fill16(void *arr, uint16_t bits, size_t n, hipStream_t stream) {
    if (n == 0) return SFA_OK;
    size_t blocks = (n / 8 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;          // 256 CUs x 8, grid-stride the rest
    hipLaunchKernelGGL(fill16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (uint16_t *)arr, bits, n);
    return check_launch("fill16_kernel");
}

}  // namespace sfa
