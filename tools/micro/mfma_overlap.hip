// Microbenchmark: how many cycles a lone wave per SIMD spends per v_mfma_f32_32x32x16_bf16 when VALU /
// transcendental / LDS work sits between the MFMAs, by register file of the MFMA operands.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_overlap.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA_VVA(s, k, q) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q))
#define MFMA_VAA(s, k, q) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "a"(k), "a"(q))
#define MFMA_AVV(o, v, p) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(p))
#define MFMA_AAV(o, v, p) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "a"(v), "v"(p))
#define MFMA_AAA(o, v, p) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "a"(v), "a"(p))
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// MODE: 0 S(v) = K(v) Q(a)   1 S(v) = K(a) Q(a)   2 O(a) = V(v) P(v)   3 O(a) = V(a) P(v)   4 O(a) = V(a) P(a)
// NV plain VALU (v_fma) and NX v_exp per PAIR of MFMAs, NL ds_read_b128 per pair (waited for two pairs later).
// SPLIT: the fillers are divided evenly behind EACH of the two MFMAs (M f M f) instead of all behind the pair (M M f f).
template <int MODE, int NV, int NX, int NL, bool SPLIT = false>
__global__ void __launch_bounds__(256, 1) k(const bf16x8 *in, float *out, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<unsigned *>(smem)[i] = i * 2654435761u;
    __syncthreads();
    bf16x8 a0 = in[threadIdx.x], a1 = in[threadIdx.x + 256], b0 = in[threadIdx.x + 512], b1 = in[threadIdx.x + 768];
    f32x16 s0, s1;
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    float x[32];                // filler operands: 32 independent registers used round-robin (no short dependency chains)
    for (int i = 0; i < 32; ++i) x[i] = 0.001f * (lane + i);
    const float c1 = 1.0001f, c2 = 0.5f;
    typedef __attribute__((address_space(3))) const u32x4 lds_u4;
    const __attribute__((address_space(3))) char *lp = (const __attribute__((address_space(3))) char *)smem + 16 * lane;
    u32x4 ld[4] = {};
    asm volatile("s_nop 8");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half == 0) {
                    if (MODE == 0) MFMA_VVA(s0, a0, b0);
                    if (MODE == 1) MFMA_VAA(s0, a0, b0);
                    if (MODE == 2) MFMA_AVV(s0, a0, b0);
                    if (MODE == 3) MFMA_AAV(s0, a0, b0);
                    if (MODE == 4) MFMA_AAA(s0, a0, b0);
                    if (!SPLIT) continue;
                } else {
                    if (MODE == 0) MFMA_VVA(s1, a0, b1);
                    if (MODE == 1) MFMA_VAA(s1, a0, b1);
                    if (MODE == 2) MFMA_AVV(s1, a0, b1);
                    if (MODE == 3) MFMA_AAV(s1, a0, b1);
                    if (MODE == 4) MFMA_AAA(s1, a0, b1);
                }
                const int lo = SPLIT ? half : 0, st = SPLIT ? 2 : 1;        // which of the fillers go here
#pragma unroll
                for (int i = lo; i < NL; i += st) ld[(p + i) & 3] = *reinterpret_cast<lds_u4 *>(lp + 1024 * ((p * NL + i) & 15));
#pragma unroll
                for (int i = lo; i < NV; i += st)       // single-issue v_fma_f32 (asm: hipcc would SLP-pack them into v_pk_fma_f32)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(p * NV + i) % 20]) : "v"(c1), "v"(c2));
#pragma unroll
                for (int i = lo; i < NX; i += st)
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x[20 + (p * NX + i) % 12]));
                if (NL && half == 1) asm volatile("" :: "v"(ld[(p + 2) & 3]));      // consume the read of two pairs ago
                FENCE();
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n s_nop 15" : "+v"(s0), "+v"(s1));
    float acc = 0;
    for (int r = 0; r < 16; ++r) acc += s0[r] + s1[r] + x[r] + x[r + 16];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NV, int NX, int NL, bool SPLIT = false>
void run(const char *name, const bf16x8 *in, float *out, unsigned long long *cyc) {
    const int iters = 200, grid = 256;
    hipLaunchKernelGGL((k<MODE, NV, NX, NL, SPLIT>), dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    hipLaunchKernelGGL((k<MODE, NV, NX, NL, SPLIT>), dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-20s %s  VALU %2d  exp %d  ds_read_b128 %d per MFMA pair: %6.1f cycles per MFMA\n", name,
           SPLIT ? "M f M f" : "M M f f", NV, NX, NL, s / grid / iters / 16.0);
}

int main() {
    bf16x8 *in; float *out; unsigned long long *cyc;
    (void)hipMalloc(&in, 1024 * 16 * 4); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemset(in, 0x3c, 1024 * 16 * 4);
    run<0, 0, 0, 0>("S(v)=K(v)Q(a)", in, out, cyc);
    run<1, 0, 0, 0>("S(v)=K(a)Q(a)", in, out, cyc);
    run<2, 0, 0, 0>("O(a)=V(v)P(v)", in, out, cyc);
    run<0, 4, 0, 0>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 8, 0, 0>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 8, 0, 0, true>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 12, 0, 0>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 12, 0, 0, true>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 16, 0, 0, true>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 8, 2, 0>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 8, 2, 0, true>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 8, 4, 0>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 8, 4, 0, true>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 6, 2, 0, true>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 4, 2, 0, true>("S(v)=K(v)Q(a)", in, out, cyc);
    run<1, 8, 2, 0, true>("S(v)=K(a)Q(a)", in, out, cyc);
    run<2, 8, 2, 0, true>("O(a)=V(v)P(v)", in, out, cyc);
    run<4, 8, 2, 0, true>("O(a)=V(a)P(a)", in, out, cyc);
    run<0, 0, 0, 2>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 0, 0, 2, true>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 0, 0, 4, true>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 6, 2, 2, true>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 8, 2, 2, true>("S(v)=K(v)Q(a)", in, out, cyc);
    run<0, 8, 4, 2>("S(v)=K(v)Q(a)", in, out, cyc);
    return 0;
}
