// Microbenchmark 2: one MFMA gap as the 4-wave prefill kernel fills it -- v_fma (VOP3, SGPR operand), v_exp,
// v_add, every second gap a v_cvt_pk, one ds_read_b128 consumed DIST gaps later behind a counted lgkmcnt --
// under variations: length of the dependent accumulator chain, 4- vs 8-byte encodings, the wait itself.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/gap_mix.hip -o build/gap_mix && build/gap_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// CHAIN: consecutive MFMAs on one accumulator before switching to the other (1 = alternate, 8 = q-major QK^T)
// E64: 8-byte encodings for v_exp / v_add.  LDS: 0 none, 1 read + wait DIST gaps later.  NF extra v_fma per gap.
template <int CHAIN, bool E64, int LDS, int DIST, int NF, bool ADDS = true, bool EXP = true>
__global__ void __launch_bounds__(256, 1) k(const bf16x8 *in, float *out, unsigned long long *cyc, int iters, float c2) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<unsigned *>(smem)[i] = 0x3c003c00u;
    __syncthreads();
    bf16x8 b0 = in[threadIdx.x + 512], b1 = in[threadIdx.x + 768];
    f32x16 s0, s1;
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = 0.001f * (lane + i);
    float lsum = 0.f, ms = 0.5f;
    unsigned pk[8] = {};
    typedef __attribute__((address_space(3))) const u32x4 lds_u4;
    const __attribute__((address_space(3))) char *lp = (const __attribute__((address_space(3))) char *)smem + 16 * lane;
    u32x4 kf[8];
    for (int i = 0; i < 8; ++i) kf[i] = *reinterpret_cast<lds_u4 *>(lp + 1024 * i);
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 8");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int which = (g / CHAIN) & 1;
            if (LDS) {      // the fragment read DIST gaps ago must have landed: DIST - 1 younger reads may be in flight
                if (DIST == 1) asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(kf[g & 7]));
                if (DIST == 2) asm volatile("s_waitcnt lgkmcnt(1)" :: "v"(kf[g & 7]));
                if (DIST == 4) asm volatile("s_waitcnt lgkmcnt(3)" :: "v"(kf[g & 7]));
            }
            if (which == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s0) : "v"(kf[g & 7]), "a"(b0));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s1) : "v"(kf[g & 7]), "a"(b1));
            if (LDS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[(g + DIST) & 7]) : "v"(lp), "n"(1024 * ((g * 5) & 15)));
            asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(x[(g + 2) & 31]) : "s"(c2), "v"(ms));
#pragma unroll
            for (int i = 0; i < NF; ++i) asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(x[(g + 9 + i) & 31]) : "s"(c2), "v"(ms));
            if (EXP) {
                if (E64) asm volatile("v_exp_f32_e64 %0, %0" : "+v"(x[(g + 1) & 31]));
                else asm volatile("v_exp_f32_e32 %0, %0" : "+v"(x[(g + 1) & 31]));
            }
            if (ADDS) {
                if (E64) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(lsum) : "v"(x[g & 31]));
                else asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(lsum) : "v"(x[g & 31]));
                if (g & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(g >> 1) & 7]) : "v"(x[(g - 1) & 31]), "v"(x[g & 31]));
            }
            FENCE();
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15" : "+v"(s0), "+v"(s1));
    float acc = lsum;
    for (int r = 0; r < 16; ++r) acc += s0[r] + s1[r] + x[r] + x[r + 16];
    for (int i = 0; i < 8; ++i) acc += __builtin_bit_cast(float, pk[i]) + __builtin_bit_cast(float, kf[i][0]);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAIN, bool E64, int LDS, int DIST, int NF, bool ADDS = true, bool EXP = true>
void run(const char *what, const bf16x8 *in, float *out, unsigned long long *cyc) {
    const int iters = 200, grid = 256;
    for (int r = 0; r < 2; ++r)
        hipLaunchKernelGGL((k<CHAIN, E64, LDS, DIST, NF, ADDS, EXP>), dim3(grid), dim3(256), 0, 0, in, out, cyc, iters, 0.1275f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("chain %d  %s  lds %d dist %d  extra fma %d  adds %d exp %d : %6.1f cycles per MFMA   (%s)\n", CHAIN, E64 ? "e64" : "e32", LDS, DIST,
           NF, (int)ADDS, (int)EXP, s / grid / iters / 16.0, what);
}

int main() {
    bf16x8 *in; float *out; unsigned long long *cyc;
    (void)hipMalloc(&in, 1024 * 16 * 4); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemset(in, 0x3c, 1024 * 16 * 4);
    run<1, false, 0, 2, 0, false, false>("fma only", in, out, cyc);
    run<1, false, 0, 2, 0, false, true>("fma + exp", in, out, cyc);
    run<1, false, 0, 2, 0>("fma + exp + add + cvt/2", in, out, cyc);
    run<8, false, 0, 2, 0>("same, 8-long accumulator chains", in, out, cyc);
    run<1, true, 0, 2, 0>("same, 8-byte encodings", in, out, cyc);
    run<8, true, 0, 2, 0>("8-chains, 8-byte encodings", in, out, cyc);
    run<8, false, 1, 2, 0>("+ ds_read_b128, waited 2 gaps later", in, out, cyc);
    run<8, false, 1, 4, 0>("+ ds_read_b128, waited 4 gaps later", in, out, cyc);
    run<8, false, 1, 1, 0>("+ ds_read_b128, waited 1 gap later", in, out, cyc);
    run<8, false, 1, 4, 1>("dist 4, one more fma per gap", in, out, cyc);
    run<8, false, 1, 4, 2>("dist 4, two more fma per gap", in, out, cyc);
    run<8, true, 1, 4, 0>("dist 4, 8-byte encodings", in, out, cyc);
    return 0;
}
