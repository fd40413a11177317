// Microbenchmark 4: what a packed-fp32 VALU op (v_pk_fma_f32 / v_pk_add_f32) or a dot op (v_dot2c_f32_bf16) costs in
// an MFMA gap, against the scalar ops it replaces.  The 4-wave prefill kernel was rebuilt on packed F / A stages
// (96 instead of 128 VALU ops per half-step) and ran 14 % SLOWER, with the row sum by v_dot2c 5 % slower; this
// isolates why: those ops wait for the matrix pipe (53 cycles per gap against 33).  profiles/r02_power_clock.txt
//   hipcc -O3 --offload-arch=gfx950 tools/micro/pk_f32.hip -o build/pk_f32 && build/pk_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// MODE 0: per gap  v_fma + v_exp + v_add (+ v_cvt_pk every 2nd)        (the shipped gap program)
//      1: even gap v_pk_fma + v_exp ; odd gap v_exp + v_pk_add + v_cvt_pk
//      2: per gap  v_fma only           3: per gap v_pk_fma only (twice the elements)
//      4: per gap  2 x v_fma            5: v_pk_fma with a VGPR pair instead of the SGPR pair operand
//      6: per gap  v_pk_add only        7: per gap v_pk_mul only
//      8: per gap  v_dot2c_f32_bf16     9: the scalar gap with the row sum by v_dot2c on the packed pair
// MFMA 0: no MFMAs at all (raw issue cost of the fillers)
template <int MODE, bool MFMA>
__global__ void __launch_bounds__(256, 1) k(const bf16x8 *in, float *out, unsigned long long *cyc, int iters, float c2) {
    const int lane = threadIdx.x & 63;
    bf16x8 a0 = in[threadIdx.x], b0 = in[threadIdx.x + 512];
    f32x16 s0, s1;
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    f32x2 x[16];
    for (int i = 0; i < 16; ++i) x[i] = f32x2{0.001f * (lane + i), 0.002f * (lane + i)};
    f32x2 lsum = {0.f, 0.f}, ms = {0.5f, 0.5f}, c2v = {c2, c2};
    const f32x2 c2p = {c2, c2};
    unsigned pk[8] = {};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (MFMA) {
                if ((g >> 3) & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s1) : "v"(a0), "a"(b0));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s0) : "v"(a0), "a"(b0));
            }
#define EL x[(g + 3) & 15][g & 1]                                 /* scalar view of "element g" */
            if (MODE == 0) {
                asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(x[(g + 5) & 15][g & 1]) : "s"(c2), "v"(ms[0]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(EL));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(lsum[0]) : "v"(x[(g + 1) & 15][g & 1]));
                if (g & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(g >> 1) & 7]) : "v"(x[g & 15][0]), "v"(x[g & 15][1]));
            } else if (MODE == 1) {
                if (!(g & 1)) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(x[(g + 5) & 15]) : "s"(c2p), "v"(ms));
                asm volatile("v_exp_f32 %0, %0" : "+v"(EL));
                if (g & 1) {
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(lsum) : "v"(x[(g + 1) & 15]));
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(g >> 1) & 7]) : "v"(x[g & 15][0]), "v"(x[g & 15][1]));
                }
            } else if (MODE == 2) {
                asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(x[(g + 5) & 15][g & 1]) : "s"(c2), "v"(ms[0]));
            } else if (MODE == 3) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(x[(g + 5) & 15]) : "s"(c2p), "v"(ms));
            } else if (MODE == 4) {
                asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(x[(g + 5) & 15][0]) : "s"(c2), "v"(ms[0]));
                asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(x[(g + 5) & 15][1]) : "s"(c2), "v"(ms[0]));
            } else if (MODE == 5) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(x[(g + 5) & 15]) : "v"(c2v), "v"(ms));
            } else if (MODE == 6) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[(g + 5) & 15]) : "v"(ms));
            } else if (MODE == 7) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[(g + 5) & 15]) : "v"(ms));
            } else if (MODE == 8) {
                asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(lsum[0]) : "s"(0x3f803f80u), "v"(pk[g & 7]));
            } else if (MODE == 9) {         // the shipped gap with the row sum taken from the PACKED pair: one op per two elements
                asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(x[(g + 5) & 15][g & 1]) : "s"(c2), "v"(ms[0]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(EL));
                if (g & 1) {
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(g >> 1) & 7]) : "v"(x[g & 15][0]), "v"(x[g & 15][1]));
                    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(lsum[0]) : "s"(0x3f803f80u), "v"(pk[((g >> 1) + 7) & 7]));
                }
            } else if (MODE == 10) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(EL));
            } else if (MODE == 11) {
                asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(EL));
            } else if (MODE == 12) {
                asm volatile("v_exp_f16 %0, %0" : "+v"(EL));
            } else if (MODE == 13) {        // the shipped gap on the legacy exponential
                asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(x[(g + 5) & 15][g & 1]) : "s"(c2), "v"(ms[0]));
                asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(EL));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(lsum[0]) : "v"(x[(g + 1) & 15][g & 1]));
                if (g & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(g >> 1) & 7]) : "v"(x[g & 15][0]), "v"(x[g & 15][1]));
            } else if (MODE == 14) {        // 16-bit exponential: fma -> f16 (v_fma_mixlo_f16), v_exp_f16, row sum by v_fma_mix_f32
                asm volatile("v_fma_mixlo_f16 %0, %0, %1, -%2" : "+v"(x[(g + 5) & 15][g & 1]) : "s"(c2), "v"(ms[0]));
                asm volatile("v_exp_f16 %0, %0" : "+v"(EL));
                asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(lsum[0]) : "v"(x[(g + 1) & 15][g & 1]));
            } else if (MODE == 15) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(EL));
                asm volatile("v_exp_f32 %0, %0" : "+v"(x[(g + 7) & 15][g & 1]));
            } else if (MODE == 16) {
                asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(EL));
                asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(x[(g + 7) & 15][g & 1]));
            } else if (MODE == 17) {
                asm volatile("v_exp_f16 %0, %0" : "+v"(EL));
                asm volatile("v_exp_f16 %0, %0" : "+v"(x[(g + 7) & 15][g & 1]));
            }
            FENCE();
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n s_nop 15" : "+v"(s0), "+v"(s1));
    float acc = lsum[0] + lsum[1] + c2v[0];
    for (int r = 0; r < 16; ++r) acc += s0[r] + s1[r] + x[r][0] + x[r][1];
    for (int i = 0; i < 8; ++i) acc += __builtin_bit_cast(float, pk[i]);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, bool MFMA>
void run(const char *what, const bf16x8 *in, float *out, unsigned long long *cyc) {
    const int iters = 200, grid = 256;
    for (int r = 0; r < 2; ++r)
        hipLaunchKernelGGL((k<MODE, MFMA>), dim3(grid), dim3(256), 0, 0, in, out, cyc, iters, 0.1275f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-62s %s : %6.1f cycles per gap\n", what, MFMA ? "behind an MFMA" : "no MFMA       ", s / grid / iters / 16.0);
}

int main() {
    bf16x8 *in; float *out; unsigned long long *cyc;
    (void)hipMalloc(&in, 1024 * 16 * 4); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemset(in, 0x3c, 1024 * 16 * 4);
#define BOTH(M, what) run<M, true>(what, in, out, cyc); run<M, false>(what, in, out, cyc);
    BOTH(2, "1 v_fma_f32")
    BOTH(4, "2 v_fma_f32")
    BOTH(3, "1 v_pk_fma_f32 (SGPR-pair operand)")
    BOTH(5, "1 v_pk_fma_f32 (VGPR operands only)")
    BOTH(6, "1 v_pk_add_f32")
    BOTH(7, "1 v_pk_mul_f32")
    BOTH(0, "scalar gap: fma + exp + add (+ cvt_pk / 2)")
    BOTH(1, "packed gap: even pk_fma + exp, odd exp + pk_add + cvt_pk")
    BOTH(8, "1 v_dot2c_f32_bf16 (SGPR ones, packed pair)")
    BOTH(9, "dot2 gap: fma + exp (+ cvt_pk + dot2c / 2)")
    BOTH(10, "1 v_exp_f32")
    BOTH(15, "2 v_exp_f32")
    BOTH(11, "1 v_exp_legacy_f32")
    BOTH(16, "2 v_exp_legacy_f32")
    BOTH(12, "1 v_exp_f16")
    BOTH(17, "2 v_exp_f16")
    BOTH(13, "scalar gap on v_exp_legacy_f32")
    BOTH(14, "16-bit gap: v_fma_mixlo_f16 + v_exp_f16 + v_fma_mix_f32")
    return 0;
}
