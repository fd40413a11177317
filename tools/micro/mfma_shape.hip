// Microbenchmark 5 (round 3): does the prefill kernel's gap program run faster on v_mfma_f32_16x16x32_bf16 than on
// v_mfma_f32_32x32x16_bf16?  MI355X_MICROARCH.md (DVFS give-back, item 7) reports that bare 16x16x32 loops deliver
// 1.12-1.15 x the FLOP/s of 32x32x16 loops at equal cycles per FLOP -- the chip holds a higher clock on that shape -- and
// that cycles therefore do not decide.  Here the two shapes carry the REAL filler mix of prefill_w4_kernel.hip per
// 32768 FLOP of matrix work (one 32x32x16 or two 16x16x32): v_fma (SGPR operand) + v_exp + v_add + every second slot a
// v_cvt_pk + half a v_max3, one ds_read_b128 of a random-data LDS image per slot (consumed eight slots later), one wave
// per SIMD, every CU busy; a fragment is read 16 slots ahead of its MFMA and the wave waits once per batch of eight,
// for a read issued nine slots earlier (as in the kernel: every LDS read >= 8 gaps ahead, one s_waitcnt per batch).
// Reported: cycles per slot, the clock the chip held (s_memtime / s_memrealtime) and the
// wall time per slot -- all three (cdna_hip_programming.md rule 28).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shape.hip -o build/mfma_shape && build/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// SHAPE 32: one 32x32x16 per slot, two accumulation chains of 8 (as the kernel's QK^T side).
// SHAPE 16: two 16x16x32 per slot, eight accumulators of 4 registers (the same 32 x 64 output tile per wave).
// FILL: 0 = bare MFMAs + the LDS read, 1 = + the softmax stages.
template <int SHAPE, int FILL>
__global__ void __launch_bounds__(256, 1) k(const unsigned *in, float *out, unsigned long long *stamps, int iters, float c2) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned *>(smem)[i] = in[i];      // random bf16 pairs
    __syncthreads();
    typedef __attribute__((address_space(3))) const u32x4 lds_u4;
    const __attribute__((address_space(3))) char *lp = (const __attribute__((address_space(3))) char *)smem + 16 * lane + 4096 * (threadIdx.x >> 6);
    bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(in + 4 * threadIdx.x), b1 = *reinterpret_cast<const bf16x8 *>(in + 4 * threadIdx.x + 1024);
    f32x16 s0, s1;
    f32x4 t[8];
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) t[i][r] = 0.f;
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = 0.001f * (lane + i);
    float lsum = 0.f, ms = 0.5f, mx = 0.f;
    unsigned pk[8] = {};
    u32x4 kf[16];
    for (int i = 0; i < 16; ++i) kf[i] = *reinterpret_cast<lds_u4 *>(lp + 1024 * (i & 3));
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 8");
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g == 0 || g == 8) asm volatile("" :: "v"(kf[g + 7]));           // one wait per batch of eight fragments: the youngest
                                                                                // of them was read nine slots ago
            if (SHAPE == 32) {
                if ((g >> 3) & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s1) : "v"(kf[g]), "v"(b1));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s0) : "v"(kf[g]), "v"(b0));
            } else {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(t[(2 * g) & 7]) : "v"(kf[g]), "v"(b0));
            }
            if (FILL) {
                asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(x[(g + 2) & 31]) : "s"(c2), "v"(ms));
                asm volatile("v_exp_f32_e32 %0, %0" : "+v"(x[(g + 1) & 31]));
            }
            if (SHAPE == 16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(t[(2 * g + 1) & 7]) : "v"(kf[g]), "v"(b1));
            // the slot's LDS read: the fragment consumed sixteen slots later (behind the slot's last MFMA that reads the old one)
            *(u32x4 *)&kf[g] = *reinterpret_cast<lds_u4 *>(lp + 1024 * ((g * 5 + it) & 3) + 16384 * (g & 1));
            if (FILL) {
                asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(lsum) : "v"(x[g & 31]));
                if (g & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(g >> 1) & 7]) : "v"(x[(g - 1) & 31]), "v"(x[g & 31]));
                else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(x[(g + 5) & 31]), "v"(x[(g + 6) & 31]));
            }
            FENCE();
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    float acc = lsum + mx;
    for (int r = 0; r < 16; ++r) acc += s0[r] + s1[r];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc += t[i][r];
    for (int i = 0; i < 32; ++i) acc += x[i];
    for (int i = 0; i < 8; ++i) acc += (float)pk[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int FILL>
void run(const char *what, const unsigned *din, float *dout, unsigned long long *dst, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, FILL>), dim3(256), dim3(256), 0, 0, din, dout, dst, iters, 0.1275f);
    hipEventRecord(e0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<SHAPE, FILL>), dim3(256), dim3(256), 0, 0, din, dout, dst, iters, 0.1275f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st(512);
    hipMemcpy(st.data(), dst, 512 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, tick = 0;
    for (int b = 0; b < 256; ++b) { cyc += st[2 * b]; tick += st[2 * b + 1]; }
    const double slots = 16.0 * iters;
    const double flop = 256.0 * 4 * slots * 32768.0 * reps;        // CUs x SIMDs x slots x FLOP per slot x launches
    printf("%-44s %6.1f cycles / slot   clock %.3f GHz   %6.2f ns / slot (in-kernel)   %7.1f TFLOP/s (wall, launches included)\n", what,
           cyc / 256 / slots, cyc / tick / 10.0, tick / 256 * 10.0 / slots, flop / (ms * 1e-3) / 1e12);
}

int main() {
    const int iters = 4000;
    std::vector<unsigned> h(16384 + 4096);
    srand(1);
    for (auto &w : h) {     // two random bf16 of N(0,1)-like magnitude per word
        auto f2b = [](float f) { unsigned u; std::memcpy(&u, &f, 4); return u >> 16; };
        const float a = (rand() / (float)RAND_MAX - 0.5f) * 3.4f, b = (rand() / (float)RAND_MAX - 0.5f) * 3.4f;
        w = f2b(a) | (f2b(b) << 16);
    }
    unsigned *din; float *dout; unsigned long long *dst;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 256 * 256 * 4); hipMalloc(&dst, 512 * 8);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) {       // interleaved rounds in one process (rule 24)
        run<32, 0>("32x32x16, bare + 1 ds_read_b128 / slot", din, dout, dst, iters);
        run<16, 0>("2 x 16x16x32, bare + 1 ds_read_b128 / slot", din, dout, dst, iters);
        run<32, 1>("32x32x16, + softmax stages", din, dout, dst, iters);
        run<16, 1>("2 x 16x16x32, + softmax stages", din, dout, dst, iters);
    }
    return 0;
}
