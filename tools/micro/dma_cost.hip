// Microbenchmark 3: what one LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB per wave) costs a lone wave per
// SIMD that is otherwise issuing MFMAs with the prefill kernel's filler mix, against the same bytes fetched
// into VGPRs (global_load_dwordx4) and stored with ds_write_b128.  Source: a 4 MiB buffer all workgroups share
// (L2-resident), EVERY workgroup streaming.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dma_cost.hip -o build/dma_cost && build/dma_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// MODE 0: no staging.  1: LDS-DMA, PER pieces per 32 gaps.  2: global_load_dwordx4 + ds_write_b128, PER per 32 gaps.
template <int MODE, int PER, bool FILL, bool RD = false, bool SWZ = false>
__global__ void __launch_bounds__(256, 1) k(const char *src, float *out, unsigned long long *cyc, int iters, float c2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a0, b0;
    for (int i = 0; i < 8; ++i) { a0[i] = (__bf16)1.0f; b0[i] = (__bf16)0.5f; }
    f32x16 s0, s1;
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = 0.001f * (lane + i);
    float lsum = 0.f, ms = 0.5f;
    typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
    const unsigned long long base = (unsigned long long)(uintptr_t)src;
    u32x4s srd = {(unsigned)base, (unsigned)(base >> 32) & 0xffffu, 4u << 20, 0x00020000u};
    // SWZ: the prefill kernel's source pattern -- 8 rows x 128 B per piece, row stride 256 B, chunks XOR-swizzled
    const int r8 = (lane >> 2) & 7, dslot = lane & 3, dsub = lane >> 5;
    const unsigned voff = SWZ ? (unsigned)r8 * 256u + 64u * dsub + 16u * (dslot ^ (r8 >> 2)) : 16u * lane;
    typedef __attribute__((address_space(3))) const u32x4 lds_u4;
    const __attribute__((address_space(3))) char *lp = (const __attribute__((address_space(3))) char *)smem + 16 * lane;
    u32x4 frag[8] = {};
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem + 16384u * wave;
    u32x4 stage[2] = {};
    unsigned pos = (blockIdx.x * 4 + wave) * 1024u;
    asm volatile("s_nop 8");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            if (g & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s1) : "v"(a0), "a"(b0));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s0) : "v"(a0), "a"(b0));
            if (FILL) {
                asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(x[(g + 2) & 31]) : "s"(c2), "v"(ms));
                asm volatile("v_exp_f32_e32 %0, %0" : "+v"(x[(g + 1) & 31]));
                asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(lsum) : "v"(x[g & 31]));
            }
            if (RD) {       // one fragment read per gap, consumed (batched wait) eight gaps later
                if ((g & 7) == 0) asm volatile("" :: "v"(frag[7]));
                frag[g & 7] = *reinterpret_cast<lds_u4 *>(lp + 1024 * ((g * 5) & 63));
            }
            if (MODE == 1 && g < PER) {
                const unsigned soff = (pos + 1024u * g) & ((4u << 20) - 1);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(lds0 + 1024u * (g & 15)), "v"(voff), "s"(srd), "s"(soff) : "memory");
            }
            if (MODE == 2 && g < PER) {
                const unsigned soff = (pos + 1024u * g) & ((4u << 20) - 1);
                if (g >= 2) {       // the load of two gaps ago: wait for it alone, store it
                    asm volatile("s_waitcnt vmcnt(1)\n\tds_write_b128 %0, %1" :: "v"(lds0 + 16u * lane + 1024u * (g & 15)), "v"(stage[g & 1]) : "memory");
                }
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(stage[g & 1]) : "v"(voff), "s"(srd), "s"(soff) : "memory");
            }
            FENCE();
        }
        pos += 1024u * 32 * 7;
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
        if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" :: "v"(stage[0]), "v"(stage[1]) : "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15" : "+v"(s0), "+v"(s1));
    float acc = lsum + *(float *)(smem + 4 * threadIdx.x);
    for (int r = 0; r < 16; ++r) acc += s0[r] + s1[r] + x[r] + x[r + 16];
    acc += __builtin_bit_cast(float, stage[0][0]) + __builtin_bit_cast(float, stage[1][0]);
    for (int i = 0; i < 8; ++i) acc += __builtin_bit_cast(float, frag[i][0]);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int PER, bool FILL, bool RD = false, bool SWZ = false>
void run(const char *what, const char *src, float *out, unsigned long long *cyc) {
    const int iters = 200, grid = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE, PER, FILL, RD, SWZ>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int r = 0; r < 2; ++r)
        hipLaunchKernelGGL((k<MODE, PER, FILL, RD, SWZ>), dim3(grid), dim3(256), 65536, 0, src, out, cyc, iters, 0.1275f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    const double per = s / grid / iters / 32.0;
    printf("%-28s %2d pieces per 32 MFMAs, fillers %d, lds reads %d, swizzled source %d: %6.1f cycles per MFMA", what, PER, (int)FILL,
           (int)RD, (int)SWZ, per);
    if (PER) printf("   (+%.0f cycles per piece)", (per - (FILL ? 34.0 : 33.3)) * 32.0 / PER);
    printf("\n");
}

int main() {
    char *src; float *out; unsigned long long *cyc;
    (void)hipMalloc(&src, 4 << 20); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemset(src, 0x3c, 4 << 20);
    run<0, 0, false>("no staging", src, out, cyc);
    run<0, 0, true>("no staging", src, out, cyc);
    run<1, 4, false>("LDS-DMA", src, out, cyc);
    run<1, 4, true>("LDS-DMA", src, out, cyc);
    run<1, 8, true>("LDS-DMA", src, out, cyc);
    run<1, 16, true>("LDS-DMA", src, out, cyc);
    run<0, 0, true, true>("no staging", src, out, cyc);
    run<1, 4, true, true>("LDS-DMA", src, out, cyc);
    run<1, 8, true, true>("LDS-DMA", src, out, cyc);
    run<1, 4, true, false, true>("LDS-DMA", src, out, cyc);
    run<1, 4, true, true, true>("LDS-DMA", src, out, cyc);
    run<1, 8, true, true, true>("LDS-DMA", src, out, cyc);
    run<2, 4, false>("global_load + ds_write_b128", src, out, cyc);
    run<2, 4, true>("global_load + ds_write_b128", src, out, cyc);
    run<2, 8, true>("global_load + ds_write_b128", src, out, cyc);
    run<2, 16, true>("global_load + ds_write_b128", src, out, cyc);
    return 0;
}
