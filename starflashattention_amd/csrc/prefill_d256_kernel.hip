// Fused attention forward (prefill) for head_dim 256 on gfx950 ("hdim 64/256"; the
// reference's head_dim is a runtime field, src/params.h:38).  bf16 / fp16, causal or full, MHA or GQA, any strides.
//
// Twice the head dimension doubles every per-row register array of the 128-wide kernels (O^T is 8 blocks
// of 16 registers, Q^T 16 fragments), so the geometry is the "decoupled" one (prefill_kernel_bm128.hip):
//   * workgroup = 128 query rows = 4 waves, ONE wave per SIMD with the whole 512-entry register file
//     (hipcc keeps the O^T accumulators in the accumulator half), 32 rows per wave;
//   * K/V tiles of 32 keys, register-staged (global loads in flight under the MFMAs of the current tile,
//     ds_write after them), double-buffered in LDS with the padded row images of prefill_core.h
//     (conflict-free ds_read_b128 / ds_read_b64_tr_b16), one barrier per tile;
//   * S^T = K . Q^T and O^T += V^T . P^T with v_mfma_f32_32x32x16 as everywhere else: the query on the lane,
//     in-lane row max / row sum, the exponentiated S^T registers are the B operand of the PV product;
//   * lazy rescale (threshold 2^8), exact-scale numerics only (fast_scale is accepted and ignored).
// This is a plain, compiler-scheduled kernel: the head_dim 256 path is a coverage row, not the headline
// (measured numbers: DESIGN.md).
#include "prefill_core.h"

namespace sfa {

namespace {

using namespace prefill;

constexpr int kD = 256, kRowsD = 128, kKeysD = 32, kThreadsD = 256;

template <class Tr, bool CAUSAL>
__global__ void __launch_bounds__(kThreadsD, 1)
prefill_d256_kernel(const PrefillKernelParams p) {
    using Vec = typename Tr::mfma_vec;
    constexpr int D = kD, NKS = D / 16, NDB = D / 32;
    constexpr int CPR = D / 8;                          // 16-byte chunks per row (32)
    constexpr int NLD = kKeysD * CPR / kThreadsD;       // chunks each thread stages per tile and tensor (4)
    constexpr int ROWSTEP = kThreadsD / CPR;            // 8 rows between a thread's chunks
    using L = Lds<D, kKeysD, 2, 2>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const BlockCoord bc = block_coord(p);               // p.nq_tiles counts 128-row tiles here
    if (bc.bh >= p.B * p.Hq) return;
    const int b = bc.bh / p.Hq, h = bc.bh % p.Hq;
    const int hk = h / (p.Hq / p.Hkv);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h2 = lane >> 5;
    const int q0 = bc.qt * kRowsD, wq0 = q0 + 32 * wave, qrow = wq0 + l31;
    const int coff = p.Sk - p.Sq;                       // causal: key j visible iff j <= i + coff

    int kv_end = p.Sk;
    if (CAUSAL) kv_end = min(p.Sk, q0 + kRowsD + coff);
    const int nt = kv_end > 0 ? (kv_end + kKeysD - 1) / kKeysD : 0;
    int ntw = nt;                                       // tiles this wave computes on (wave-uniform)
    if (CAUSAL) ntw = (wq0 + 31 + coff >= 0) ? min(nt, (wq0 + 31 + coff) / kKeysD + 1) : 0;
    const int lim = CAUSAL ? min(p.Sk - 1, qrow + coff) : p.Sk - 1;

    // Q^T fragments: lane holds Q[row][16 ks + 8 h2 .. +8]
    Vec qf[NKS];
    {
        const uint16_t *qp = p.q + b * p.qs[0] + h * p.qs[1] + (long long)min(qrow, p.Sq - 1) * p.qs[2] + 8 * h2;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(qp + 16 * ks));
    }

    // ---- staging: thread owns chunks (row st_row + i * ROWSTEP, chunk st_ch), i < NLD, of every tile ----
    const int st_row = tid / CPR, st_ch = tid % CPR;
    const char *const kg = reinterpret_cast<const char *>(p.k + b * p.ks[0] + hk * p.ks[1]);
    const char *const vg = reinterpret_cast<const char *>(p.v + b * p.vs[0] + hk * p.vs[1]);
    const long long k_rowb = 2 * p.ks[2], v_rowb = 2 * p.vs[2];
    // (named scalars and macros: an array of staging registers, or a lambda capturing them, ends up in scratch)
    static_assert(NLD == 4, "staging registers are named kr0..kr3 / vr0..vr3");
    uint4 kr0, kr1, kr2, kr3, vr0, vr1, vr2, vr3;
    kr0 = kr1 = kr2 = kr3 = vr0 = vr1 = vr2 = vr3 = make_uint4(0, 0, 0, 0);
#define SFA_D256_LD1(I, T)      /* rows past the sequence end re-read its last row (masked later) */   \
    {                                                                                               \
        const long long row_ = min((T) * kKeysD + st_row + (I) * ROWSTEP, p.Sk - 1);                \
        kr##I = *reinterpret_cast<const uint4 *>(kg + row_ * k_rowb + 16 * st_ch);                  \
        vr##I = *reinterpret_cast<const uint4 *>(vg + row_ * v_rowb + 16 * st_ch);                  \
    }
#define SFA_D256_ST1(I, BUF)                                                                        \
    {                                                                                               \
        *reinterpret_cast<uint4 *>(smem + (BUF) * L::KTILE + L::KS * (st_row + (I) * ROWSTEP) + 16 * st_ch) = kr##I;              \
        *reinterpret_cast<uint4 *>(smem + L::V_BASE + (BUF) * L::VTILE + L::VS * (st_row + (I) * ROWSTEP) + 16 * st_ch) = vr##I;  \
    }
#define SFA_D256_LOAD(T) SFA_D256_LD1(0, T) SFA_D256_LD1(1, T) SFA_D256_LD1(2, T) SFA_D256_LD1(3, T)
#define SFA_D256_STORE(BUF) SFA_D256_ST1(0, BUF) SFA_D256_ST1(1, BUF) SFA_D256_ST1(2, BUF) SFA_D256_ST1(3, BUF)
    // this lane's LDS read bases (everything else is an immediate)
    const char *const k_rd = smem + L::KS * l31 + 16 * h2;
    const char *const v_rd = smem + L::V_BASE + L::VS * (4 * h2 + ((lane & 15) >> 2)) + 32 * ((lane >> 4) & 1) +
                             16 * ((lane & 3) >> 1) + 8 * (lane & 1);

    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float msc = ninf(), lsum = 0.f;
    const float c2 = p.scale_log2;

    if (nt > 0) {
        SFA_D256_LOAD(0)
        SFA_D256_STORE(0)
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) { SFA_D256_LOAD(t + 1) }        // in flight under this tile's MFMAs
        if (t < ntw) {
            const char *kb = k_rd + buf * L::KTILE, *vb = v_rd + buf * L::VTILE;
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                s = Tr::mfma32(bitcast<Vec>(*reinterpret_cast<const uint4 *>(kb + 32 * ks)), qf[ks], s);
            const int kbase = t * kKeysD;
            if ((CAUSAL && kbase + 31 > wq0 + coff) || kbase + kKeysD > p.Sk) mask_half(s, kbase, h2, lim);   // wave-uniform
            const float mxl = lane_rowmax(s);
            if (__any(mxl * c2 > msc + kRescaleThr)) {   // lazy rescale: rare after the first tiles
                const float mnew = fmaxf(msc, half_max(mxl) * c2);
                const float alpha = (mnew == ninf()) ? 1.0f : fast_exp2(msc - mnew);
                msc = mnew;
                lsum *= alpha;
#pragma unroll
                for (int d = 0; d < NDB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            }
            const float msafe = (msc == ninf()) ? 0.f : msc;
            uint32_t pk[8];
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float e0 = fast_exp2(fmaf(s[r], c2, -msafe)), e1 = fast_exp2(fmaf(s[r + 1], c2, -msafe));
                rs += e0 + e1;
                pk[r >> 1] = Tr::pack2(e0, e1);
            }
            lsum += rs;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                uint4 w;
                w.x = pk[4 * k + 0]; w.y = pk[4 * k + 1]; w.z = pk[4 * k + 2]; w.w = pk[4 * k + 3];
#pragma unroll
                for (int d = 0; d < NDB; ++d) {
                    const i16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(vb + L::VS * 16 * k + 64 * d));
                    const i16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(vb + L::VS * (16 * k + 8) + 64 * d));
                    u32x4 av;
                    const u32x2 a_lo = bitcast<u32x2>(t0), a_hi = bitcast<u32x2>(t1);
                    av[0] = a_lo[0]; av[1] = a_lo[1]; av[2] = a_hi[0]; av[3] = a_hi[1];
                    o[d] = Tr::mfma32(bitcast<Vec>(av), bitcast<Vec>(w), o[d]);
                }
            }
        }
        if (t + 1 < nt) { SFA_D256_STORE(buf ^ 1) }     // the other buffer: last read before the previous barrier
        __syncthreads();
    }

#undef SFA_D256_LOAD
#undef SFA_D256_STORE
#undef SFA_D256_LD1
#undef SFA_D256_ST1

    const float ltot = half_sum(lsum);
    const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
    if (qrow < p.Sq) {
        uint16_t *orow = p.o + b * p.os[0] + h * p.os[1] + (long long)qrow * p.os[2];
        store_o_row<Tr, D>(orow, o, inv, h2);
        if (p.lse && h2 == 0)
            p.lse[((long long)b * p.Hq + h) * p.Sq + qrow] = ltot > 0.f ? (msc + __log2f(ltot)) * kLn2 : ninf();
    }
}

template <class Tr>
int launch_d256_t(const PrefillKernelParams &p_in, bool causal, hipStream_t stream) {
    PrefillKernelParams p = p_in;
    p.nq_tiles = (p.Sq + kRowsD - 1) / kRowsD;
    const int lds = Lds<kD, kKeysD, 2, 2>::TOTAL;
    dim3 grid(8u * p.bh_per_xcd * p.nq_tiles), block(kThreadsD);
    static DynLdsAttr attr_c, attr_f;
    if (const int rc = causal ? attr_c.ensure(reinterpret_cast<const void *>(&prefill_d256_kernel<Tr, true>), lds, "prefill_d256_kernel")
                              : attr_f.ensure(reinterpret_cast<const void *>(&prefill_d256_kernel<Tr, false>), lds, "prefill_d256_kernel"))
        return rc;
    if (causal) hipLaunchKernelGGL((prefill_d256_kernel<Tr, true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((prefill_d256_kernel<Tr, false>), grid, block, lds, stream, p);
    return check_launch("prefill_d256_kernel");
}

}  // namespace

int launch_prefill_d256(const PrefillKernelParams &p, int dtype, bool causal, hipStream_t stream) {
    if ((long long)8 * p.bh_per_xcd * ((p.Sq + kRowsD - 1) / kRowsD) > 0x7fffffffll)
        return fail(SFA_ERR_BAD_SHAPE, "sfa_prefill_fwd: grid too large");
    if (dtype == SFA_DTYPE_FP16) return launch_d256_t<Fp16>(p, causal, stream);
    if (dtype == SFA_DTYPE_BF16) return launch_d256_t<Bf16>(p, causal, stream);
    return fail(SFA_ERR_BAD_DTYPE, "sfa_prefill_fwd: dtype %d is not fp16(0)/bf16(1)", dtype);
}

}  // namespace sfa
