// Fused attention forward (prefill) for gfx950 (MI355X): the "decoupled" geometry.
// Same math, LDS images and slot-ordered half-step (prefill_core.h) as prefill_kernel.hip, but
//   * workgroup = 128 query rows = 4 wave64s (one per SIMD), TWO workgroups resident per CU;
//   * K/V tiles of 32 keys (one half-step per tile, one 4-wave barrier per tile), K triple- and V
//     double-buffered: 3*32*(2D+16) + 2*32*(2D+64) = 46.6 KB of LDS per workgroup at D = 128.
// Why: in the 256-row kernel the two waves that share a SIMD belong to ONE workgroup and meet at
// one barrier per tile; age arbitration makes the older wave finish each step ~25 % sooner and it
// then idles ~1000 of every ~3650 cycles waiting for the younger one, during which the SIMD runs a
// single, latency-bound wave (in-kernel stamps).  Here the two waves of a SIMD belong
// to DIFFERENT workgroups: neither ever waits for the other, and the four waves that do share a
// barrier all have the same age.  The price is that every K/V tile is staged per 128 rows instead
// of per 256 (twice the staging instructions per MFMA).
//
// Pipeline, per wave (32 query rows), per 32-key tile t:
//     H(t): QK^T(K(t+1)) || max,exp(S(t)),   PV(S(t), V(t)) || lane max(S(t+1))
//   stores of K(t+3) and V(t+1) ride in H(t)'s QK slots, loads of K(t+4) and V(t+2) in its PV slots;
//   barrier(t) follows.  K runs two tiles ahead so the first K fragments of H(t+1) are read during
//   the last slots of H(t).  Buffer safety: K(t+3) overwrites K(t), last read in H(t-1); V(t+1)
//   overwrites V(t-1), last read in H(t-1) -- both before barrier(t-1).
#include <cstdlib>

#include "prefill_core.h"

namespace sfa {

namespace {

using namespace prefill;

constexpr int kBM2 = 128, kBN2 = 32, kThreads2 = 256;

template <class Tr, int D, bool CAUSAL, int PF, int ORD>
__global__ void __launch_bounds__(kThreads2, 2)
prefill_kernel_bm128(const PrefillKernelParams p) {
    using Vec = typename Tr::mfma_vec;
    constexpr int NKS = D / 16, NDB = D / 32, NPV = 2 * NDB;
    constexpr int CPR = D / 8;                      // 16-B chunks per row
    constexpr int NLD = kBN2 * CPR / kThreads2;     // chunks staged per thread per tile (2 or 1)
    constexpr int ROWSTEP = kThreads2 / CPR;
    using L = Lds<D, kBN2, 3, 2>;
    static_assert(NLD >= 1 && NLD <= 2, "staging registers are named kr0, kr1");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const BlockCoord bc = block_coord(p);           // p.nq_tiles counts 128-row tiles here
    if (bc.bh >= p.B * p.Hq) return;
    const int b = bc.bh / p.Hq, h = bc.bh % p.Hq;
    const int hk = h / (p.Hq / p.Hkv);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h2 = lane >> 5;
    const int q0 = bc.qt * kBM2;
    const int wq0 = q0 + 32 * wave;
    const int qrow = wq0 + l31;
    const int coff = p.Sk - p.Sq;

    Vec qf[1][NKS];
    int lim[1];
    {
        const uint16_t *qp = p.q + b * p.qs[0] + h * p.qs[1] + (long long)min(qrow, p.Sq - 1) * p.qs[2] + 8 * h2;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            qf[0][ks] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(qp + 16 * ks));
        lim[0] = CAUSAL ? min(p.Sk - 1, qrow + coff) : p.Sk - 1;
    }

    int kv_end = p.Sk;
    if (CAUSAL) kv_end = min(p.Sk, q0 + kBM2 + coff);
    const int nt = kv_end > 0 ? (kv_end + kBN2 - 1) / kBN2 : 0;
    int ntw = nt;
    if (CAUSAL) ntw = (wq0 + 31 + coff >= 0) ? min(nt, (wq0 + 31 + coff) / kBN2 + 1) : 0;

    // ---- staging ----
    const int st_row = tid / CPR, st_ch = tid % CPR;
    const char *const kg = reinterpret_cast<const char *>(p.k + b * p.ks[0] + hk * p.ks[1]);
    const char *const vg = reinterpret_cast<const char *>(p.v + b * p.vs[0] + hk * p.vs[1]);
    const long long k_tile_bytes = 2ll * kBN2 * p.ks[2], v_tile_bytes = 2ll * kBN2 * p.vs[2];
    const unsigned k_rowb = (unsigned)(2 * p.ks[2]), v_rowb = (unsigned)(2 * p.vs[2]);
    char *const k_w = smem + L::KS * st_row + 16 * st_ch;
    char *const v_w = smem + L::V_BASE + L::VS * st_row + 16 * st_ch;
    uint4 kr0, kr1, vr0, vr1;
    kr0 = kr1 = vr0 = vr1 = make_uint4(0, 0, 0, 0);

#define SFA_LD1(DST, G, ROWB, TB, I, KT)                                                            \
    DST = *reinterpret_cast<const uint4 *>(                                                         \
        (G) + (KT) * (TB) + (unsigned)(st_row + (I) * ROWSTEP) * (ROWB) + 16u * st_ch)
#define SFA_LD1C(DST, G, ROWB, I, KT)                                                               \
    DST = *reinterpret_cast<const uint4 *>(                                                         \
        (G) + (long long)min((KT) * kBN2 + st_row + (I) * ROWSTEP, p.Sk - 1) * (ROWB) + 16u * st_ch)
#define SFA_LOAD_ONE(R0, R1, G, ROWB, TB, KT)                                                       \
    do {                                                                                            \
        const int kt_ = (KT);                                                                       \
        if ((kt_ + 1) * kBN2 <= p.Sk) {                                                             \
            SFA_LD1(R0, G, ROWB, TB, 0, kt_);                                                       \
            if (NLD > 1) SFA_LD1(R1, G, ROWB, TB, 1, kt_);                                          \
        } else {                                                                                    \
            SFA_LD1C(R0, G, ROWB, 0, kt_);                                                          \
            if (NLD > 1) SFA_LD1C(R1, G, ROWB, 1, kt_);                                             \
        }                                                                                           \
    } while (0)
#define SFA_LOAD_K(KT) SFA_LOAD_ONE(kr0, kr1, kg, k_rowb, k_tile_bytes, KT)
#define SFA_LOAD_V(KT) SFA_LOAD_ONE(vr0, vr1, vg, v_rowb, v_tile_bytes, KT)
#define SFA_STORE_ONE(W, RS, BUF, R0, R1)                                                           \
    do {                                                                                            \
        *reinterpret_cast<uint4 *>((W) + (BUF)) = R0;                                               \
        if (NLD > 1) *reinterpret_cast<uint4 *>((W) + (BUF) + ROWSTEP * (RS)) = R1;                  \
    } while (0)
#define SFA_STORE_K(KBUF) SFA_STORE_ONE(k_w, L::KS, KBUF, kr0, kr1)
#define SFA_STORE_V(VBUF) SFA_STORE_ONE(v_w, L::VS, VBUF, vr0, vr1)

    constexpr int NOPS = 2 * NLD;       // op n: even = K chunk n/2, odd = V chunk n/2
    // Branch-free staging loads for the main loop: tiles past the end re-read the last tile (their
    // data is never used), and the one ragged tile (Sk % 32 != 0) swaps in a row-clamped lane offset
    // with a v_cndmask instead of taking a branch.
    const int n_kv_tiles = (p.Sk + kBN2 - 1) / kBN2;
    const int ragged_tile = (p.Sk % kBN2) ? n_kv_tiles - 1 : -1;
    // (named scalars, not arrays: arrays captured by the lambdas below end up in scratch)
    const int row0_ = st_row, row1_ = st_row + ROWSTEP;
    const int last0_ = p.Sk - 1 - (n_kv_tiles - 1) * kBN2;                 // last valid row of the last tile
    const unsigned ow_k0 = (unsigned)row0_ * k_rowb + 16u * st_ch, ow_k1 = (unsigned)row1_ * k_rowb + 16u * st_ch;
    const unsigned ow_v0 = (unsigned)row0_ * v_rowb + 16u * st_ch, ow_v1 = (unsigned)row1_ * v_rowb + 16u * st_ch;
    const unsigned or_k0 = (unsigned)min(row0_, last0_) * k_rowb + 16u * st_ch;
    const unsigned or_k1 = (unsigned)min(row1_, last0_) * k_rowb + 16u * st_ch;
    const unsigned or_v0 = (unsigned)min(row0_, last0_) * v_rowb + 16u * st_ch;
    const unsigned or_v1 = (unsigned)min(row1_, last0_) * v_rowb + 16u * st_ch;
    auto load_op = [&](int n, int kt_k, int kt_v) {
        const int tk = min(kt_k, n_kv_tiles - 1), tv = min(kt_v, n_kv_tiles - 1);      // scalar
        const bool rk = tk == ragged_tile, rv = tv == ragged_tile;
        if (n == 0) kr0 = *reinterpret_cast<const uint4 *>(kg + tk * k_tile_bytes + (rk ? or_k0 : ow_k0));
        if (n == 1) vr0 = *reinterpret_cast<const uint4 *>(vg + tv * v_tile_bytes + (rv ? or_v0 : ow_v0));
        if (NLD > 1 && n == 2) kr1 = *reinterpret_cast<const uint4 *>(kg + tk * k_tile_bytes + (rk ? or_k1 : ow_k1));
        if (NLD > 1 && n == 3) vr1 = *reinterpret_cast<const uint4 *>(vg + tv * v_tile_bytes + (rv ? or_v1 : ow_v1));
    };
    auto store_op = [&](int n, int kbuf, int vbuf) {
        if (n == 0) *reinterpret_cast<uint4 *>(k_w + kbuf) = kr0;
        if (n == 1) *reinterpret_cast<uint4 *>(v_w + vbuf) = vr0;
        if (NLD > 1 && n == 2) *reinterpret_cast<uint4 *>(k_w + kbuf + ROWSTEP * L::KS) = kr1;
        if (NLD > 1 && n == 3) *reinterpret_cast<uint4 *>(v_w + vbuf + ROWSTEP * L::VS) = vr1;
    };

    Acc<D, 1> acc;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.o[0][d][r] = 0.f;
    constexpr bool PS = (ORD == 6);
    acc.msc[0] = PS ? 0.f : ninf();
    acc.lsum[0] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc.cinit[0][r] = 0.f;
    const float c2 = p.scale_log2;
    const char *const k_rd = smem + L::KS * l31 + 16 * h2;
    const char *const v_rd = smem + L::V_BASE + L::VS * (4 * h2 + ((lane & 15) >> 2)) +
                             32 * ((lane >> 4) & 1) + 16 * ((lane & 3) >> 1) + 8 * (lane & 1);
    auto mask_bits = [&](int kbase) -> int {
        return ((CAUSAL && (kbase + 31 > wq0 + coff)) || (kbase + 32 > p.Sk)) ? 1 : 0;
    };

    // ---- prologue: K(0..2), V(0) into LDS, K(3) and V(1) in flight, scores of tile 0 ----
    f32x16 sA[1], sB[1];
    float mxA[1] = {ninf()}, mxB[1] = {ninf()};
#pragma unroll
    for (int r = 0; r < 16; ++r) { sA[0][r] = 0.f; sB[0][r] = 0.f; }
    uint4 kx0, kx1, ky0, ky1;
    kx0 = kx1 = ky0 = ky1 = make_uint4(0, 0, 0, 0);
    if (nt > 0) {
        SFA_LOAD_K(0);
        SFA_LOAD_V(0);
        SFA_LOAD_ONE(kx0, kx1, kg, k_rowb, k_tile_bytes, 1);
        SFA_LOAD_ONE(ky0, ky1, kg, k_rowb, k_tile_bytes, 2);
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[0][ks]));    // see prefill_kernel.hip
    if (PS) {                       // prescaled mode: fold scale * log2(e) into Q (prefill_core.h)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 w = bitcast<u32x4>(qf[0][ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = Tr::pack2(Tr::lo_f32(w[i]) * c2, Tr::hi_f32(w[i]) * c2);
            qf[0][ks] = bitcast<Vec>(w);
        }
    }
    if (nt > 0) {
        SFA_STORE_K(0);
        SFA_STORE_V(0);
        SFA_STORE_ONE(k_w, L::KS, L::KTILE, kx0, kx1);
        SFA_STORE_ONE(k_w, L::KS, 2 * L::KTILE, ky0, ky1);
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NOPS; ++n) load_op(n, 3, 1);
    Vec kpre[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) kpre[i] = bitcast<Vec>(make_uint4(0, 0, 0, 0));
    if (ntw > 0) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            sA[0] = Tr::mfma32(bitcast<Vec>(*reinterpret_cast<const uint4 *>(k_rd + 32 * ks)), qf[0][ks], sA[0]);
#pragma unroll
        for (int i = 0; i < PF; ++i)
            kpre[i] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(k_rd + L::KTILE + 32 * i));
        mxA[0] = lane_rowmax(sA[0]);
        if (PS) {                   // the first tile sets the reference outright
            if (mask_bits(0)) {
                mask_half(sA[0], 0, h2, lim[0]);
                mxA[0] = lane_rowmax(sA[0]);
            }
            const float mx = half_max(mxA[0]);
            const float m0 = (mx == ninf()) ? 0.f : mx;
            acc.msc[0] = m0;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sA[0][r] -= m0; acc.cinit[0][r] = -m0; }
        }
    }
    __syncthreads();        // step 0 stores K(3) into K(0)'s buffer: every wave must be done with tile 0

    int kcur = 0, vcur = 0;             // byte offsets of K(t) and V(t)
#define SFA_NEXT3(X) (((X) == 2 * L::KTILE) ? 0 : (X) + L::KTILE)
#define SFA_ADVANCE()                                                                               \
    do {                                                                                            \
        kcur = SFA_NEXT3(kcur);                                                                     \
        vcur ^= L::VTILE;                                                                           \
    } while (0)
    // non-overlapped staging (TAIL and idle steps): store K(t+3), V(t+1); sync; load K(t+4), V(t+2)
#define SFA_STAGE_AND_SYNC(T)                                                                       \
    do {                                                                                            \
        SFA_STORE_K(kcur);          /* K(t+3) takes K(t)'s buffer */                                \
        SFA_STORE_V(vcur ^ L::VTILE);                                                               \
        __syncthreads();                                                                            \
        _Pragma("unroll") for (int n_ = 0; n_ < NOPS; ++n_) load_op(n_, (T) + 4, (T) + 2);          \
        SFA_FENCE();                                                                                \
    } while (0)

    // One FULL step on tile t: S_OLD = scores of tile t, S_NEW receives the scores of tile t+1.
    // Stores of K(t+3), V(t+1) ride in the QK slots, loads of K(t+4), V(t+2) in the PV slots.
#define SFA_STEP(S_NEW, S_OLD, MX_NEW, MX_OLD)                                                      \
    do {                                                                                            \
        const int k1_ = SFA_NEXT3(kcur), k2_ = SFA_NEXT3(k1_);                                      \
        const int kst_ = kcur, vst_ = vcur ^ L::VTILE;                                              \
        const int tk_ = t + 4, tv_ = t + 2;                                                         \
        auto st_hook = [&](int i) {                                                                 \
            _Pragma("unroll")                                                                       \
            for (int n = (i - 1) * NOPS / (NKS - 1); n < i * NOPS / (NKS - 1); ++n) store_op(n, kst_, vst_); \
        };                                                                                          \
        auto ld_hook = [&](int j) {                                                                 \
            _Pragma("unroll")                                                                       \
            for (int n = j * NOPS / NPV; n < (j + 1) * NOPS / NPV; ++n) load_op(n, tk_, tv_);       \
        };                                                                                          \
        h_block<Tr, D, 1, PF, ORD, 0, 0, true, true, decltype(st_hook), decltype(ld_hook), 0>(      \
            k_rd + k1_, v_rd + vcur, k_rd + k2_, qf, S_NEW, S_OLD, acc, c2, MX_OLD, MX_NEW,         \
            mask_bits((t + (PS ? 1 : 0)) * kBN2), (t + (PS ? 1 : 0)) * kBN2, h2, lim, kpre, st_hook, ld_hook); \
        __syncthreads();                                                                            \
        SFA_ADVANCE();                                                                              \
    } while (0)

    // FULL steps.  One copy of the step; the two score accumulators swap by register moves at the
    // end of it (16 v_mov: two code copies that swap roles by name cost more registers than the
    // 256-VGPR budget at two waves per SIMD has -- the Q fragments got spilled).
    int t = 0;
    for (; t + 1 < ntw; ++t) {
        SFA_STEP(sB, sA, mxB, mxA);
        sA[0] = sB[0];
        mxA[0] = mxB[0];
    }
    // TAIL step: this wave's last tile
    if (t < ntw) {
        h_block<Tr, D, 1, PF, ORD, 0, 0, false, false>(k_rd, v_rd + vcur, k_rd, qf, sB, sA, acc, c2, mxA, mxB,
                                                       mask_bits(t * kBN2), t * kBN2, h2, lim, kpre);
        SFA_STAGE_AND_SYNC(t);
        SFA_ADVANCE();
        ++t;
    }
    for (; t < nt; ++t) {
        SFA_STAGE_AND_SYNC(t);
        SFA_ADVANCE();
    }

    // ---- epilogue ----
    const float ltot = half_sum(acc.lsum[0]);
    const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
    if (qrow < p.Sq) {
        uint16_t *orow = p.o + b * p.os[0] + h * p.os[1] + (long long)qrow * p.os[2];
        store_o_row<Tr, D>(orow, acc.o[0], inv, h2);
        if (p.lse && h2 == 0) {
            const float lse = ltot > 0.f ? (acc.msc[0] + __log2f(ltot)) * kLn2 : ninf();
            p.lse[((long long)b * p.Hq + h) * p.Sq + qrow] = lse;
        }
    }
#undef SFA_STEP
#undef SFA_STAGE_AND_SYNC
#undef SFA_ADVANCE
#undef SFA_NEXT3
#undef SFA_LOAD_K
#undef SFA_LOAD_V
#undef SFA_LOAD_ONE
#undef SFA_STORE_K
#undef SFA_STORE_V
#undef SFA_STORE_ONE
#undef SFA_LD1
#undef SFA_LD1C
}

template <class Tr, int D>
int launch_t(const PrefillKernelParams &p_in, bool causal, int force, hipStream_t stream) {
    PrefillKernelParams p = p_in;
    p.nq_tiles = (p.Sq + kBM2 - 1) / kBM2;              // 128-row q-tiles
    size_t lds = Lds<D, kBN2, 3, 2>::TOTAL;
#ifdef SFA_WITH_VARIANTS      // diagnostic, the A/B library only: one workgroup (one wave per SIMD) per CU
    if (g_knobs.bm128_one_wg.load(std::memory_order_relaxed) > 0) {
        lds = 100 * 1024;
        static DynLdsAttr attr;
        if (const int rc = attr.ensure(reinterpret_cast<const void *>(&prefill_kernel_bm128<Tr, D, true, 2, 6>), (int)lds,
                                       "prefill_kernel_bm128"))
            return rc;
    }
#endif
    dim3 grid(8u * p.bh_per_xcd * p.nq_tiles), block(kThreads2);
    // same policy as the 256-row kernel: exact scale unless the caller opted into the prescaled-Q
    // flavour (SFA_PREFILL_IMPL 21 / 22 force one or the other)
    const bool prescaled = force == 0 ? p.fast_scale != 0 : force == 1;
    if (prescaled) {
        if (causal) hipLaunchKernelGGL((prefill_kernel_bm128<Tr, D, true, 2, 6>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((prefill_kernel_bm128<Tr, D, false, 2, 6>), grid, block, lds, stream, p);
    } else {
        if (causal) hipLaunchKernelGGL((prefill_kernel_bm128<Tr, D, true, 2, 2>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((prefill_kernel_bm128<Tr, D, false, 2, 2>), grid, block, lds, stream, p);
    }
    return check_launch("prefill_kernel_bm128");
}

}  // namespace

int launch_prefill_bm128(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream,
                         int force) {   // force: 0 = by policy, 1 = prescaled, 2 = exact scale
    if (dtype == SFA_DTYPE_FP16) {
        if (head_dim == 128) return launch_t<Fp16, 128>(p, causal, force, stream);
        if (head_dim == 64) return launch_t<Fp16, 64>(p, causal, force, stream);
    } else if (dtype == SFA_DTYPE_BF16) {
        if (head_dim == 128) return launch_t<Bf16, 128>(p, causal, force, stream);
        if (head_dim == 64) return launch_t<Bf16, 64>(p, causal, force, stream);
    } else {
        return fail(SFA_ERR_BAD_DTYPE, "sfa_prefill_fwd: dtype %d is not fp16(0)/bf16(1)", dtype);
    }
    return fail(SFA_ERR_UNSUPPORTED_HEAD_DIM, "sfa_prefill_fwd: head_dim %d not in {64, 128}", head_dim);
}

}  // namespace sfa
