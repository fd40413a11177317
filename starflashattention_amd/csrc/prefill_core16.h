// Pipelined half-step of the prefill kernel on v_mfma_f32_16x16x32_{bf16,f16} (prefill_kernel16.hip).
// Same slot structure as h_block in prefill_core.h -- one LDS fragment per slot, the MFMAs it feeds,
// the fragment read PF slots ahead, a slice of the softmax -- but every slot issues TWO 16x16x32
// MFMAs (one per 16-query block of the wave) where the 32x32x16 form issues one: the same MFMA-pipe
// cycles per FLOP, and the chip holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS
// give-back item 7).
//
// Fragment maps (lane l: c = l & 15, g = l >> 4), wave = 32 query rows = 2 blocks of 16:
//   QK^T:  S^T[key][query] = K . Q^T      A = K rows [16 keys][32 d]: lane holds K[16kt + c][32ks + 8g ..+8]
//                                         B = Q^T:                   lane holds Q[16qb + c][32ks + 8g ..+8]
//          accumulator s[qb][kt] (4 regs): element r = key 16kt + 4g + r of query 16qb + c
//   PV:    O^T[d][query] += V^T . P^T     B = P^T [32 keys][16 queries]: lane's k-slots 8g + j hold
//                                             j < 4: key 4g + j        = s[qb][0][j]
//                                             j >= 4: key 16 + 4g + j-4 = s[qb][1][j-4]
//                                         so the exponentiated accumulators, packed, ARE the B operand;
//                                         A = V^T [16 d][32 keys] with the same key order: two
//                                             ds_read_b64_tr_b16 (rows 4g.. and 16+4g.. of the half-tile)
//          accumulator o[qb][dt] (4 regs): element r = d 16dt + 4g + r of query 16qb + c
// The query sits on the lane in both accumulators; a query's keys are split over the four 16-lane
// groups, so row max / row sum finish with v_permlane32_swap + v_permlane16_swap.
#pragma once
#include "prefill_core.h"

namespace sfa {
namespace prefill {

//   K rows: 2*D + 16 bytes (ds_read_b128 of this operand is 2-way bank conflicted on any padded
//           image -- the two half-groups of a 16-lane service group read chunks g and g+1)
//   V rows: 2*D + 32 bytes: a 32-lane half of ds_read_b64_tr_b16 reads 8 consecutive rows x 32 bytes,
//           32 r mod 256 (D=128) / 160 r mod 256 (D=64) are 8 distinct multiples of 32: conflict-free
template <int D, int BN = kBN, int NKB = 3, int NVB = 3> struct Lds16 {
    static constexpr int KS = 2 * D + 16;
    static constexpr int VS = 2 * D + 32;
    static constexpr int KTILE = BN * KS;
    static constexpr int VTILE = BN * VS;
    static constexpr int V_BASE = NKB * KTILE;
    static constexpr int TOTAL = NKB * KTILE + NVB * VTILE;
    static_assert(NVB * VTILE < 65536 && NKB * KTILE < 65536, "ds immediates are 16 bit");
};

template <class Tr> struct Mfma16;
template <> struct Mfma16<Bf16> {
    static __device__ __forceinline__ f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma16<Fp16> {
    static __device__ __forceinline__ f32x4 run(f16x8 a, f16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

// max / sum over the four lanes {c, c+16, c+32, c+48} that share a query
__device__ __forceinline__ float quad_max(float x) { return row_pair_max(half_max(x)); }
__device__ __forceinline__ float quad_sum(float x) { return row_pair_sum(half_sum(x)); }

// Per-wave online-softmax state: two 16-query blocks.
//   EXACT = false: Q carries scale*log2(e) (rounded to 16 bit), scores are exp2 arguments, msc in log2 units
//   EXACT = true : Q as given, scores and msc in raw units, one v_mul by scale*log2(e) in front of v_exp
template <int D>
struct Acc16 {
    f32x4 o[2][D / 16];
    float msc[2];           // reference max of the exponentials
    float lsum[2];          // this lane's share of the running row sum
    f32x4 cinit[2];         // -msc in all 4 registers: C operand of the first QK^T MFMA of a half-step
};

// key of s[kt][r] = kbase + 16kt + 4g + r
__device__ __forceinline__ void mask_half16(f32x4 (&s)[2], int kbase, int g, int lim) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (kbase + 16 * kt + 4 * g + r > lim) s[kt][r] = ninf();
}
__device__ __forceinline__ float lane_rowmax16(const f32x4 (&s)[2]) {
    return max3(max3(s[0][0], s[0][1], s[0][2]), max3(s[0][3], s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3]));
}

// Finish the row max of freshly computed scores of both query blocks (already relative to acc.msc).
// Lazy rescale, one wave-uniform branch for both blocks.
template <int D, bool EXACT>
__device__ __forceinline__ void finish16(f32x4 (&s)[2][2], Acc16<D> &acc, float c2, float mxl0, float mxl1,
                                         int masked, int kbase, int g, const int (&lim)[2]) {
    if (masked) {                                       // wave-uniform, diagonal / ragged tiles only
        mask_half16(s[0], kbase, g, lim[0]);
        mask_half16(s[1], kbase, g, lim[1]);
        mxl0 = lane_rowmax16(s[0]);
        mxl1 = lane_rowmax16(s[1]);
    }
    const float mx0 = quad_max(mxl0), mx1 = quad_max(mxl1);
    const float thr = EXACT ? kRescaleThr / c2 : kRescaleThr;
    if (__any(fmaxf(mx0, mx1) > thr)) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float mx = qb ? mx1 : mx0;
            const float d = (mx > thr) ? mx : 0.f;      // rows that did not rise keep their reference
            const float alpha = fast_exp2(EXACT ? -d * c2 : -d);
            acc.msc[qb] += d;
            acc.lsum[qb] *= alpha;
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc.o[qb][dt][r] *= alpha;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[qb][kt][r] -= d;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.cinit[qb][r] = -acc.msc[qb];
        }
    }
}

// One pipelined half-step:
//   sN[qb][kt] <- scores of K rows [32*HN, +32) of the tile at kb       (DO_QK; 2*D/16 MFMAs)
//   sO[qb][kt]  = finished scores of keys [32*HO, +32) of the tile whose V is at vb: exponentiated in
//                 place, packed to 16 bit, O^T += V^T . P^T               (2*D/16 MFMAs)
// then the row max of sN is finished (mask_n / kbase_n describe sN).
// kb / vb / kb_pref include this lane's read base.
template <class Tr, int D, bool EXACT, int PF, int HN, int HO, bool DO_QK, bool PREF, class QkHook = NoHook,
          class PvHook = NoHook, int PH = 1 - HN>
__device__ __forceinline__ void h_block16(const char *kb, const char *vb, const char *kb_pref,
                                          const typename Tr::mfma_vec (&qf)[2][D / 32], f32x4 (&sN)[2][2],
                                          f32x4 (&sO)[2][2], Acc16<D> &acc, float c2, int mask_n, int kbase_n,
                                          int g, const int (&lim)[2], typename Tr::mfma_vec (&kpre)[PF],
                                          const QkHook &qk_hook = QkHook(), const PvHook &pv_hook = PvHook()) {
    using Vec = typename Tr::mfma_vec;
    using L = Lds16<D>;
    constexpr int NKS = D / 32;             // k-steps of one QK^T accumulator
    constexpr int NSL = 2 * NKS;            // QK slots: (ks, kt), one K fragment and two MFMAs each
    constexpr int NPV = D / 16;             // PV slots: one V^T fragment (d tile) and two MFMAs each
    constexpr int PPS = 8 / NSL;            // softmax element pairs per QK slot (1 at D=128, 2 at D=64)
    static_assert(PF < NSL && PF < NPV, "prefetch distance");

    auto ld_k = [&](int i) -> Vec {         // slot i: ks = i >> 1, kt = i & 1
        return bitcast<Vec>(*reinterpret_cast<const uint4 *>(kb + L::KS * (32 * HN + 16 * (i & 1)) + 64 * (i >> 1)));
    };
    auto ld_v = [&](int dt) -> Vec {
        const i16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_i16x4 *)(vb + L::VS * (32 * HO) + 32 * dt));
        const i16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_i16x4 *)(vb + L::VS * (32 * HO + 16) + 32 * dt));
        u32x4 av;
        const u32x2 a_lo = bitcast<u32x2>(t0), a_hi = bitcast<u32x2>(t1);
        av[0] = a_lo[0]; av[1] = a_lo[1]; av[2] = a_hi[0]; av[3] = a_hi[1];
        return bitcast<Vec>(av);
    };

    // softmax of sO in element pairs p = 0..7: qb = p >> 2, kt = (p >> 1) & 1, registers 2(p&1), 2(p&1)+1;
    // pb[qb][p & 3] is register p & 3 of the B operand of query block qb
    uint32_t pb[2][4];
    float rs0[2] = {0.f, 0.f}, rs1[2] = {0.f, 0.f};
    auto stage_x = [&](int p) {
        f32x4 &s4 = sO[p >> 2][(p >> 1) & 1];
        const int r = 2 * (p & 1);
        s4[r] = fast_exp2(EXACT ? s4[r] * c2 : s4[r]);
        s4[r + 1] = fast_exp2(EXACT ? s4[r + 1] * c2 : s4[r + 1]);
    };
    auto stage_a = [&](int p) {
        const float a = sO[p >> 2][(p >> 1) & 1][2 * (p & 1)], b = sO[p >> 2][(p >> 1) & 1][2 * (p & 1) + 1];
        rs0[p >> 2] += a;
        rs1[p >> 2] += b;
        pb[p >> 2][p & 3] = Tr::pack2(a, b);
    };
    auto soft_slot = [&](int u) {           // slot u: v_exp of the pairs of group u, sum+pack of group u-1
#pragma unroll
        for (int p = (u - 1) * PPS; p < u * PPS; ++p)
            if (p >= 0 && p < 8) stage_a(p);
#pragma unroll
        for (int p = u * PPS; p < (u + 1) * PPS; ++p)
            if (p >= 0 && p < 8) stage_x(p);
    };

    Vec kf[NSL], vf[NPV];
    // ---- slot 0 ----
    if (DO_QK) {
#pragma unroll
        for (int i = 0; i < PF; ++i) kf[i] = kpre[i];
        kf[PF] = ld_k(PF);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) sN[qb][0] = Mfma16<Tr>::run(kf[0], qf[qb][0], acc.cinit[qb]);
        soft_slot(0);
    } else {
#pragma unroll
        for (int i = 0; i < PF; ++i) vf[i] = ld_v(i);
#pragma unroll
        for (int u = 0; u <= 8 / PPS; ++u) soft_slot(u);
    }
    SFA_FENCE();
    if (DO_QK) {
#pragma unroll
        for (int i = 1; i < NSL; ++i) {
            if (i + PF < NSL) kf[i + PF] = ld_k(i + PF); else vf[i + PF - NSL] = ld_v(i + PF - NSL);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
                sN[qb][i & 1] = Mfma16<Tr>::run(kf[i], qf[qb][i >> 1], (i >> 1) == 0 ? acc.cinit[qb] : sN[qb][i & 1]);
            soft_slot(i);
            qk_hook(i);
            SFA_FENCE();
        }
    }
    float m0[2] = {ninf(), ninf()}, m1[2] = {ninf(), ninf()};
#pragma unroll
    for (int j = 0; j < NPV; ++j) {
        if (j + PF < NPV) {
            vf[j + PF] = ld_v(j + PF);
        } else if (PREF) {                  // last PF slots: first K fragments of the next half-step
            const int i = j + PF - NPV;
            kpre[i] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(
                kb_pref + L::KS * (32 * PH + 16 * (i & 1)) + 64 * (i >> 1)));
        }
        if (j == 0 && DO_QK) soft_slot(NSL);        // sum+pack of the last pair group
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            uint4 w;
            w.x = pb[qb][0]; w.y = pb[qb][1]; w.z = pb[qb][2]; w.w = pb[qb][3];
            acc.o[qb][j] = Mfma16<Tr>::run(vf[j], bitcast<Vec>(w), acc.o[qb][j]);
        }
        if (DO_QK && j >= NPV / 2) {        // lane max of the new scores, a quarter (or half) per slot
            constexpr int QS = 4 / (NPV / 2);           // (qb, kt) quads per slot: 1 at D=128, 2 at D=64
#pragma unroll
            for (int x = (j - NPV / 2) * QS; x < (j - NPV / 2 + 1) * QS; ++x) {
                const f32x4 &q4 = sN[x >> 1][x & 1];
                m0[x >> 1] = max3(m0[x >> 1], q4[0], q4[1]);
                m1[x >> 1] = max3(m1[x >> 1], q4[2], q4[3]);
            }
        }
        pv_hook(j);
        SFA_FENCE();
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) acc.lsum[qb] += rs0[qb] + rs1[qb];
    if (DO_QK)
        finish16<D, EXACT>(sN, acc, c2, fmaxf(m0[0], m1[0]), fmaxf(m0[1], m1[1]), mask_n, kbase_n, g, lim);
}

}  // namespace prefill
}  // namespace sfa
