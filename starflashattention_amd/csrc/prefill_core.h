// Core of the gfx950 prefill kernels: the padded LDS images, the per-wave online-softmax state and
// the explicitly slot-ordered pipelined half-step (h_block).  Shared by prefill_kernel.hip
// (256-row workgroups, 64-key tiles) and prefill_kernel_bm128.hip (128-row workgroups, 32-key tiles).
// Design notes: prefill_kernel.hip.
#pragma once
#include <type_traits>

#include "prefill_common.h"

namespace sfa {
namespace prefill {

constexpr float kRescaleThr = 8.0f;     // log2 units

//   K rows: 2*D + 16 bytes.  ds_read_b128 lane groups read 16 rows (distinct mod 16) at one chunk:
//           slot = (17*row + ch) mod 16 (D=128), (9*row + ch) mod 16 (D=64) -> conflict-free.
//   V rows: 2*D + 64 bytes.  a 32-lane half of ds_read_b64_tr_b16 reads 4 consecutive rows x 64
//           contiguous bytes: 320q mod 256 = 64q (D=128), 192q mod 256 = {0,192,128,64} (D=64)
//           -> the four rows tile the 256-byte bank row, conflict-free.
template <int D, int BN = kBN, int NKB = 3, int NVB = 3> struct Lds {
    static constexpr int KS = 2 * D + 16;           // K row stride (bytes)
    static constexpr int VS = 2 * D + 64;           // V row stride
    static constexpr int KTILE = BN * KS;
    static constexpr int VTILE = BN * VS;
    static constexpr int V_BASE = NKB * KTILE;      // K[NKB] then V[NVB]
    static constexpr int TOTAL = NKB * KTILE + NVB * VTILE;
    static_assert(NVB * VTILE < 65536 && NKB * KTILE < 65536, "ds immediates are 16 bit");
};

// key of register r = kbase + (r&3) + 8*(r>>2) + 4*h2
__device__ __forceinline__ void mask_half(f32x16 &s, int kbase, int h2, int lim) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (kbase + (r & 3) + 8 * (r >> 2) + 4 * h2 > lim) s[r] = ninf();
}

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

__device__ __forceinline__ float lane_rowmax(const f32x16 &s) {
    float m0 = max3(s[0], s[1], s[2]), m1 = max3(s[3], s[4], s[5]);
    m0 = max3(m0, s[6], s[7]);
    m1 = max3(m1, s[8], s[9]);
    m0 = max3(m0, s[10], s[11]);
    m1 = max3(m1, s[12], s[13]);
    return fmaxf(max3(m0, s[14], s[15]), m1);
}

#define SFA_FENCE() __builtin_amdgcn_sched_barrier(0)
// sched_barrier orders the machine scheduler only: LLVM's IR passes still SINK a running sum or max
// whose only use is at the end of the half-step down to that use -- all 16 row-sum adds (and in
// prescaled mode the lane max) end up bunched behind the last PV MFMA.  Pinning each update to its
// slot with an empty asm measured 1.5-2 % SLOWER, so the sinking is left alone.

// Per-wave online-softmax state of NQB query blocks.
template <int D, int NQB>
struct Acc {
    f32x16 o[NQB][D / 32];      // O^T accumulators
    float msc[NQB];             // reference max the exponentials are taken against (log2 units)
    float lsum[NQB];            // this lane's share of the running row sum
    f32x16 cinit[NQB];          // prescaled mode (ORD 6): -msc in all 16 registers, the C operand of
                                // the first QK^T MFMA, so scores come out of the MFMA already
                                // relative to the reference max and in log2 units
};

// Prescaled mode: finish the row max of freshly computed scores s (already relative to acc.msc, log2
// units).  Lazy rescale: only when some row of the wave rose more than kRescaleThr above the reference
// do O, the row sum, the pending scores and cinit move to the new reference (wave-uniform, rare).
template <int D, int NQB>
__device__ __forceinline__ void finish_prescaled(f32x16 &s, Acc<D, NQB> &acc, int q, float mxl, int masked,
                                                 int kbase, int h2, int lim) {
    if (masked) {                                       // wave-uniform, diagonal / ragged tiles only
        mask_half(s, kbase, h2, lim);
        mxl = lane_rowmax(s);
    }
    // a row's maximum is the larger of its two lanes' maxima, so "some row rose above the threshold"
    // needs no cross-lane exchange; the exchange happens inside the rare branch only
    if (__any(mxl > kRescaleThr)) {
        const float mx = half_max(mxl);                 // both lane halves hold the same query
        const float d = fmaxf(mx, 0.f);                 // rows that did not rise keep their reference
        const float alpha = fast_exp2(-d);
        acc.msc[q] += d;
        acc.lsum[q] *= alpha;
#pragma unroll
        for (int b = 0; b < D / 32; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc.o[q][b][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] -= d; acc.cinit[q][r] = -acc.msc[q]; }
    }
}

// One pipelined half-step in explicit slot order, for all NQB query blocks of the wave:
//   sN[q] <- scores of K rows [32*HN, +32) of the tile at kb             (DO_QK; NKS*NQB MFMAs)
//   sO[q]  = scores of keys [32*HO, +32) of the tile whose V is at vb: row max finished (slot 0),
//            exponentiated in place, packed to 16 bit, O^T += V^T . P^T  (NPV*NQB MFMAs)
// kb / vb / kb_pref already include this lane's read base (Lds<D> comment).
//   kpre[PF]  in: first PF K fragments of this half-step (read from LDS earlier);
//             out (PREF): first PF fragments of the next half-step, rows [32*PH, +32) at kb_pref
//   mxO[q]    in: this lane's max over the 16 scores in sO[q] (before masking)
//   mxN[q]    out: this lane's max over the 16 new scores
//   mask_o    bit q set: sO[q] holds keys that must be masked (diagonal / ragged tiles)
//   PF        how many slots ahead of its MFMAs a fragment is read
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

template <class Tr, int D, int NQB, int PF, int ORD, int HN, int HO, bool DO_QK, bool PREF, class QkHook = NoHook, class PvHook = NoHook, int PH = 1 - HN>
__device__ __forceinline__ void h_block(const char *kb, const char *vb, const char *kb_pref,
                                        const typename Tr::mfma_vec (&qf)[NQB][D / 16],
                                        f32x16 (&sN)[NQB], f32x16 (&sO)[NQB], Acc<D, NQB> &acc, float c2,
                                        const float (&mxO)[NQB], float (&mxN)[NQB], int mask_o, int kbase_o,
                                        int h2, const int (&lim)[NQB], typename Tr::mfma_vec (&kpre)[PF],
                                        const QkHook &qk_hook = QkHook(), const PvHook &pv_hook = PvHook()) {
    // qk_hook(i) / pv_hook(j): extra work the caller wants issued inside QK slot i / PV slot j
    // (staging loads and stores spread under the MFMAs instead of bunched at the barrier)
    using Vec = typename Tr::mfma_vec;
    constexpr int NKS = D / 16, NDB = D / 32, NPV = 2 * NDB;
    constexpr int KS = Lds<D>::KS, VS = Lds<D>::VS;
    constexpr int EP = 16 / NPV;            // elements per early PV slot     (elements 8..15)
    constexpr int EM = 32 / NPV;            // new scores max-ed per late PV slot
    // ORD == 6, "prescaled": Q was multiplied by scale*log2(e) when it was loaded and the first
    // QK^T MFMA starts from C = -msc, so sO IS the exp2 argument: no scale/subtract VALU pass.  The
    // row max of the new scores is finished at the END of the half-step that computed them (mask_o /
    // kbase_o then describe sN, and mxO / mxN are unused), so msc is final before the next
    // half-step's first MFMA reads cinit.
    constexpr bool PS = (ORD == 6);

    auto ld_k = [&](int ks) -> Vec {
        return bitcast<Vec>(*reinterpret_cast<const uint4 *>(kb + KS * 32 * HN + 32 * ks));
    };
    auto ld_v = [&](int j) -> Vec {         // A operand of PV MFMAs j: d block j % NDB, k-step j / NDB
        const int d = j % NDB, k = j / NDB;
        const i16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_i16x4 *)(vb + VS * 16 * (2 * HO + k) + 64 * d));
        const i16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_i16x4 *)(vb + VS * (16 * (2 * HO + k) + 8) + 64 * d));
        u32x4 av;
        const u32x2 a_lo = bitcast<u32x2>(t0), a_hi = bitcast<u32x2>(t1);
        av[0] = a_lo[0]; av[1] = a_lo[1]; av[2] = a_hi[0]; av[3] = a_hi[1];
        return bitcast<Vec>(av);
    };

    Vec kf[NKS], vf[NPV];
    // ---- slot 0: first QK MFMAs next to the finish of sO's row max ----
    if (DO_QK) {
#pragma unroll
        for (int i = 0; i < PF; ++i) kf[i] = kpre[i];
        if (PF < NKS) kf[PF] = ld_k(PF); else vf[PF - NKS] = ld_v(PF - NKS);
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
        for (int q = 0; q < NQB; ++q) sN[q] = Tr::mfma32(kf[0], qf[q][0], PS ? acc.cinit[q] : z);
    } else {
#pragma unroll
        for (int i = 0; i < PF; ++i) vf[i] = ld_v(i);
    }
    // which slot a pair's first stage runs in (see the staging note below); PPS pairs share a slot
    constexpr int PPS = (NKS >= 8) ? 1 : 2;
    if (PS && (NKS >= 8 || NKS == 4) && DO_QK) {    // stage X of the first pair(s) already in slot 0
#pragma unroll
        for (int q = 0; q < NQB; ++q)
#pragma unroll
            for (int e = 0; e < 2 * PPS; ++e) sO[q][e] = fast_exp2(sO[q][e]);
    }
    float msafe[NQB] = {};
#pragma unroll
    for (int q = 0; q < NQB && !PS; ++q) {
        float mxl = mxO[q];
        if (mask_o & (1 << q)) {                        // wave-uniform, diagonal / ragged tiles only
            mask_half(sO[q], kbase_o, h2, lim[q]);
            mxl = lane_rowmax(sO[q]);
        }
        // (a row's two lanes share msc, so the trigger needs no cross-lane exchange)
        if (__any(mxl * c2 > acc.msc[q] + kRescaleThr)) {       // rare after the first tiles
            const float mx = half_max(mxl) * c2;        // both lane halves hold the same query
            const float mnew = fmaxf(acc.msc[q], mx);
            const float alpha = (mnew == ninf()) ? 1.0f : fast_exp2(acc.msc[q] - mnew);
            acc.msc[q] = mnew;
            acc.lsum[q] *= alpha;
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc.o[q][d][r] *= alpha;
        }
        msafe[q] = (acc.msc[q] == ninf()) ? 0.f : acc.msc[q];
    }
    SFA_FENCE();

    uint32_t pk[NQB][8];                    // P^T packed: pk[q][4k .. 4k+3] is the B operand of k-step k
    // (v_pk_fma_f32 / v_pk_add_f32 on register pairs halve the instruction count of the scale and
    // row-sum passes but measured 3 % SLOWER than the scalar forms: 884 vs 915 TFLOPS)
    float rs0[NQB], rs1[NQB];
#pragma unroll
    for (int q = 0; q < NQB; ++q) { rs0[q] = 0.f; rs1[q] = 0.f; }
    auto soft1 = [&](int e) {               // element e of every query block; packs completed pairs
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            sO[q][e] = PS ? fast_exp2(sO[q][e]) : fast_exp2(fmaf(sO[q][e], c2, -msafe[q]));
            if (e & 1) { rs1[q] += sO[q][e]; pk[q][e >> 1] = Tr::pack2(sO[q][e - 1], sO[q][e]); }
            else { rs0[q] += sO[q][e]; }
        }
    };

    // ORD == 2: the same work software-pipelined across slots in three stages per element pair --
    // F (scale+subtract), X (v_exp), A (row sum + pack) -- so no instruction sits right behind
    // the one it depends on (fma -> exp -> add/cvt back to back stalls on VALU/TRANS latency).
    // Pair g (elements 2g, 2g+1) does F in soft-slot g, X in g+1, A in g+2; soft-slot u is QK slot
    // u+1 for u < NKS-1 and PV slot u-(NKS-1) after that (pairs 0..3 must be packed before the first PV
    // MFMA, pairs 4..7 before PV slot NPV/2).  head_dim 64 has only 3 + 4 such slots: there pairs 0..3
    // start together in soft-slot 0 and pairs 4..7 in soft-slot 2 (prescaled mode: two pairs per slot).
    constexpr bool STAGED = (ORD == 2 || ORD == 6) && (NKS >= 8 || NKS == 4) && DO_QK;     // ORD: 0 = plain slices, 2 = staged, 1 = VALU before MFMA (A/B: no gain)
    auto fslot = [](int g) -> int { return NKS >= 8 ? g : (g < 4 ? 0 : 2); };      // exact mode: slot of stage F
    auto xslot = [](int g) -> int { return g / PPS; };                             // prescaled mode: slot of stage X
    auto stage_f = [&](int g) {
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            sO[q][2 * g] = fmaf(sO[q][2 * g], c2, -msafe[q]);
            sO[q][2 * g + 1] = fmaf(sO[q][2 * g + 1], c2, -msafe[q]);
        }
    };
    auto stage_x = [&](int g) {
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            sO[q][2 * g] = fast_exp2(sO[q][2 * g]);
            sO[q][2 * g + 1] = fast_exp2(sO[q][2 * g + 1]);
        }
    };
    auto stage_a = [&](int g) {
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            rs0[q] += sO[q][2 * g];
            rs1[q] += sO[q][2 * g + 1];
            pk[q][g] = Tr::pack2(sO[q][2 * g], sO[q][2 * g + 1]);
        }
    };
    auto staged_slot = [&](int u) {
        if (PS) {                           // two stages: X in slot xslot(g), A one slot later
            const int w = u + 1;            // (slot 0 already did X of the pairs with xslot 0)
#pragma unroll
            for (int g = 0; g < 8; ++g)
                if (xslot(g) + 1 == w) stage_a(g);
#pragma unroll
            for (int g = 0; g < 8; ++g)
                if (xslot(g) == w) stage_x(g);
            return;
        }
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (fslot(g) + 2 == u) stage_a(g);
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (fslot(g) + 1 == u) stage_x(g);
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (fslot(g) == u) stage_f(g);
    };

    if (DO_QK) {
#pragma unroll
        for (int i = 1; i < NKS; ++i) {     // elements 0..7 spread over slots 1..NKS-1
            if (i + PF < NKS) kf[i + PF] = ld_k(i + PF); else vf[i + PF - NKS] = ld_v(i + PF - NKS);
            if (ORD != 1) {
#pragma unroll
                for (int q = 0; q < NQB; ++q) sN[q] = Tr::mfma32(kf[i], qf[q][i], sN[q]);
            }
            if (STAGED) {
                staged_slot(i - 1);
            } else {
#pragma unroll
                for (int e = (i - 1) * 8 / (NKS - 1); e < i * 8 / (NKS - 1); ++e) soft1(e);
            }
            qk_hook(i);
            if (ORD == 1) {     // VALU slice first: it runs while this slot's fragment is still in flight
#pragma unroll
                for (int q = 0; q < NQB; ++q) sN[q] = Tr::mfma32(kf[i], qf[q][i], sN[q]);
            }
            SFA_FENCE();
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) soft1(e);
        SFA_FENCE();
    }
    float m0[NQB], m1[NQB];
#pragma unroll
    for (int q = 0; q < NQB; ++q) { m0[q] = ninf(); m1[q] = ninf(); }
#pragma unroll
    for (int j = 0; j < NPV; ++j) {
        if (j + PF < NPV) {
            vf[j + PF] = ld_v(j + PF);
        } else if (PREF) {                  // last PF slots: first K fragments of the next half-step
            kpre[j + PF - NPV] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(
                kb_pref + KS * 32 * PH + 32 * (j + PF - NPV)));
        }
        auto pv_mfma = [&]() {
#pragma unroll
            for (int q = 0; q < NQB; ++q) {
                uint4 w;
                w.x = pk[q][4 * (j / NDB) + 0]; w.y = pk[q][4 * (j / NDB) + 1];
                w.z = pk[q][4 * (j / NDB) + 2]; w.w = pk[q][4 * (j / NDB) + 3];
                acc.o[q][j % NDB] = Tr::mfma32(vf[j], bitcast<Vec>(w), acc.o[q][j % NDB]);
            }
        };
        if (ORD != 1) pv_mfma();
        if (STAGED) {
            staged_slot(NKS - 1 + j);
        } else if (j < NPV / 2) {
#pragma unroll
            for (int e = 0; e < EP; ++e) soft1(8 + EP * j + e);
        }
        if (j >= NPV / 2 && DO_QK) {
#pragma unroll
            for (int q = 0; q < NQB; ++q)
#pragma unroll
                for (int e = 0; e < EM; e += 4) {
                    const int r = EM * (j - NPV / 2) + e;
                    m0[q] = max3(m0[q], sN[q][r], sN[q][r + 1]);
                    m1[q] = max3(m1[q], sN[q][r + 2], sN[q][r + 3]);
                }
        }
        pv_hook(j);
        if (ORD == 1) pv_mfma();
        SFA_FENCE();
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
        acc.lsum[q] += rs0[q] + rs1[q];
        if (!PS) mxN[q] = fmaxf(m0[q], m1[q]);
    }
    if (PS && DO_QK) {
#pragma unroll
        for (int q = 0; q < NQB; ++q) finish_prescaled<D, NQB>(sN[q], acc, q, fmaxf(m0[q], m1[q]),
                                                               (mask_o >> q) & 1, kbase_o, h2, lim[q]);
    }
}

// Epilogue store of one 32-row query block: O[row][0..D) = o^T * inv, converted to 16 bit.
// The 32x32 accumulator leaves every output row split across the two lane halves (lane l: columns
// 8g..8g+3 of group g, lane l+32: 8g+4..8g+7), i.e. 8-byte stores.  One v_permlane32_swap per dword
// trades group g's upper half against group g+1's lower half, after which each lane owns 16
// contiguous bytes: half as many (twice as wide) store instructions -- the tail is store-ISSUE
// bound (cdna_hip_programming.md T21).  `row` points at column 0 of this lane's output row.
template <class Tr, int D>
__device__ __forceinline__ void store_o_row(uint16_t *row, const f32x16 (&o)[D / 32], float inv, int h2) {
#pragma unroll
    for (int d = 0; d < D / 32; ++d) {
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
            uint32_t ax = Tr::pack2(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv);
            uint32_t ay = Tr::pack2(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
            uint32_t bx = Tr::pack2(o[d][4 * g + 4] * inv, o[d][4 * g + 5] * inv);
            uint32_t by = Tr::pack2(o[d][4 * g + 6] * inv, o[d][4 * g + 7] * inv);
            const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            // lower lanes: [own g | upper's g] = columns 8g..8g+7; upper lanes: [lower's g+1 | own g+1]
            *reinterpret_cast<uint4 *>(row + 32 * d + 8 * g + 8 * h2) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
        }
    }
}

}  // namespace prefill
}  // namespace sfa
