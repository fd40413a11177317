// Fused attention forward (prefill) for gfx950 (MI355X): the 4-wave, one-wave-per-SIMD, persistent
// kernel.  bf16 / fp16, head_dim 128, causal or full, MHA or GQA, any strides.  The reference has no
// prefill kernel; this is the kernel BASELINE.json's headline metric is quoted on (SURVEY.md 8(a) A-new).
//
// Why this structure (the 8-wave kernel in prefill_kernel.hip is its fallback for small grids and
// head_dim 64): with two waves per SIMD both the vector-issue port and the matrix pipe of a SIMD were
// saturated at ~1.0 PFLOP/s and the older wave of each SIMD idled a quarter of the time at the barrier.
// Here ONE wave owns a SIMD and its whole 512-entry register file:
//   * workgroup = 4 waves = one 256-row q-tile; a wave owns 64 query rows = two 32-row query blocks, so
//     every K / V fragment read from LDS feeds TWO MFMAs (half the LDS operand traffic per FLOP).
//   * O^T (128 registers) and the Q^T fragments (64) live in the ACCUMULATOR half of the register file
//     for the whole q-tile: the MFMAs are inline asm with "a"-constrained operands, so hipcc allocates
//     them there and never copies them (the compiler-scheduled NQB = 2 attempt of round 1 drowned in
//     v_accvgpr moves).  Scores, P, the K / V fragments and the softmax state stay in the 256 arch VGPRs.
//   * S^T = K . Q^T and O^T += V^T . P^T with v_mfma_f32_32x32x16: the query sits on the lane in both
//     accumulators, the exponentiated S^T registers ARE the B operand of the PV product (no LDS round
//     trip for P), V^T comes out of the row-major tile through ds_read_b64_tr_b16.
//   * K / V tiles (64 keys) arrive by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction,
//     no VGPRs, no ds_write pass) into 3-deep rings, K three tiles and V two tiles ahead of the compute,
//     ONE barrier per tile; the buffer descriptor's bounds check zero-fills the rows past the end of a
//     ragged last tile.  The LDS image is the 8-row x 32-column sub-tiled, XOR-swizzled one of
//     cdna_hip_programming.md T10(a): conflict-free for the ds_read_b128 row reads AND the transposed
//     reads, one base register (+ its ^32 twin) per tensor, every other address bit an immediate.  DMA
//     writes LDS linearly, so the swizzle is applied to the per-lane SOURCE address.
//   * 256 persistent workgroups (one per CU) walk a static, XCD-aware list of q-tiles: blockIdx & 7
//     labels the XCD, which owns a contiguous range of (batch, head)s, so the K/V of the few heads in
//     flight on an XCD are shared through its L2.  Under the causal mask a unit of work is a balanced PAIR
//     of q-tiles of one head (heaviest remaining + lightest: constant cost).  The DMA producers run
//     ahead of the compute ACROSS q-tile and head boundaries, so the K/V stream never drains at a seam.
//   * The half-tile (32-key) software pipeline, the explicit slot order, the lazy rescale and the two
//     numeric flavours (exact scale = default, prescaled Q = opt-in fast_scale) are those of the 8-wave
//     kernel (prefill_core.h), with every slot now carrying two MFMAs per fragment.
//
// Hazards hipcc does not see inside the asm MFMAs (cdna_hip_programming.md section 5.7) and how each is
// covered: (1) an MFMA's result read or overwritten by the VALU needs the MFMA to have drained -- in the
// steady state at least two other MFMAs sit between producer and consumer; everywhere else settle()
// inserts the wait states; (2) a VALU-written VGPR used as an MFMA operand needs two wait states -- the
// packed P registers are written at least one slot before their PV MFMA, and the first MFMA behind any
// freshly written operand carries an `s_nop 1`.
#include "prefill_core.h"

namespace sfa {

namespace {

using namespace prefill;

namespace w4 {

constexpr int kRows = 256;          // query rows per workgroup (q-tile)
constexpr int kKeys = 64;           // keys per K/V tile
constexpr int kThreadsW4 = 256;     // 4 waves, one per SIMD
constexpr int kRing = 3;            // LDS ring depth of K and of V
constexpr float kThr = 8.0f;        // lazy-rescale threshold (log2 units)

// LDS image of one [64 keys][D] 16-bit tile: 8-row groups of D/32 sub-tiles of 8 rows x 64 B.
//   off(row, ch) = RG*(row>>3) + 512*(ch>>2) + 64*(row&7) + 16*((ch&3) ^ ((row>>2)&3))      (ch = 16-B chunk of the row)
template <int D> struct Img {
    static constexpr int RG = 512 * (D / 32);       // bytes of one 8-row group
    static constexpr int TILE = 8 * RG;             // 64 rows
    static constexpr int K_BASE = 0;
    static constexpr int V_BASE = kRing * TILE;
    static constexpr int TOTAL = 2 * kRing * TILE;
};

typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) const u32x4 lds_cu4;
__device__ __forceinline__ u32x4 lds_read16(const lds_char *p) { return *reinterpret_cast<lds_cu4 *>(p); }

// ---- MFMA wrappers: operands by register file ---------------------------------------
// s (VGPR) = k (VGPR) . q (AGPR) [+ s]
template <class Tr>
__device__ __forceinline__ void mfma_qk_first(f32x16 &s, typename Tr::mfma_vec k, typename Tr::mfma_vec q) {
    if constexpr (Tr::id == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(k), "a"(q));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(s) : "v"(k), "a"(q));
}
template <class Tr>
__device__ __forceinline__ void mfma_qk_first_c(f32x16 &s, typename Tr::mfma_vec k, typename Tr::mfma_vec q, const f32x16 &c) {
    // C operand = a VALU-written register tuple: two wait states in front (hazard (2))
    if constexpr (Tr::id == 1) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s) : "v"(k), "a"(q), "v"(c));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s) : "v"(k), "a"(q), "v"(c));
}
template <class Tr>
__device__ __forceinline__ void mfma_qk(f32x16 &s, typename Tr::mfma_vec k, typename Tr::mfma_vec q) {
    if constexpr (Tr::id == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q));
}
// o (AGPR) += v (VGPR) . p (VGPR);  NOP: the operands may have been written by the VALU just before
template <class Tr, bool NOP>
__device__ __forceinline__ void mfma_pv(f32x16 &o, typename Tr::mfma_vec v, typename Tr::mfma_vec pfrag) {
    if constexpr (Tr::id == 1) {
        if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pfrag));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pfrag));
    } else {
        if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pfrag));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(pfrag));
    }
}
// Let every MFMA issued so far drain before the VALU touches a result (hazard (1)): 2 x 16 wait states,
// tied to the values so neither side of the fence can be scheduled across it.
__device__ __forceinline__ void settle(f32x16 &x) { asm volatile("s_nop 15\n\ts_nop 15" : "+v"(x)); }
__device__ __forceinline__ void settle_acc(f32x16 &x) { asm volatile("s_nop 15\n\ts_nop 15" : "+a"(x)); }

// One 1-KiB LDS-DMA piece: 64 lanes x 16 B from `srd`[voff + soff] to LDS[lds .. lds + 1024).
// M0 is written in the statement that uses it (hipcc does not preserve it around asm).
__device__ __forceinline__ void dma_piece(unsigned lds, unsigned voff, u32x4s srd, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

template <int D, int NQB>
struct Acc {
    f32x16 o[NQB][D / 32];      // O^T accumulators (AGPRs)
    float msc[NQB];             // reference max the exponentials are taken against (log2 units)
    float lsum[NQB];            // this lane's share of the running row sum
    f32x16 cinit[NQB];          // prescaled flavour: -msc in all 16 registers (C operand of the first QK^T MFMA)
};

// O, the row sum, (prescaled: the pending scores and cinit) move to a new reference max.  Rare.
template <int D, int NQB>
__device__ __forceinline__ void rescale_o(Acc<D, NQB> &acc, int q, float alpha) {
#pragma unroll
    for (int d = 0; d < D / 32; ++d) settle_acc(acc.o[q][d]);       // PV MFMAs of the previous half-step may be in flight
    acc.lsum[q] *= alpha;
#pragma unroll
    for (int d = 0; d < D / 32; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.o[q][d][r] *= alpha;
}

struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

// One pipelined half-step in explicit slot order, for both query blocks of the wave (prefill_core.h's
// h_block on this kernel's register files and LDS image):
//   sN[q] <- scores of K rows [32*HN, +32) of the tile at kbuf                      (DO_QK; NKS*NQB MFMAs)
//   sO[q]  = scores of keys [32*HO, +32) of the tile whose V is at vbuf: row max finished (slot 0),
//            exponentiated in place, packed to 16 bit, O^T += V^T . P^T            (NPV*NQB MFMAs)
// lds: the workgroup's LDS; k_e / v_e: this lane's read offsets (ks even / e = 0; the odd twins are ^32);
// kbuf / vbuf / kbuf_pref: ring offsets.
template <class Tr, int D, int NQB, int PF, int ORD, int HN, int HO, bool DO_QK, bool PREF, class QkHook = NoHook,
          class PvHook = NoHook, int PH = 1 - HN>
__device__ __forceinline__ void half_step(const lds_char *lds, unsigned k_e, unsigned v_e, int kbuf, int vbuf, int kbuf_pref,
                                          const typename Tr::mfma_vec (&qf)[NQB][D / 16], f32x16 (&sN)[NQB],
                                          f32x16 (&sO)[NQB], Acc<D, NQB> &acc, float c2, const float (&mxO)[NQB],
                                          float (&mxN)[NQB], int mask_o, int kbase_o, int h2, const int (&lim)[NQB],
                                          typename Tr::mfma_vec (&kpre)[PF], const QkHook &qk_hook = QkHook(),
                                          const PvHook &pv_hook = PvHook()) {
    using Vec = typename Tr::mfma_vec;
    constexpr int NKS = D / 16, NDB = D / 32, NPV = 2 * NDB;
    constexpr int RG = Img<D>::RG;
    constexpr bool PS = (ORD == 6);
    static_assert(NKS >= 8, "the staged softmax below assumes >= 8 QK slots (head_dim >= 128)");

    const lds_char *const kb_e = lds + (k_e + kbuf), *const kb_o = lds + ((k_e ^ 32) + kbuf);
    const lds_char *const vb_0 = lds + (v_e + vbuf), *const vb_1 = lds + ((v_e ^ 32) + vbuf);
    const lds_char *const kp_e = lds + (k_e + kbuf_pref), *const kp_o = lds + ((k_e ^ 32) + kbuf_pref);
    auto ld_k = [&](int ks) -> Vec {        // rows 32*HN + (lane & 31), chunk 2*ks + h2
        const lds_char *b = (ks & 1) ? kb_o : kb_e;
        return bitcast<Vec>(lds_read16(b + 4 * RG * HN + 512 * (ks >> 1)));
    };
    auto ld_kp = [&](int ks) -> Vec {
        const lds_char *b = (ks & 1) ? kp_o : kp_e;
        return bitcast<Vec>(lds_read16(b + 4 * RG * PH + 512 * (ks >> 1)));
    };
    auto ld_v = [&](int j) -> Vec {         // A operand of PV MFMAs j: d block j % NDB, k-step j / NDB of this half
        const int d = j % NDB, s = 2 * HO + j / NDB;
        const i16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(vb_0 + RG * (2 * s) + 512 * d));
        const i16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(vb_1 + RG * (2 * s + 1) + 512 * d));
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
        static_assert(PF < NKS, "the K fragments of a half-step outnumber the prefetch distance");
        kf[PF] = ld_k(PF);
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            if (PS) mfma_qk_first_c<Tr>(sN[q], kf[0], qf[q][0], acc.cinit[q]);
            else mfma_qk_first<Tr>(sN[q], kf[0], qf[q][0]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < PF; ++i) vf[i] = ld_v(i);
    }
    if (PS && DO_QK) {          // stage X of the first pair already in slot 0
#pragma unroll
        for (int q = 0; q < NQB; ++q)
#pragma unroll
            for (int e = 0; e < 2; ++e) sO[q][e] = fast_exp2(sO[q][e]);
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
        if (__any(mxl * c2 > acc.msc[q] + kThr)) {      // rare after the first tiles
            const float mx = half_max(mxl) * c2;        // both lane halves hold the same query
            const float mnew = fmaxf(acc.msc[q], mx);
            const float alpha = (mnew == ninf()) ? 1.0f : fast_exp2(acc.msc[q] - mnew);
            acc.msc[q] = mnew;
            rescale_o<D, NQB>(acc, q, alpha);
        }
        msafe[q] = (acc.msc[q] == ninf()) ? 0.f : acc.msc[q];
    }
    SFA_FENCE();

    uint32_t pk[NQB][8];                    // P^T packed: pk[q][4k .. 4k+3] is the B operand of k-step k
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
    // The softmax work software-pipelined across slots in three stages per element pair -- F
    // (scale+subtract), X (v_exp), A (row sum + pack) -- so no instruction sits right behind the one it
    // depends on.  Pair g (elements 2g, 2g+1) does F in soft-slot g, X in g+1, A in g+2; soft-slot u is QK
    // slot u+1 for u < NKS-1 and PV slot u-(NKS-1) after that (pairs 0..3 must be packed before the first
    // PV MFMA, pairs 4..7 before PV slot NPV/2).  Prescaled flavour: two stages (X, A).
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
        if (PS) {
            const int w = u + 1;            // (slot 0 already did X of pair 0)
#pragma unroll
            for (int g = 0; g < 8; ++g)
                if (g + 1 == w) stage_a(g);
#pragma unroll
            for (int g = 0; g < 8; ++g)
                if (g == w) stage_x(g);
            return;
        }
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (g + 2 == u) stage_a(g);
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (g + 1 == u) stage_x(g);
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (g == u) stage_f(g);
    };

    if (DO_QK) {
#pragma unroll
        for (int i = 1; i < NKS; ++i) {
            if (i + PF < NKS) kf[i + PF] = ld_k(i + PF); else vf[i + PF - NKS] = ld_v(i + PF - NKS);
#pragma unroll
            for (int q = 0; q < NQB; ++q) mfma_qk<Tr>(sN[q], kf[i], qf[q][i]);
            staged_slot(i - 1);
            qk_hook(i);
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
    constexpr int EP = 16 / NPV;            // elements per early PV slot (no-QK form)
    constexpr int EM = 32 / NPV;            // new scores max-ed per late PV slot
#pragma unroll
    for (int j = 0; j < NPV; ++j) {
        if (j + PF < NPV) {
            vf[j + PF] = ld_v(j + PF);
        } else if (PREF) {                  // last PF slots: first K fragments of the next half-step
            kpre[j + PF - NPV] = ld_kp(j + PF - NPV);
        }
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            uint4 w;
            w.x = pk[q][4 * (j / NDB) + 0]; w.y = pk[q][4 * (j / NDB) + 1];
            w.z = pk[q][4 * (j / NDB) + 2]; w.w = pk[q][4 * (j / NDB) + 3];
            // the P registers of a k-step are packed in the slot right before its first MFMA
            if (q == 0 && (j % NDB) == 0) mfma_pv<Tr, true>(acc.o[q][j % NDB], vf[j], bitcast<Vec>(w));
            else mfma_pv<Tr, false>(acc.o[q][j % NDB], vf[j], bitcast<Vec>(w));
        }
        if (DO_QK) {
            staged_slot(NKS - 1 + j);
        } else if (j < NPV / 2) {
#pragma unroll
            for (int e = 0; e < EP; ++e) soft1(8 + EP * j + e);
        }
        if (j >= NPV / 2 && DO_QK) {        // lane max of the new scores (their MFMAs ended >= NPV/2 slots ago)
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
        SFA_FENCE();
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
        acc.lsum[q] += rs0[q] + rs1[q];
        if (!PS) mxN[q] = fmaxf(m0[q], m1[q]);
    }
    if (PS && DO_QK) {
        // prescaled: finish the row max of the NEW scores now (already relative to msc, log2 units), so msc
        // is final before the next half-step's first MFMA reads cinit
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            float mxl = fmaxf(m0[q], m1[q]);
            if ((mask_o >> q) & 1) {
                mask_half(sN[q], kbase_o, h2, lim[q]);
                mxl = lane_rowmax(sN[q]);
            }
            if (__any(mxl > kThr)) {
                const float mx = half_max(mxl);
                const float d = fmaxf(mx, 0.f);         // rows that did not rise keep their reference
                const float alpha = fast_exp2(-d);
                acc.msc[q] += d;
                rescale_o<D, NQB>(acc, q, alpha);
#pragma unroll
                for (int r = 0; r < 16; ++r) { sN[q][r] -= d; acc.cinit[q][r] = -acc.msc[q]; }
            }
        }
    }
}

}  // namespace w4

// Which items (q-tiles) a workgroup walks, in which order.  blockIdx & 7 labels the XCD (round-robin
// dispatch; a speed hint only), which owns heads [xcd * bh_per_xcd, +bh_per_xcd); its work list is
// head-major, U units per head -- causal: unit i = the q-tile pair (nq-1-i, i); full: unit i = q-tile i --
// and the XCD's workgroup `slot` takes units slot, slot + nslots, ...  All scalar.
struct W4Cursor {
    int hl, i;          // head index inside the XCD's range, unit inside the head
    int sub;            // causal: 0 = the heavy q-tile of the pair, 1 = the light one
    int t, nt;          // tile inside the item, tiles of the item
    int b, h, qt;       // batch, head, q-tile
    bool live;
};

template <class Tr, int D, bool CAUSAL, int ORD>
__global__ void __launch_bounds__(w4::kThreadsW4, 1)
prefill_w4_kernel(const PrefillKernelParams p) {
    using namespace w4;
    using Vec = typename Tr::mfma_vec;
    constexpr int NQB = 2, PF = 2;
    constexpr bool PS = (ORD == 6);
    constexpr int PSO = PS ? 32 : 0;
    constexpr int NKS = D / 16, NDB = D / 32;
    constexpr int NJ = D / 64;                  // 128-byte column pieces per row
    using L = Img<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h2 = lane >> 5;
    const int coff = p.Sk - p.Sq;               // causal: key j visible iff j <= i + coff
    const int BH = p.B * p.Hq;
    const int nq = (p.Sq + kRows - 1) / kRows;
    const int U = CAUSAL ? (nq + 1) / 2 : nq;   // units per head
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;

    auto item_tiles = [&](int qt) -> int {
        int kv_end = p.Sk;
        if (CAUSAL) kv_end = min(p.Sk, qt * kRows + kRows + coff);
        return kv_end > 0 ? (kv_end + kKeys - 1) / kKeys : 0;
    };
    // position the cursor on the first existing item at or after (hl, i, sub); skip_empty: also skip items
    // without any tile (causal rows that see no key)
    auto seek = [&](W4Cursor &c, bool skip_empty) {
        while (c.hl < p.bh_per_xcd) {
            const int bh = xcd * p.bh_per_xcd + c.hl;
            if (bh >= BH) break;
            const int heavy = CAUSAL ? nq - 1 - c.i : c.i;
            const bool exists = c.sub == 0 || (CAUSAL && heavy != c.i);
            if (exists) {
                if (c.sub == 0 || !CAUSAL) { c.b = bh / p.Hq; c.h = bh - c.b * p.Hq; }
                c.qt = c.sub == 0 ? heavy : c.i;
                c.nt = item_tiles(c.qt);
                c.t = 0;
                if (!skip_empty || c.nt > 0) { c.live = true; return; }
            }
            if (CAUSAL && c.sub == 0) { c.sub = 1; continue; }
            c.sub = 0;
            c.i += nslots;
            while (c.i >= U) { c.i -= U; ++c.hl; }
        }
        c.live = false;
    };
    auto next_item = [&](W4Cursor &c, bool skip_empty) {
        if (CAUSAL && c.sub == 0) { c.sub = 1; }
        else {
            c.sub = 0;
            c.i += nslots;
            while (c.i >= U) { c.i -= U; ++c.hl; }
        }
        seek(c, skip_empty);
    };
    auto first_item = [&](W4Cursor &c, bool skip_empty) {
        c.hl = slot / U; c.i = slot % U; c.sub = 0; c.t = 0; c.nt = 0; c.b = 0; c.h = 0; c.qt = 0; c.live = false;
        seek(c, skip_empty);
    };

    // ---- LDS-DMA producers ----
    // Wave w stages rows [16w, 16w+16) of every tile: row groups 2w (half 0) and 2w+1 (half 1), NJ pieces of
    // 8 rows x 128 B each.  Lane -> (sub-tile lane>>5, row (lane>>2)&7, slot lane&3) of its piece; the source
    // chunk is slot ^ ((row>>2)&3) so that LDS, written linearly, holds the swizzled image.
    const int r8 = (lane >> 2) & 7, dslot = lane & 3, dsub = lane >> 5;
    const unsigned k_rowb = (unsigned)(2 * p.ks[2]), v_rowb = (unsigned)(2 * p.vs[2]);
    const unsigned kvoff0 = (unsigned)r8 * k_rowb + 64u * dsub + 16u * (dslot ^ (r8 >> 2));
    const unsigned kvoff1 = (unsigned)(r8 + 8) * k_rowb + 64u * dsub + 16u * (dslot ^ (2 + (r8 >> 2)));
    const unsigned vvoff0 = (unsigned)r8 * v_rowb + 64u * dsub + 16u * (dslot ^ (r8 >> 2));
    const unsigned vvoff1 = (unsigned)(r8 + 8) * v_rowb + 64u * dsub + 16u * (dslot ^ (2 + (r8 >> 2)));
    const lds_char *const lds = (const lds_char *)smem;
    const unsigned lds0 = (unsigned)(uintptr_t)lds;         // LDS byte address of the dynamic segment
    // One head's K (or V) rows form a buffer of `extent` bytes (launch_prefill_w4 keeps it below 2 GiB); the
    // descriptor a wave uses for a tile starts at ITS 16 rows of that tile and ends with the head, so rows
    // past the sequence end read as zeros.  Per tile the descriptor only moves by one tile's bytes.
    struct Desc { unsigned lo, hi; int left; };
    const int k_extent = (p.Sk - 1) * (int)k_rowb + 2 * D, v_extent = (p.Sk - 1) * (int)v_rowb + 2 * D;
    const int k_tileb = kKeys * (int)k_rowb, v_tileb = kKeys * (int)v_rowb;
    const int G = p.Hq / p.Hkv;
    auto desc_at_head = [&](bool is_k, int b, int h) -> Desc {
        const int hk = h / G;
        const uint16_t *head = is_k ? p.k + b * p.ks[0] + hk * p.ks[1] : p.v + b * p.vs[0] + hk * p.vs[1];
        const unsigned skip = 16u * wave * (is_k ? k_rowb : v_rowb);
        const unsigned long long base = (unsigned long long)(uintptr_t)head + skip;
        return Desc{(unsigned)base, (unsigned)(base >> 32), (is_k ? k_extent : v_extent) - (int)skip};
    };
    auto desc_advance = [&](Desc &d, int tileb) {
        const unsigned lo = d.lo + (unsigned)tileb;
        d.hi += lo < d.lo ? 1u : 0u;
        d.lo = lo;
        d.left -= tileb;
    };
    auto issue_tile = [&](const Desc &d, bool is_k, int ring_off) {
        u32x4s srd;
        srd[0] = d.lo;
        srd[1] = d.hi & 0xffffu;
        srd[2] = (unsigned)max(d.left, 0);
        srd[3] = 0x00020000u;
        const unsigned dst = lds0 + (is_k ? L::K_BASE : L::V_BASE) + ring_off + 2 * wave * L::RG;
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                dma_piece(dst + half * L::RG + 1024 * j, is_k ? (half ? kvoff1 : kvoff0) : (half ? vvoff1 : vvoff0), srd,
                          128u * j);
    };

    // ---- this lane's LDS read bases (the odd twins are ^32) ----
    const int kx = (l31 >> 2) & 3;
    const unsigned k_e = L::K_BASE + L::RG * (l31 >> 3) + 64 * (l31 & 7) + 16 * (h2 ^ kx);
    const int vy = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    const unsigned v_e = L::V_BASE + 64 * (4 * h2 + ((lane & 15) >> 2)) + 16 * (vy ^ h2) + 8 * (lane & 1);

    const float c2 = p.scale_log2;

    // ---- cursors: the compute, and the DMA producer running ahead of it.  The K producer is three stream
    // positions ahead of the compute, the V producer two: V re-issues the tile K issued one call earlier,
    // from the descriptor K's call left behind for it (vpend). ----
    W4Cursor cc, pc;
    first_item(cc, false);
    first_item(pc, true);
    Desc kd = {0, 0, 0}, vd = {0, 0, 0}, vpend = {0, 0, 0};
    bool vpend_live = false;
    if (pc.live) { kd = desc_at_head(true, pc.b, pc.h); vd = desc_at_head(false, pc.b, pc.h); }
    int kring_p = 0, vring_p = 0;               // ring byte offsets the producers write next
    auto ring_next = [](int x) -> int { return x == (kRing - 1) * L::TILE ? 0 : x + L::TILE; };
    auto produce_v = [&]() {                    // V of the stream position K produced last time
        if (vpend_live) issue_tile(vpend, false, vring_p);
        vring_p = kring_p;
    };
    auto produce_k = [&]() {
        vpend_live = pc.live;
        if (pc.live) {
            issue_tile(kd, true, kring_p);
            vpend = vd;
            if (++pc.t < pc.nt) {
                desc_advance(kd, k_tileb);
                desc_advance(vd, v_tileb);
            } else {
                next_item(pc, true);
                if (pc.live) { kd = desc_at_head(true, pc.b, pc.h); vd = desc_at_head(false, pc.b, pc.h); }
            }
        }
        kring_p = ring_next(kring_p);
    };
    // stream prologue: K(0), K(1), V(0) must be visible before the first step; K(2), V(1) in flight
    produce_k(); produce_v(); produce_k(); produce_v(); produce_k();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    int kcur = 0, vcur = 0;                     // ring byte offsets of the compute's current tile
#define SFA_W4_SYNC_AND_STAGE()                                                                     \
    do {                                                                                            \
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                               \
        produce_v();                                                                                \
        produce_k();                                                                                \
    } while (0)

    Vec qf[NQB][NKS];
    auto load_q = [&](int b, int h, int qt) {
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            const int qrow = qt * kRows + 64 * wave + 32 * q + l31;
            const uint16_t *qp = p.q + b * p.qs[0] + h * p.qs[1] + (long long)min(qrow, p.Sq - 1) * p.qs[2] + 8 * h2;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                u32x4 w = bitcast<u32x4>(*reinterpret_cast<const uint4 *>(qp + 16 * ks));
                if (PS) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = Tr::pack2(Tr::lo_f32(w[i]) * c2, Tr::hi_f32(w[i]) * c2);
                }
                qf[q][ks] = bitcast<Vec>(w);
            }
        }
    };

    while (cc.live) {
        const int qt = cc.qt, nt = cc.nt;
        const int b = cc.b, h = cc.h;
        load_q(b, h, qt);
        // Q^T now sits in the accumulator file; two wait states between the moves and the first MFMA
#pragma unroll
        for (int q = 0; q < NQB; ++q)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) asm volatile("s_nop 1" : "+a"(qf[q][ks]));
        const int wq0 = qt * kRows + 64 * wave;                 // this wave's first query row
        int ntw = nt;                                           // tiles this wave computes on (wave-uniform)
        if (CAUSAL) ntw = (wq0 + 63 + coff >= 0) ? min(nt, (wq0 + 63 + coff) / kKeys + 1) : 0;
        int lim[NQB];                                           // last visible key of this lane's rows
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            const int qrow = wq0 + 32 * q + l31;
            lim[q] = CAUSAL ? min(p.Sk - 1, qrow + coff) : p.Sk - 1;
        }
        // bit q set: the 32 keys starting at kbase need masking for query block q (wave-uniform)
        auto mask_bits = [&](int kbase) -> int {
            int m = 0;
#pragma unroll
            for (int q = 0; q < NQB; ++q)
                if ((CAUSAL && (kbase + 31 > wq0 + 32 * q + coff)) || (kbase + 32 > p.Sk)) m |= 1 << q;
            return m;
        };

        Acc<D, NQB> acc;
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc.o[q][d][r] = 0.f;
            acc.msc[q] = PS ? 0.f : ninf();
            acc.lsum[q] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc.cinit[q][r] = 0.f;
        }

        // ---- scores of the first half-tile, first fragments of the second ----
        f32x16 sA[NQB], sB[NQB];
        float mxA[NQB], mxB[NQB];
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            mxA[q] = ninf(); mxB[q] = ninf();
#pragma unroll
            for (int r = 0; r < 16; ++r) { sA[q][r] = 0.f; sB[q][r] = 0.f; }
        }
        Vec kpre[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) kpre[i] = bitcast<Vec>(make_uint4(0, 0, 0, 0));
        if (ntw > 0) {
            const lds_char *const kb_e = lds + (k_e + kcur), *const kb_o = lds + ((k_e ^ 32) + kcur);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const Vec a = bitcast<Vec>(lds_read16(((ks & 1) ? kb_o : kb_e) + 512 * (ks >> 1)));
#pragma unroll
                for (int q = 0; q < NQB; ++q) {
                    if (ks == 0) mfma_qk_first<Tr>(sA[q], a, qf[q][0]);
                    else mfma_qk<Tr>(sA[q], a, qf[q][ks]);
                }
            }
#pragma unroll
            for (int i = 0; i < PF; ++i)
                kpre[i] = bitcast<Vec>(lds_read16(((i & 1) ? kb_o : kb_e) + 4 * L::RG + 512 * (i >> 1)));
#pragma unroll
            for (int q = 0; q < NQB; ++q) {
                settle(sA[q]);
                mxA[q] = lane_rowmax(sA[q]);
                if (PS) {           // the first half-tile sets the reference outright (scores may sit far below 0)
                    if (mask_bits(0) & (1 << q)) {
                        mask_half(sA[q], 0, h2, lim[q]);
                        mxA[q] = lane_rowmax(sA[q]);
                    }
                    const float mx = half_max(mxA[q]);
                    const float m0 = (mx == ninf()) ? 0.f : mx;
                    acc.msc[q] = m0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { sA[q][r] -= m0; acc.cinit[q][r] = -m0; }
                }
            }
        }

        int t = 0;
        // ---- FULL steps: this wave needs the next tile as well.  The DMA pieces of K(t+3) and V(t+2) ride in
        // the QK slots of H2, right behind the barrier that freed their ring slots.
        //   H1(t): QK^T(B_t)     || max,exp(A_t),   PV(A_t) || lane max(B_t)
        //   H2(t): QK^T(A_{t+1}) || max,exp(B_t),   PV(B_t) || lane max(A_{t+1})
        for (; t + 1 < ntw; ++t) {
            const int k1 = ring_next(kcur);
            const int kbase = t * kKeys;
            half_step<Tr, D, NQB, PF, ORD, 1, 0, true, true>(lds, k_e, v_e, kcur, vcur, k1, qf, sB, sA, acc, c2, mxA, mxB,
                                                            mask_bits(kbase + PSO), kbase + PSO, h2, lim, kpre);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            auto dma_hook = [&](int i) {
                if (i == 1) produce_v();
                if (i == 4) produce_k();
            };
            half_step<Tr, D, NQB, PF, ORD, 0, 1, true, true>(lds, k_e, v_e, k1, vcur, k1, qf, sA, sB, acc, c2, mxB, mxA,
                                                            mask_bits(kbase + 32 + PSO), kbase + 32 + PSO, h2, lim, kpre,
                                                            dma_hook);
            kcur = k1;
            vcur = ring_next(vcur);
        }
        // ---- TAIL step: this wave's last tile (its second half computes no new scores) ----
        if (t < ntw) {
            const int kbase = t * kKeys;
            half_step<Tr, D, NQB, PF, ORD, 1, 0, true, false>(lds, k_e, v_e, kcur, vcur, kcur, qf, sB, sA, acc, c2, mxA, mxB,
                                                             mask_bits(kbase + PSO), kbase + PSO, h2, lim, kpre);
            SFA_W4_SYNC_AND_STAGE();
            half_step<Tr, D, NQB, PF, ORD, 0, 1, false, false>(lds, k_e, v_e, kcur, vcur, kcur, qf, sA, sB, acc, c2, mxB, mxA,
                                                              mask_bits(kbase + 32), kbase + 32, h2, lim, kpre);
            kcur = ring_next(kcur);
            vcur = ring_next(vcur);
            ++t;
        }
        // ---- idle steps (causal: tiles beyond this wave's diagonal): keep staging for the others ----
        for (; t < nt; ++t) {
            SFA_W4_SYNC_AND_STAGE();
            kcur = ring_next(kcur);
            vcur = ring_next(vcur);
        }

        // ---- epilogue: normalise, convert, store O[row][:] ----
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
#pragma unroll
            for (int d = 0; d < NDB; ++d) settle_acc(acc.o[q][d]);
            const int qrow = wq0 + 32 * q + l31;
            const float ltot = half_sum(acc.lsum[q]);
            const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
            if (qrow < p.Sq) {
                uint16_t *orow = p.o + b * p.os[0] + h * p.os[1] + (long long)qrow * p.os[2];
                store_o_row<Tr, D>(orow, acc.o[q], inv, h2);
                if (p.lse && h2 == 0) {
                    const float lse = ltot > 0.f ? (acc.msc[q] + __log2f(ltot)) * kLn2 : ninf();
                    p.lse[((long long)b * p.Hq + h) * p.Sq + qrow] = lse;
                }
            }
        }
        next_item(cc, false);
    }
    // drain: DMA pieces issued for stream positions nobody consumes do not exist (the producers stop at the
    // end of the list), but the last steps' pieces must have landed before the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef SFA_W4_SYNC_AND_STAGE
}

template <class Tr, int D, int ORD>
int launch_w4_t(const PrefillKernelParams &p, bool causal, hipStream_t stream) {
    using namespace w4;
    const int lds = Img<D>::TOTAL;
    // one workgroup per CU, fewer when the XCD lists are shorter than 32 units
    const int nq = (p.Sq + kRows - 1) / kRows;
    const long long units_xcd = (long long)p.bh_per_xcd * (causal ? (nq + 1) / 2 : nq);
    const int nslots = (int)(units_xcd < 32 ? units_xcd : 32);
    dim3 grid(8u * nslots), block(kThreadsW4);
    static DynLdsAttr attr_c, attr_f;
    if (const int rc = causal ? attr_c.ensure(reinterpret_cast<const void *>(&prefill_w4_kernel<Tr, D, true, ORD>), lds,
                                              "prefill_w4_kernel")
                              : attr_f.ensure(reinterpret_cast<const void *>(&prefill_w4_kernel<Tr, D, false, ORD>), lds,
                                              "prefill_w4_kernel"))
        return rc;
    if (causal) hipLaunchKernelGGL((prefill_w4_kernel<Tr, D, true, ORD>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((prefill_w4_kernel<Tr, D, false, ORD>), grid, block, lds, stream, p);
    return check_launch("prefill_w4_kernel");
}

}  // namespace

// force: 0 = flavour by policy (exact unless the caller opted into fast_scale), 1 = prescaled, 2 = exact
int launch_prefill_w4(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream, int force) {
    if (dtype != SFA_DTYPE_FP16 && dtype != SFA_DTYPE_BF16)
        return fail(SFA_ERR_BAD_DTYPE, "sfa_prefill_fwd: dtype %d is not fp16(0)/bf16(1)", dtype);
    if (head_dim != 128)
        return fail(SFA_ERR_UNSUPPORTED_HEAD_DIM, "sfa_prefill_fwd: the 4-wave kernel serves head_dim 128 (got %d)", head_dim);
    // the K/V rows of one head are addressed through a 32-bit buffer descriptor
    const long long k_ext = (long long)(p.Sk - 1) * 2 * p.ks[2] + 2 * head_dim, v_ext = (long long)(p.Sk - 1) * 2 * p.vs[2] + 2 * head_dim;
    if (k_ext >= (1ll << 31) || v_ext >= (1ll << 31) || p.ks[2] * 2 >= (1ll << 24) || p.vs[2] * 2 >= (1ll << 24))
        return fail(SFA_ERR_BAD_SHAPE, "sfa_prefill_fwd: one head's K/V rows span more than 2 GiB");
    const bool prescaled = force == 0 ? p.fast_scale != 0 : force == 1;
    if (dtype == SFA_DTYPE_FP16)
        return prescaled ? launch_w4_t<Fp16, 128, 6>(p, causal, stream) : launch_w4_t<Fp16, 128, 2>(p, causal, stream);
    return prescaled ? launch_w4_t<Bf16, 128, 6>(p, causal, stream) : launch_w4_t<Bf16, 128, 2>(p, causal, stream);
}

}  // namespace sfa
