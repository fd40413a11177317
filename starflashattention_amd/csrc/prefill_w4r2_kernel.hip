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
#include "prefill_w4_common.h"

namespace sfa {

namespace {

using namespace prefill;

namespace w4 {

using namespace w4c;

constexpr int kRows = 256;          // query rows per workgroup (q-tile)
constexpr int kKeys = 64;           // keys per K/V tile

// LDS image of one [64 keys][D] 16-bit tile: 8-row groups of D/32 sub-tiles of 8 rows x 64 B.
//   off(row, ch) = RG*(row>>3) + 512*(ch>>2) + 64*(row&7) + 16*((ch&3) ^ ((row>>2)&3))      (ch = 16-B chunk of the row)
// RING tiles of K, then RING tiles of V (RING = 3: the DMA of a tile has one tile time to land; 4: two).
template <int D, int RING = 3> struct Img {
    static constexpr int RG = 512 * (D / 32);       // bytes of one 8-row group
    static constexpr int TILE = 8 * RG;             // 64 rows
    static constexpr int K_BASE = 0;
    static constexpr int V_BASE = RING * TILE;
    static constexpr int Q_BASE = 2 * RING * TILE;  // the Q rows of the next q-tile, one 64-row image per wave (wave-private)
    static constexpr int TOTAL = Q_BASE + 4 * TILE;
};

// Per-wave online-softmax state of the two query blocks.
template <int D>
struct Acc {
    f32x16 o[2][D / 32];        // O^T accumulators (AGPRs)
    float msc[2];               // reference max the exponentials are taken against (log2 units)
    float msafe[2];             // msc, or 0 while a row has seen no key yet (what the scale/subtract uses)
    float thr[2];               // exact flavour: msc + kThr, the lazy-rescale trigger (kept so the per-half-step test is mul + compare)
    float lsum[2];              // this lane's share of the running row sum
    float alpha[2];             // a rescale of O decided but not yet applied (see hstep); 1 = none
    uint32_t pk[2][8];          // P^T of the half-tile being consumed, packed: pk[q][4k .. 4k+3] = B operand of k-step k
    f32x16 cinit[2];            // prescaled flavour: -msc in all 16 registers (C operand of the first QK^T MFMA)
};

// O moves to a new reference max.  Rare.
template <int D>
__device__ __forceinline__ void rescale_o(Acc<D> &acc, int q, float alpha) {
#pragma unroll
    for (int d = 0; d < D / 32; ++d) settle_acc(acc.o[q][d]);       // PV MFMAs of the previous half-step may be in flight
#pragma unroll
    for (int d = 0; d < D / 32; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.o[q][d][r] *= alpha;
}



// ---- the element pipeline ----------------------------------------------------------------------
// One wave alone on its SIMD hides work behind an MFMA only in the gap right behind THAT MFMA, and only about
// one v_exp + three plain VALU + one LDS read of it (tools/micro/mfma_overlap.hip: M f M f with 3 VALU + 1 exp
// + 1 ds_read per gap = 34.7 cycles per MFMA; the same fillers bunched behind a PAIR of MFMAs = 59).  So a
// half-step is 32 GAPS -- one MFMA each -- and the softmax of a half-tile is cut into per-ELEMENT stages that
// are dealt out one per gap:
//     F  s = s * c2 - msafe        (exact flavour only)
//     X  s = exp2(s)
//     A  lsum += s; every second element: pack the pair to 16 bit
// (Row sums on the matrix pipe instead -- a fifth MFMA per PV block with an all-ones A operand, no v_add in any
// gap -- measured SLOWER: 965 vs 1005 TFLOPS causal; the register file is full and the pipe gets 12.5 % more work.)
// The 32 score registers a lane holds for a half-tile (2 query blocks x 16) are walked in the order their PV
// MFMAs need them: element i -> block i>>3 = (k-step, query block) in the order (0,q0) (0,q1) (1,q0) (1,q1),
// register 8*kstep + (i&7).  X of element i runs in gap i - 8, F one gap earlier, A one gap later: the first
// eight elements (block (0,q0)) are exponentiated in the LAST eight gaps of the half-step that computed them.
// MFMA order: QK^T query-block-major (gaps 0-7 q0, 8-15 q1: q0's scores are complete half-way through), then
// PV in the block order above (gaps 16-19, 20-23, 24-27, 28-31).
// Row max of the NEW scores: q0 in gaps 9-16 (decision in gap 17), q1 in gaps 17-24 (decision in gap 25).
// A decision to move the reference max (rare) takes effect at once for msc / msafe / lsum -- every row-sum
// add under the old reference is over by then -- but O still has PV MFMAs of the old reference ahead of it, so
// the factor is parked in acc.alpha and applied at the entry of the NEXT half-step (`pend`).
// LDS reads, one per gap, each at least eight gaps ahead of its MFMA: the 16 transposed V reads in gaps 0-15,
// the eight K fragments of the NEXT half-step in gaps 16-23 (kpre).
//
// State at entry (and at exit, for sN): elements 0..7 exponentiated, 0..6 summed, pairs 0..2 packed, element 8
// scaled.  lead_in() establishes it for the first half-tile of a q-tile; hstep_last() consumes the last one.
constexpr int kLead = 8;

__device__ __forceinline__ constexpr int el_q(int i) { return (i >> 3) & 1; }
__device__ __forceinline__ constexpr int el_r(int i) { return 8 * (i >> 4) + (i & 7); }

// The stages are inline asm: a volatile asm statement keeps its place among the (volatile asm) MFMAs, whereas
// plain arithmetic floats -- instruction selection had bunched ten v_exp at the top of a block and left most
// MFMA gaps with nothing but their LDS read.  hipcc still allocates every register.
template <class Tr, int ORD, int I>
__device__ __forceinline__ void st_f(f32x16 (&s)[2], const float (&msafe)[2], float c2) {
    if constexpr (ORD != 6 && I >= 0 && I < 32)
        asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(s[el_q(I)][el_r(I)]) : "s"(c2), "v"(msafe[el_q(I)]));
}
template <int I>
__device__ __forceinline__ void st_x(f32x16 (&s)[2]) {
    if constexpr (I >= 0 && I < 32) asm volatile("v_exp_f32 %0, %0" : "+v"(s[el_q(I)][el_r(I)]));
}
template <class Tr, int I>
__device__ __forceinline__ void st_a(f32x16 (&s)[2], float (&lsum)[2], uint32_t (&pk)[2][8]) {
    if constexpr (I >= 0 && I < 32) {
        constexpr int q = el_q(I), r = el_r(I);
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(lsum[q]) : "v"(s[q][r]));
        if constexpr (I & 1) {
            if constexpr (Tr::id == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[q][r >> 1]) : "v"(s[q][r - 1]), "v"(s[q][r]));
            else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[q][r >> 1]) : "v"(s[q][r - 1]), "v"(s[q][r]));
        }
    }
}
// The reference max of query block q against freshly computed scores s (masked if need be): decide, and if it
// moves, move msc / msafe / lsum now and park the factor for O.  Exact flavour: s are raw Q.K^T; prescaled: s
// are already relative to msc in log2 units and move with it.
template <class Tr, int D, int ORD>
__device__ __forceinline__ void decide(Acc<D> &acc, int q, f32x16 &s, float mxl, float c2, int &pend) {
    if (ORD == 6) {
        if (__any(mxl > kThr)) {
            const float d = fmaxf(half_max(mxl), 0.f);          // rows that did not rise keep their reference
            const float al = fast_exp2(-d);
            acc.msc[q] += d;
            acc.lsum[q] *= al;
            acc.alpha[q] = al;
            pend = 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] -= d; acc.cinit[q][r] = -acc.msc[q]; }
        }
        return;
    }
    // (a row's two lanes share msc, so the trigger needs no cross-lane exchange)
    if (__any(mxl * c2 > acc.thr[q])) {                         // rare after the first tiles
        const float mx = half_max(mxl) * c2;                    // both lane halves hold the same query
        const float mnew = fmaxf(acc.msc[q], mx);
        const float al = (mnew == ninf()) ? 1.0f : fast_exp2(acc.msc[q] - mnew);
        acc.msc[q] = mnew;
        acc.thr[q] = mnew + kThr;
        acc.msafe[q] = (mnew == ninf()) ? 0.f : mnew;
        acc.lsum[q] *= al;
        acc.alpha[q] = al;
        pend = 1;
    }
}

// Apply the rescales of O decided during the previous half-step (wave-uniform, rare).
template <int D>
__device__ __forceinline__ void apply_pending(Acc<D> &acc, int &pend) {
    if (pend) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { rescale_o<D>(acc, q, acc.alpha[q]); acc.alpha[q] = 1.0f; }
        pend = 0;
    }
}

// First half-tile of a q-tile: its scores s were just computed outside the pipeline.  Sets the reference max
// outright and brings s into the entry state of hstep().
template <class Tr, int D, int ORD>
__device__ __forceinline__ void lead_in(Acc<D> &acc, f32x16 (&s)[2], float c2, int mask, int h2, const int (&lim)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        settle(s[q]);
        if (mask & (1 << q)) mask_keys(s[q], 0, h2, lim[q]);
        const float mx = half_max(lane_rowmax(s[q]));
        if (ORD == 6) {
            const float m0 = (mx == ninf()) ? 0.f : mx;
            acc.msc[q] = m0;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[q][r] -= m0; acc.cinit[q][r] = -m0; }
        } else {
            acc.msc[q] = mx * c2;
            acc.thr[q] = acc.msc[q] + kThr;
            acc.msafe[q] = (mx == ninf()) ? 0.f : acc.msc[q];
        }
    }
    static_for<kLead + 1>([&](auto ic) { st_f<Tr, ORD, decltype(ic)::value>(s, acc.msafe, c2); });
    static_for<kLead>([&](auto ic) { st_x<decltype(ic)::value>(s); });
    static_for<kLead - 1>([&](auto ic) { st_a<Tr, decltype(ic)::value>(s, acc.lsum, acc.pk); });
}

// One pipelined half-step of 32 gaps (see above):
//   sN <- scores of K rows [32*HN, +32) of the tile at kbuf, both query blocks          (gaps 0-15)
//   sO  = scores of keys [32*HO, +32) of the tile whose V is at vbuf, in the entry state: finished,
//         O^T += V^T . P^T                                                              (gaps 16-31)
//   and sN is left in the entry state for the next half-step.
// mask_n bit q: sN[q] holds keys that must be masked (diagonal / ragged tiles); kbase_n = their first key.
// kpre[PF] in: the first PF K fragments of this half-step; out (PREF): those of the next one (rows
// [32*PH, +32) of the tile at kbuf_pref).  hook(n): extra work for gap n (the LDS-DMA pieces of H2).
// ABL (diagnostic builds only, results wrong by construction): 4 = no softmax stages, 8 = no LDS fragment reads.
template <class Tr, int D, int PF, int ORD, int HN, int HO, bool PREF, int PH, int ABL = 0, class Hook = NoHook>
__device__ __forceinline__ void hstep(const lds_char *lds, unsigned k_e, unsigned v_e, int kbuf, int vbuf, int kbuf_pref,
                                      const typename Tr::mfma_vec (&qf)[2][D / 16], f32x16 (&sN)[2], f32x16 (&sO)[2],
                                      Acc<D> &acc, int &pend, float c2, int mask_n, int kbase_n, int h2,
                                      const int (&lim)[2], typename Tr::mfma_vec (&kpre)[PF], const Hook &hook = Hook()) {
    using Vec = typename Tr::mfma_vec;
    constexpr int NKS = D / 16, NDB = D / 32;
    constexpr int RG = Img<D>::RG;
    constexpr bool PS = (ORD == 6);
    static_assert(NKS == 8 && NDB == 4 && PF == 8, "the gap program below is written for head_dim 128");

    const lds_char *const vb_0 = lds + (v_e + vbuf), *const vb_1 = lds + ((v_e ^ 32) + vbuf);
    const lds_char *const kp_e = lds + (k_e + kbuf_pref), *const kp_o = lds + ((k_e ^ 32) + kbuf_pref);
    auto ld_kp = [&](int ks) -> Vec {
        return bitcast<Vec>(lds_read16(((ks & 1) ? kp_o : kp_e) + 4 * RG * PH + 512 * (ks >> 1)));
    };
    // transposed read e (0 / 1) of V fragment j (A operand of the PV MFMAs of d block j % 4, k-step j / 4)
    auto ld_vt = [&](int j, int e) -> u32x2 {
        const int d = j % NDB, s = 2 * HO + j / NDB;
        return bitcast<u32x2>(__builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_i16x4 *)((e ? vb_1 : vb_0) + RG * (2 * s + e) + 512 * d)));
    };

    apply_pending<D>(acc, pend);

    Vec kf[NKS];
    u32x2 vlo[2 * NDB], vhi[2 * NDB];
#pragma unroll
    for (int i = 0; i < PF; ++i) kf[i] = kpre[i];
    asm volatile("" :: "v"(kf[NKS - 1]));   // one wait for all eight K fragments (read >= 8 gaps ago), see the gap loop
    float m0, m1;                           // lane max of the new scores, q0 / q1 (first written in gaps 9 / 17)
    static_for<32>([&](auto ic) {
        constexpr int n = decltype(ic)::value;
        // ---- the MFMA of this gap ----
        if constexpr (n < 16) {
            constexpr int q = n >> 3, ks = n & 7;
            if constexpr (ks == 0) {
                if constexpr (PS) mfma_qk_first_c<Tr>(sN[q], kf[0], qf[q][0], acc.cinit[q]);
                else mfma_qk_first<Tr>(sN[q], kf[0], qf[q][0]);
            } else {
                mfma_qk<Tr>(sN[q], kf[ks], qf[q][ks]);
            }
        } else {
            constexpr int blk = (n - 16) >> 2, d = (n - 16) & 3, q = blk & 1, ks = blk >> 1, j = NDB * ks + d;
            u32x4 av, pv;
            av[0] = vlo[j][0]; av[1] = vlo[j][1]; av[2] = vhi[j][0]; av[3] = vhi[j][1];
            pv[0] = acc.pk[q][4 * ks + 0]; pv[1] = acc.pk[q][4 * ks + 1];
            pv[2] = acc.pk[q][4 * ks + 2]; pv[3] = acc.pk[q][4 * ks + 3];
            mfma_pv<Tr, false>(acc.o[q][d], bitcast<Vec>(av), bitcast<Vec>(pv));
        }
        // ---- one LDS read ----
        if constexpr (ABL & 8) {
            if constexpr (n == 0) {
#pragma unroll
                for (int i = PF; i < NKS; ++i) kf[i] = kf[i & 1];
#pragma unroll
                for (int i = 0; i < 2 * NDB; ++i) { vlo[i] = bitcast<u32x4>(kf[0]).xy; vhi[i] = bitcast<u32x4>(kf[1]).xy; }
            }
        } else if constexpr (n < 16) {
            if constexpr (n & 1) vhi[n >> 1] = ld_vt(n >> 1, 1);
            else vlo[n >> 1] = ld_vt(n >> 1, 0);
        } else if constexpr (n < 16 + NKS && PREF) kpre[n - 16] = ld_kp(n - 16);
        // One s_waitcnt per batch of fragments instead of one per MFMA (each costs an issue slot of the gap):
        // naming the YOUNGEST read of a batch makes hipcc wait for the whole batch here, and every batch was
        // issued at least eight gaps ago.
        if constexpr (!(ABL & 8)) {
            if constexpr (n == 15) asm volatile("" :: "v"(vhi[NDB - 1]));
            if constexpr (n == 23) asm volatile("" :: "v"(vhi[2 * NDB - 1]));
        }
        // ---- softmax stages of the half-tile being consumed ----
        if constexpr (!(ABL & 4)) {
        if constexpr (n == 0) st_a<Tr, kLead - 1>(sO, acc.lsum, acc.pk);
        st_f<Tr, ORD, n + kLead + 1>(sO, acc.msafe, c2);
        st_x<n + kLead>(sO);
        if constexpr (n >= 1) st_a<Tr, n + kLead - 1>(sO, acc.lsum, acc.pk);
        }
        // ---- row max of the new scores, and their lead stages ----
        if constexpr (n == 9) { if (mask_n & 1) mask_keys(sN[0], kbase_n, h2, lim[0]); }     // wave-uniform, diagonal / ragged tiles only
        if constexpr (n == 9) st_max2(m0, sN[0][0], sN[0][1]);
        if constexpr (n > 9 && n <= 16) st_max3(m0, sN[0][2 * (n - 9)], sN[0][2 * (n - 9) + 1]);
        if constexpr (n == 17) {
            decide<Tr, D, ORD>(acc, 0, sN[0], m0, c2, pend);
            if (mask_n & 2) mask_keys(sN[1], kbase_n, h2, lim[1]);
        }
        if constexpr (n == 17) st_max2(m1, sN[1][0], sN[1][1]);
        if constexpr (n > 17 && n <= 24) st_max3(m1, sN[1][2 * (n - 17)], sN[1][2 * (n - 17) + 1]);
        if constexpr (n == 25) decide<Tr, D, ORD>(acc, 1, sN[1], m1, c2, pend);
        if constexpr (!(ABL & 4)) {
        if constexpr (n >= 23) st_f<Tr, ORD, n - 23>(sN, acc.msafe, c2);         // elements 0..8
        if constexpr (n >= 24) st_x<n - 24>(sN);                                 // elements 0..7
        if constexpr (n >= 25) st_a<Tr, n - 25>(sN, acc.lsum, acc.pk);           // elements 0..6
        }
        hook(n);
        SFA_FENCE();
    });
}

// Last half-step of a q-tile for this wave: no new scores.  Finishes sO (entry state) and adds its P.V.
template <class Tr, int D, int ORD, int HO>
__device__ __forceinline__ void hstep_last(const lds_char *lds, unsigned v_e, int vbuf, f32x16 (&sO)[2], Acc<D> &acc,
                                           int &pend, float c2) {
    using Vec = typename Tr::mfma_vec;
    constexpr int NDB = D / 32;
    constexpr int RG = Img<D>::RG;
    const lds_char *const vb_0 = lds + (v_e + vbuf), *const vb_1 = lds + ((v_e ^ 32) + vbuf);
    apply_pending<D>(acc, pend);
    Vec vf[2 * NDB];
#pragma unroll
    for (int j = 0; j < 2 * NDB; ++j) {
        const int d = j % NDB, s = 2 * HO + j / NDB;
        const u32x2 lo = bitcast<u32x2>(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(vb_0 + RG * (2 * s) + 512 * d)));
        const u32x2 hi = bitcast<u32x2>(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(vb_1 + RG * (2 * s + 1) + 512 * d)));
        u32x4 av;
        av[0] = lo[0]; av[1] = lo[1]; av[2] = hi[0]; av[3] = hi[1];
        vf[j] = bitcast<Vec>(av);
    }
    st_a<Tr, kLead - 1>(sO, acc.lsum, acc.pk);
    static_for<32>([&](auto ic) { if constexpr (decltype(ic)::value > kLead) st_f<Tr, ORD, decltype(ic)::value>(sO, acc.msafe, c2); });
    static_for<32>([&](auto ic) { if constexpr (decltype(ic)::value >= kLead) st_x<decltype(ic)::value>(sO); });
    static_for<32>([&](auto ic) { if constexpr (decltype(ic)::value >= kLead) st_a<Tr, decltype(ic)::value>(sO, acc.lsum, acc.pk); });
    static_for<16>([&](auto ic) {
        constexpr int m = decltype(ic)::value, blk = m >> 2, d = m & 3, q = blk & 1, ks = blk >> 1;
        u32x4 pv;
        pv[0] = acc.pk[q][4 * ks + 0]; pv[1] = acc.pk[q][4 * ks + 1];
        pv[2] = acc.pk[q][4 * ks + 2]; pv[3] = acc.pk[q][4 * ks + 3];
        mfma_pv<Tr, m == 0>(acc.o[q][d], vf[NDB * ks + d], bitcast<Vec>(pv));
    });
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
    int live;           // (int: a struct copy with padding bytes goes through scratch)
};

template <class Tr, int D, bool CAUSAL, int ORD, int RING, int DIAG, int DMA_AT = 0>
__global__ void __launch_bounds__(w4::kThreadsW4, 1)
prefill_w4r2_kernel(const PrefillKernelParams p) {
    using namespace w4;
    using Vec = typename Tr::mfma_vec;
    constexpr int NQB = 2, PF = 8;
    constexpr bool PS = (ORD == 6);
    constexpr int NKS = D / 16, NDB = D / 32;
    constexpr int NJ = D / 64;                  // 128-byte column pieces per row
    using L = Img<D, RING>;
    constexpr int NDMA = 4 * NJ;                // LDS-DMA pieces a wave issues per step (K tile + V tile)
    static_assert(RING == 3 || RING == 4, "ring depth");
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
                c.qt = ((c.sub == 0) != (CAUSAL && (DIAG & 4096) != 0)) ? heavy : c.i;    // (diagnostic 4096: the light q-tile of a pair first)
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
    // the buffer descriptor of a tile; `live` false (the producer has run out of tiles): zero bytes, so the
    // pieces still issue -- no branch in the MFMA gaps -- and simply zero-fill a ring slot nobody will read
    auto make_srd = [&](const Desc &d, bool live) -> u32x4s {
        u32x4s srd;
        srd[0] = d.lo;
        srd[1] = d.hi & 0xffffu;
        srd[2] = (live && !(DIAG & 1024)) ? (unsigned)max(d.left, 0) : 0u;     // (diagnostic 1024: every piece out of bounds -- issued, zero-filled, nothing fetched)
        srd[3] = 0x00020000u;
        return srd;
    };
    // piece idx (0 .. 2*NJ-1) of this wave's share of a tile: row group idx / NJ, column piece idx % NJ
    auto issue_piece = [&](const u32x4s &srd, bool is_k, int ring_off, int idx) {
        const int half = idx / NJ, j = idx % NJ;
        const unsigned dst = lds0 + (is_k ? L::K_BASE : L::V_BASE) + ring_off + 2 * wave * L::RG;
        if (DIAG & 2) return;                   // timing-only ablation: no LDS-DMA
        if (DIAG & 32) {                        // timing-only ablation: everything but the load itself
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dst + half * L::RG + 1024 * j), "v"(kvoff0), "s"(srd), "s"(128u * j) : "memory");
            return;
        }
        dma_piece(dst + half * L::RG + 1024 * j, is_k ? (half ? kvoff1 : kvoff0) : (half ? vvoff1 : vvoff0), srd, 128u * j);
    };
    auto issue_tile = [&](const Desc &d, bool is_k, int ring_off) {
        const u32x4s srd = make_srd(d, true);
#pragma unroll
        for (int idx = 0; idx < 2 * NJ; ++idx) issue_piece(srd, is_k, ring_off, idx);
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
    bool thin_tail = false;                     // the last produce_k() had nothing left to issue
    if (pc.live) { kd = desc_at_head(true, pc.b, pc.h); vd = desc_at_head(false, pc.b, pc.h); }
    int kring_p = 0, vring_p = 0;               // ring byte offsets the producers write next
    auto ring_next = [](int x) -> int { return x == (RING - 1) * L::TILE ? 0 : x + L::TILE; };
    auto produce_v = [&]() {                    // V of the stream position K produced last time
        if (vpend_live) issue_tile(vpend, false, vring_p);
        vring_p = kring_p;
    };
    // the same, one piece per call (spread over the MFMA gaps of H2): V pieces first, then K pieces
    u32x4s piece_srd = {0, 0, 0, 0};            // descriptor of the tile whose pieces are being dealt out
    auto produce_v_piece = [&](int idx) {
        if (idx == 0) piece_srd = make_srd(vpend, vpend_live);
        issue_piece(piece_srd, false, vring_p, idx);
        if (idx == 2 * NJ - 1) vring_p = kring_p;
    };
    auto k_advance = [&]() {
        vpend = vd;
        if (++pc.t < pc.nt) {
            desc_advance(kd, k_tileb);
            desc_advance(vd, v_tileb);
        } else {
            next_item(pc, true);
            if (pc.live) { kd = desc_at_head(true, pc.b, pc.h); vd = desc_at_head(false, pc.b, pc.h); }
        }
    };
    auto produce_k_piece = [&](int idx) {
        if (idx == 0) piece_srd = make_srd(kd, pc.live);
        issue_piece(piece_srd, true, kring_p, idx);
        if (idx == 2 * NJ - 1) {
            vpend_live = pc.live;
            thin_tail = !pc.live;
            if (pc.live) k_advance();
            kring_p = ring_next(kring_p);
        }
    };
    auto produce_k = [&]() {
        vpend_live = pc.live;
        thin_tail = !pc.live;
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
    // Before the barrier of a step the pieces issued behind the PREVIOUS barrier must have landed.  Ring 3:
    // those are the youngest ones -> vmcnt(0).  Ring 4: one more step's pieces may stay in flight ->
    // vmcnt(NDMA), as long as that younger step really issued all of its pieces (it does not once the
    // producer has run out of tiles: then drain).  Ops hipcc issues in between (Q loads, O stores) are
    // younger than the pieces waited for, so they only make the wait stricter.
    auto wait_and_sync = [&]() {
        if (DIAG & 16) return;                  // timing-only ablation: no wait, no barrier
        if (DIAG & 64) { asm volatile("s_barrier" ::: "memory"); return; }     // timing-only ablation: barrier, no wait for the pieces
        if (RING == 3 || thin_tail) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NDMA) : "memory");
    };
    // DIAG: one workgroup stamps s_memtime at five points of steps 8..15 of its first item into p.lse (as
    // u64[wave][step][8]); the stamp drains lgkmcnt, so read SHARES from it, not absolute speed.
    int item_no = 0;                            // q-tiles this workgroup has finished
    auto stamp = [&](int step, int which) {     // steps 8..15 of the first q-tile and steps 0..7 of the second
        if ((DIAG & 1) && blockIdx.x == 8 && p.lse && ((item_no == 0 && step >= 8 && step < 16) || (item_no == 1 && step < 8))) {
            unsigned long long tm;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) :: "memory");
            if (lane == 0) reinterpret_cast<unsigned long long *>(p.lse)[((item_no * 4 + wave) * 8 + (step & 7)) * 8 + which] = tm;
        }
    };
    // stream prologue: K(0), K(1), V(0) must be visible before the first step; the rest in flight
    produce_k(); produce_v(); produce_k(); produce_v(); produce_k();
    if (RING == 4) { produce_v(); produce_k(); }
    thin_tail = !pc.live && !vpend_live;
    if (RING == 3 || thin_tail) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(2 * NDMA) : "memory");

    int kcur = 0, vcur = 0;                     // ring byte offsets of the compute's current tile
#define SFA_W4_SYNC_AND_STAGE()                                                                     \
    do {                                                                                            \
        wait_and_sync();                                                                            \
        produce_v();                                                                                \
        produce_k();                                                                                \
    } while (0)

    // Q rows reach the accumulator file through LDS: a lane pair owns ONE query row, so loading the operand layout
    // straight from memory touches 32 rows with 32 B each per instruction and every cache line eight times -- the 16
    // loads of a wave took 1-2k cycles to ISSUE and the address path was busy with them for most of the ~8k-cycle
    // epilogue they were meant to hide under (q-tile stamps).  As LDS-DMA pieces in the K image (8 rows x 128 B per
    // piece, a quarter of the line visits) into a wave-private 64-row image, read back like K fragments.
    // request_q: 16 pieces, issued where load_q() was; fetch_q: when the rows are needed.  Rows past Sq read as zeros.
    Vec qf[NQB][NKS];
    const unsigned q_rowb = (unsigned)(2 * p.qs[2]);
    const unsigned qvoff0 = (unsigned)r8 * q_rowb + 64u * dsub + 16u * (dslot ^ (r8 >> 2));
    const unsigned qvoff1 = (unsigned)(r8 + 8) * q_rowb + 64u * dsub + 16u * (dslot ^ (2 + (r8 >> 2)));
    auto load_q = [&](int b, int h, int qt) {
        const int row0 = qt * kRows + 64 * wave;
        const unsigned long long base = (unsigned long long)(uintptr_t)(p.q + b * p.qs[0] + h * p.qs[1]) + (unsigned long long)row0 * q_rowb;
        u32x4s srd;
        srd[0] = (unsigned)base;
        srd[1] = (unsigned)(base >> 32) & 0xffffu;
        srd[2] = row0 < p.Sq ? (unsigned)(p.Sq - 1 - row0) * q_rowb + 2u * D : 0u;
        srd[3] = 0x00020000u;
        const unsigned dst = lds0 + L::Q_BASE + wave * L::TILE;
#pragma unroll
        for (int rg = 0; rg < 8; ++rg)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                dma_piece(dst + rg * L::RG + 1024 * j, (rg & 1) ? qvoff1 : qvoff0, srd, 128u * j + 16u * (rg >> 1) * q_rowb);
    };
    const unsigned q_e = L::Q_BASE + L::RG * (l31 >> 3) + 64 * (l31 & 7) + 16 * (h2 ^ ((l31 >> 2) & 3));
    auto fetch_q = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the pieces (and whatever this wave stored since)
        const lds_char *const qe = lds + (q_e + wave * L::TILE), *const qo = lds + ((q_e ^ 32) + wave * L::TILE);
#pragma unroll
        for (int q = 0; q < NQB; ++q)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                qf[q][ks] = bitcast<Vec>(lds_read16(((ks & 1) ? qo : qe) + 4 * L::RG * q + 512 * (ks >> 1)));
    };
    // prescaled flavour: fold scale * log2(e) into Q once per q-tile, when the rows are first needed (not where
    // they are requested: that would wait for the loads on the spot)
    auto prescale_q = [&]() {
        if (!PS) return;
#pragma unroll
        for (int q = 0; q < NQB; ++q)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                u32x4 w = bitcast<u32x4>(qf[q][ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = Tr::pack2(Tr::lo_f32(w[i]) * c2, Tr::hi_f32(w[i]) * c2);
                qf[q][ks] = bitcast<Vec>(w);
            }
    };

    constexpr bool QPRE = !(DIAG & 128);        // (diagnostic 128: Q rows loaded at the start of their own q-tile)
    if (cc.live && QPRE) load_q(cc.b, cc.h, cc.qt);
    // DIAG 256: workgroup 8 stamps six points of each of its first 16 q-tiles (kept in scalars, stored behind the
    // q-tile's epilogue) into p.lse as u64[wave][item][8]: 0 start, 1 Q rows in registers, 2 first half-tile scored
    // and led in, 3 end of the full steps, 4 end of the tail and idle steps, 5 end of the epilogue; [6] = ntw | nt << 32,
    // [7] = the q-tile in s_memrealtime ticks (100 MHz; cycles / ticks = the shader clock the q-tile ran at)
    unsigned long long its[6] = {0, 0, 0, 0, 0, 0}, rt0 = 0, rt1 = 0;
    auto istamp = [&](int which) {
        if ((DIAG & 256) && blockIdx.x == 8) {
            unsigned long long tm, rt;
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm), "=s"(rt) :: "memory");
            its[which] = tm;
            if (which == 0) rt0 = rt;
            if (which == 5) rt1 = rt;
        }
    };
    while (cc.live) {
        const int qt = cc.qt, nt = cc.nt;
        const int b = cc.b, h = cc.h;
        istamp(0);
        if (!QPRE) load_q(b, h, qt);
        fetch_q();
        prescale_q();
        W4Cursor nx;                            // the item after this one (set where its Q rows are requested)
        // Q^T sits in the accumulator file (written from the LDS image just now); two wait states before the first MFMA
#pragma unroll
        for (int q = 0; q < NQB; ++q)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) asm volatile("s_nop 1" : "+a"(qf[q][ks]));
        istamp(1);
        const int wq0 = qt * kRows + 64 * wave;                 // this wave's first query row
        int ntw = nt;                                           // tiles this wave computes on (wave-uniform)
        // (stopping every wave at its own diagonal tile and letting it idle at the barriers, against running all
        // four to the q-tile's last tile over fully masked scores -- DIAG 512: the same exact-scale, 3 % slower prescaled)
        if (CAUSAL && !(DIAG & 512)) ntw = (wq0 + 63 + coff >= 0) ? min(nt, (wq0 + 63 + coff) / kKeys + 1) : 0;
        int lim[NQB];                                           // last visible key of this lane's rows
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            const int qrow = wq0 + 32 * q + l31;
            lim[q] = CAUSAL ? min(p.Sk - 1, qrow + coff) : p.Sk - 1;
        }
        // bit q set: the 32 keys starting at kbase need masking for query block q (wave-uniform): kbase lies
        // beyond the last half-tile that block sees whole -- one threshold per block and q-tile
        int whole[NQB];
#pragma unroll
        for (int q = 0; q < NQB; ++q) whole[q] = CAUSAL ? min(wq0 + 32 * q + coff - 31, p.Sk - 32) : p.Sk - 32;
        auto mask_bits = [&](int kbase) -> int {
            int m = 0;
#pragma unroll
            for (int q = 0; q < NQB; ++q)
                if (kbase > whole[q]) m |= 1 << q;
            return m;
        };

        Acc<D> acc;
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc.o[q][d][r] = 0.f;
            acc.msc[q] = PS ? 0.f : ninf();
            acc.msafe[q] = 0.f;
            acc.thr[q] = ninf();
            acc.lsum[q] = 0.f;
            acc.alpha[q] = 1.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc.pk[q][i] = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc.cinit[q][r] = 0.f;
        }
        int pend = 0;                                           // a rescale of O is parked in acc.alpha (wave-uniform)

        // ---- scores of the first half-tile (outside the pipeline), first fragments of the second ----
        f32x16 sA[NQB], sB[NQB];
#pragma unroll
        for (int q = 0; q < NQB; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) { sA[q][r] = 0.f; sB[q][r] = 0.f; }
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
            lead_in<Tr, D, ORD>(acc, sA, c2, mask_bits(0), h2, lim);
        }

        int t = 0;
        istamp(2);
        // ---- FULL steps: this wave needs the next tile as well.  A and B are the score registers of the two
        // 32-key halves of a tile; each half-step computes one and consumes the other:
        //   H1(t): S(B_t) = K(t)[32:64] Q^T     || softmax(A_t), O += P(A_t) V(t)[0:32]
        //   barrier(t)            -- K(t+2), V(t+1) visible; the slots of K(t), V(t-1) free
        //   H2(t): S(A_t+1) = K(t+1)[0:32] Q^T  || softmax(B_t), O += P(B_t) V(t)[32:64] || LDS-DMA of K(t+3), V(t+2)
        for (; t + 1 < ntw; ++t) {
            const int k1 = ring_next(kcur);
            const int kbase = t * kKeys;
            stamp(t, 0);
            hstep<Tr, D, PF, ORD, 1, 0, true, 0, DIAG & 12>(lds, k_e, v_e, kcur, vcur, k1, qf, sB, sA, acc, pend, c2,
                                                 mask_bits(kbase + 32), kbase + 32, h2, lim, kpre);
            stamp(t, 1);
            if (DIAG & 1) {
                if (RING == 3 || thin_tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
                stamp(t, 2);
            }
            wait_and_sync();
            stamp(t, 3);
            auto dma_hook = [&](int n) {                        // one piece per gap, from gap DMA_AT on
                if (n >= DMA_AT && n < DMA_AT + 2 * NJ) produce_v_piece(n - DMA_AT);
                else if (n >= DMA_AT + 2 * NJ && n < DMA_AT + 4 * NJ) produce_k_piece(n - DMA_AT - 2 * NJ);
            };
            hstep<Tr, D, PF, ORD, 0, 1, true, 1, DIAG & 12>(lds, k_e, v_e, k1, vcur, k1, qf, sA, sB, acc, pend, c2,
                                                 mask_bits(kbase + 64), kbase + 64, h2, lim, kpre, dma_hook);
            stamp(t, 4);
            kcur = k1;
            vcur = ring_next(vcur);
        }
        istamp(3);
        // ---- TAIL step: this wave's last tile (its second half computes no new scores) ----
        if (t < ntw) {
            const int kbase = t * kKeys;
            hstep<Tr, D, PF, ORD, 1, 0, false, 0>(lds, k_e, v_e, kcur, vcur, kcur, qf, sB, sA, acc, pend, c2,
                                                  mask_bits(kbase + 32), kbase + 32, h2, lim, kpre);
            SFA_W4_SYNC_AND_STAGE();
            hstep_last<Tr, D, ORD, 1>(lds, v_e, vcur, sB, acc, pend, c2);
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

        istamp(4);
        // The next q-tile's Q rows are requested HERE, behind this wave's last barrier of the q-tile: their HBM
        // latency (~3.5k cycles, q-tile stamps) then hides under the epilogue (~4k cycles).  Any earlier and the
        // loads sit in front of the LDS-DMA pieces in the wave's in-order vector-memory queue, so the next
        // barrier's wait for the pieces waits for Q as well (measured: the whole workgroup then stalls there).
        nx = cc;
        next_item(nx, false);
        if (nx.live && QPRE) load_q(nx.b, nx.h, nx.qt);
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
                if (!(DIAG & 257) && p.lse && h2 == 0) {
                    const float lse = ltot > 0.f ? (acc.msc[q] + __log2f(ltot)) * kLn2 : ninf();
                    p.lse[((long long)b * p.Hq + h) * p.Sq + qrow] = lse;
                }
            }
        }
        istamp(5);
        if ((DIAG & 256) && blockIdx.x == 8 && item_no < 16 && p.lse && lane == 0) {
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(p.lse) + (wave * 16 + item_no) * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) dst[i] = its[i];
            dst[6] = (unsigned long long)ntw | ((unsigned long long)nt << 32);
            dst[7] = rt1 - rt0;
        }
        ++item_no;
        cc = nx;
    }
    // drain: DMA pieces issued for stream positions nobody consumes do not exist (the producers stop at the
    // end of the list), but the last steps' pieces must have landed before the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef SFA_W4_SYNC_AND_STAGE
}

template <class Tr, int D, int ORD, int RING, int DIAG, int DMA_AT = 0>
int launch_w4_t(const PrefillKernelParams &p, bool causal, hipStream_t stream) {
    using namespace w4;
    const int lds = Img<D, RING>::TOTAL;
    // one workgroup per CU, fewer when the XCD lists are shorter than 32 units
    const int nq = (p.Sq + kRows - 1) / kRows;
    const long long units_xcd = (long long)p.bh_per_xcd * (causal ? (nq + 1) / 2 : nq);
    const int nslots = (int)(units_xcd < 32 ? units_xcd : 32);
    dim3 grid(8u * nslots), block(kThreadsW4);
    static DynLdsAttr attr_c, attr_f;
    if (const int rc = causal ? attr_c.ensure(reinterpret_cast<const void *>(&prefill_w4r2_kernel<Tr, D, true, ORD, RING, DIAG, DMA_AT>), lds,
                                              "prefill_w4r2_kernel")
                              : attr_f.ensure(reinterpret_cast<const void *>(&prefill_w4r2_kernel<Tr, D, false, ORD, RING, DIAG, DMA_AT>), lds,
                                              "prefill_w4r2_kernel"))
        return rc;
    if (causal) hipLaunchKernelGGL((prefill_w4r2_kernel<Tr, D, true, ORD, RING, DIAG, DMA_AT>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((prefill_w4r2_kernel<Tr, D, false, ORD, RING, DIAG, DMA_AT>), grid, block, lds, stream, p);
    return check_launch("prefill_w4r2_kernel");
}

}  // namespace

constexpr int kW4Ring = 3;          // shipped LDS ring depth (3 or 4)

// force: 0 = flavour by policy (exact unless the caller opted into fast_scale), 1 = prescaled, 2 = exact
int launch_prefill_w4r2(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream, int force) {
    if (dtype != SFA_DTYPE_FP16 && dtype != SFA_DTYPE_BF16)
        return fail(SFA_ERR_BAD_DTYPE, "sfa_prefill_fwd: dtype %d is not fp16(0)/bf16(1)", dtype);
    if (head_dim != 128)
        return fail(SFA_ERR_UNSUPPORTED_HEAD_DIM, "sfa_prefill_fwd: the 4-wave kernel serves head_dim 128 (got %d)", head_dim);
    // the K/V rows of one head are addressed through a 32-bit buffer descriptor
    const long long k_ext = (long long)(p.Sk - 1) * 2 * p.ks[2] + 2 * head_dim, v_ext = (long long)(p.Sk - 1) * 2 * p.vs[2] + 2 * head_dim;
    const long long q_ext = (long long)(p.Sq - 1) * 2 * p.qs[2] + 2 * head_dim;
    if (k_ext >= (1ll << 31) || v_ext >= (1ll << 31) || q_ext >= (1ll << 31) || p.ks[2] * 2 >= (1ll << 24) || p.vs[2] * 2 >= (1ll << 24) ||
        p.qs[2] * 2 >= (1ll << 24))
        return fail(SFA_ERR_BAD_SHAPE, "sfa_prefill_fwd: one head's Q/K/V rows span more than 2 GiB");
    // force 3 / 4: the other ring depth / the stamping build (bf16, exact) -- A/B and diagnostics only
    if (force == 3) return launch_w4_t<Bf16, 128, 2, 7 - kW4Ring, 0>(p, causal, stream);
    if (force == 4) return launch_w4_t<Bf16, 128, 2, kW4Ring, 1>(p, causal, stream);
    if (force == 16) return launch_w4_t<Bf16, 128, 2, kW4Ring, 256>(p, causal, stream);     // q-tile level stamps
    if (force == 19) return launch_w4_t<Bf16, 128, 2, kW4Ring, 384>(p, causal, stream);     // q-tile stamps without Q prefetch
    if (force == 17) return launch_w4_t<Bf16, 128, 2, kW4Ring, 512>(p, causal, stream);     // all waves run to the last tile
    if (force == 18) return launch_w4_t<Bf16, 128, 6, kW4Ring, 512>(p, causal, stream);
#ifdef SFA_WITH_VARIANTS      // A/B orderings and timing-only ablations (results of the latter wrong by construction): the A/B library only
    if (force == 30) return launch_w4_t<Bf16, 128, 2, kW4Ring, 4096>(p, causal, stream); // the light q-tile of a causal pair first (correct results)
    if (force == 5) return launch_w4_t<Bf16, 128, 2, kW4Ring, 2>(p, causal, stream);     // no LDS-DMA
    if (force == 6) return launch_w4_t<Bf16, 128, 2, kW4Ring, 4>(p, causal, stream);     // no softmax stages
    if (force == 7) return launch_w4_t<Bf16, 128, 2, kW4Ring, 8>(p, causal, stream);     // no LDS fragment reads
    if (force == 8) return launch_w4_t<Bf16, 128, 2, kW4Ring, 16>(p, causal, stream);    // no barrier
    if (force == 9) return launch_w4_t<Bf16, 128, 2, kW4Ring, 30>(p, causal, stream);    // MFMAs only
    if (force == 10) return launch_w4_t<Bf16, 128, 2, 3, 0, 24>(p, causal, stream);      // DMA pieces in the last gaps of H2
    if (force == 11) return launch_w4_t<Bf16, 128, 2, kW4Ring, 32>(p, causal, stream);   // no load instruction, all else kept
    if (force == 12) return launch_w4_t<Bf16, 128, 2, kW4Ring, 64>(p, causal, stream);   // no wait for the pieces
    if (force == 14) return launch_w4_t<Bf16, 128, 2, kW4Ring, 128>(p, causal, stream);  // next q-tile's Q rows NOT prefetched
    if (force == 15) return launch_w4_t<Bf16, 128, 6, kW4Ring, 128>(p, causal, stream);  // the same, prescaled flavour
    if (force == 13) return launch_w4_t<Bf16, 128, 2, kW4Ring, 80>(p, causal, stream);   // pieces issued, never waited for, no barrier
    if (force == 20) return launch_w4_t<Bf16, 128, 2, kW4Ring, 1024>(p, causal, stream); // pieces issued with empty descriptors: no memory traffic
    if (force == 21) return launch_w4_t<Bf16, 128, 2, kW4Ring, 256 + 1024>(p, causal, stream);  // q-tile stamps (cycles AND clock) of the ablations:
    if (force == 22) return launch_w4_t<Bf16, 128, 2, kW4Ring, 256 + 2>(p, causal, stream);     //   a shorter time can be a faster clock, not
    if (force == 23) return launch_w4_t<Bf16, 128, 2, kW4Ring, 256 + 4>(p, causal, stream);     //   fewer cycles (zeros in LDS draw less power)
    if (force == 24) return launch_w4_t<Bf16, 128, 2, kW4Ring, 256 + 30>(p, causal, stream);
    if (force == 25) return launch_w4_t<Bf16, 128, 2, kW4Ring, 256 + 32>(p, causal, stream);
    if (force == 26) return launch_w4_t<Bf16, 128, 2, kW4Ring, 256 + 16>(p, causal, stream);
    if (force == 27) return launch_w4_t<Bf16, 128, 2, kW4Ring, 256 + 64>(p, causal, stream);
    if (force == 28) return launch_w4_t<Bf16, 128, 2, kW4Ring, 256 + 8>(p, causal, stream);
    if (force == 29) return launch_w4_t<Bf16, 128, 6, kW4Ring, 256>(p, causal, stream);         // prescaled flavour, stamped
#endif
    const bool prescaled = force == 0 ? p.fast_scale != 0 : force == 1;
    if (dtype == SFA_DTYPE_FP16)
        return prescaled ? launch_w4_t<Fp16, 128, 6, kW4Ring, 0>(p, causal, stream)
                         : launch_w4_t<Fp16, 128, 2, kW4Ring, 0>(p, causal, stream);
    return prescaled ? launch_w4_t<Bf16, 128, 6, kW4Ring, 0>(p, causal, stream)
                     : launch_w4_t<Bf16, 128, 2, kW4Ring, 0>(p, causal, stream);
}

}  // namespace sfa
