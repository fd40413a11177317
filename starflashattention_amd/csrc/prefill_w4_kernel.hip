// Fused attention forward (prefill) for gfx950 (MI355X): the 4-wave, one-wave-per-SIMD, persistent
// kernel, round 3: ONE continuous half-step pipeline across q-tiles.  bf16 / fp16, head_dim 128, causal or
// full, MHA or GQA, any strides.  The reference has no prefill kernel; this is the kernel BASELINE.json's
// headline metric is quoted on (SURVEY.md 8(a) A-new).
//
// Structure (unchanged from round 2, prefill_w4r2_kernel.hip -- kept in the A/B library):
//   * workgroup = 4 waves = one 256-row q-tile; a wave owns 64 query rows = two 32-row query blocks, so every
//     K / V fragment read from LDS feeds TWO MFMAs; one wave per SIMD with the whole 512-entry register file.
//   * O^T (128 registers) and the Q^T fragments (64) live in the ACCUMULATOR half of the register file, by NAME
//     (a[0:127], a[128:191]: "the asm-owned half of the register file" below).  Scores, P, the K / V fragments, the
//     masks and the softmax state stay in the 256 arch VGPRs.
//   * S^T = K . Q^T and O^T += V^T . P^T with v_mfma_f32_32x32x16: the query sits on the lane in both
//     accumulators, the exponentiated S^T registers ARE the B operand of the PV product, V^T comes out of
//     the row-major tile through ds_read_b64_tr_b16.
//   * K / V tiles (64 keys) arrive by LDS-DMA into 3-deep rings, K three tiles and V two tiles ahead of the
//     compute, ONE barrier per tile; the LDS image is the 8-row x 32-column sub-tiled, XOR-swizzled one of
//     cdna_hip_programming.md T10(a) (the swizzle sits in the per-lane SOURCE address).
//   * 256 persistent workgroups walk a static, XCD-aware list of q-tiles (causal: balanced pairs of one head);
//     the DMA producers run ahead of the compute across q-tile and head boundaries.
//   * a half-step = 32 GAPS of one MFMA each, the softmax of a half-tile dealt out as per-element stages.
//
// What round 3 changes: the q-tile SEAM.  Round 2 scored the first half-tile of a q-tile outside the pipeline,
// finished the last one with an unpipelined tail, advanced its cursors with integer divisions and ran the epilogue
// of all four waves behind the last barrier: 9-10 k cycles per q-tile (2.8 tile steps), 10.6 % of the causal
// headline launch (gpurun stamps).  Now
//   * the LAST half-step of a q-tile is an ordinary half-step whose QK^T side already scores the first half-tile
//     of the NEXT q-tile (its K tile is simply the next position of the stream, its Q rows were read into the
//     accumulator file behind the last QK^T MFMA that needed the old ones), and whose decision stage sets the new
//     reference maximum outright while the finished q-tile's row sums move to `Fin`;
//   * the first P.V MFMAs of a q-tile start O from a zero C operand (no 128 accumulator writes);
//   * a wave that has passed its causal diagonal finishes with a pipelined consume-only half-step and stores its
//     rows in the steps it would otherwise idle through; only the wave that owns the diagonal's end has an
//     exposed epilogue;
//   * the cursors advance by additions (one division per launch), and the Q rows of the next q-tile are requested
//     two to three steps ahead behind a COUNTED vmcnt, and fetched into the accumulator file in the MFMA gaps of the
//     wave's last half-step that still scores (or in an idle step), not at the top of the q-tile;
//   * masks are the C operand of a block's first QK^T MFMA (the inner tiles' half-steps carry no mask code);
//   * kernel arguments are precomputed on the host (W4Args) and re-read in bursts where a q-tile begins or ends.
// Measurements, and what was tried and not kept: DESIGN.md 5.2 "Round 3", profiles/r03_power_clock.txt.
//
// Hazards hipcc does not see inside the asm MFMAs (cdna_hip_programming.md section 5.7): (1) an MFMA's result
// read or overwritten by the VALU needs the MFMA to have drained -- in the full half-step at least two other
// MFMAs sit between producer and consumer; the half-steps without a P.V side carry a settle() where those MFMAs
// are missing; (2) a VALU-written VGPR used as an MFMA operand needs two wait states -- the packed P registers
// are written at least four gaps before their PV MFMA.  tools/check_mfma_hazards.py checks both on the ISA.
#include "prefill_w4_common.h"

namespace sfa {

namespace {

using namespace prefill;

namespace w4 {
typedef float f32x2 __attribute__((ext_vector_type(2)));

using namespace w4c;

constexpr int kRows = 256;          // query rows per workgroup (q-tile)
constexpr int kKeys = 64;           // keys per K/V tile
constexpr int kRing = 3;            // ring depth (tiles of K, tiles of V)

// LDS image of one [64 keys][D] 16-bit tile: 8-row groups of D/32 sub-tiles of 8 rows x 64 B.
//   off(row, ch) = RG*(row>>3) + 512*(ch>>2) + 64*(row&7) + 16*((ch&3) ^ ((row>>2)&3))      (ch = 16-B chunk of the row)
template <int D> struct Img {
    static constexpr int RG = 512 * (D / 32);       // bytes of one 8-row group
    static constexpr int TILE = 8 * RG;             // 64 rows
    static constexpr int K_BASE = 0;
    static constexpr int V_BASE = kRing * TILE;
    static constexpr int Q_BASE = 2 * kRing * TILE; // the Q rows of the next q-tile, one 64-row image per wave (wave-private)
    static constexpr int TOTAL = Q_BASE + 4 * TILE;
};

// ---- the asm-owned half of the register file ------------------------------------------------------------------
// O^T lives in a[0:127] (query block q, 32-row d block d: a[16 (4 q + d) : +15]) and the Q^T fragments in a[128:191]
// (query block q, k-step ks: a[128 + 4 (8 q + ks) : +3]) for the whole kernel, by NAME: as "a"-constrained C++ values
// hipcc moved them -- new q-tile, new registers, 64 v_accvgpr_mov across a loop's back edge, and with them gone from its
// budget it spilled the half-step loop's working set (a reload costs an s_waitcnt vmcnt(0), which drains the LDS-DMA).
// Every statement that touches them names all 192 as clobbers, so hipcc keeps its own values (spill copies in the
// accumulator file included) out of them wherever an MFMA is near -- i.e. everywhere;
// tests/test_w4_hazards_cpu.py checks on the ISA that no instruction outside an asm statement names a0..a191.
#define SFA_AOWN \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
    "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
    "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
    "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
    "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
    "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
    "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
    "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", \
    "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
    "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", \
    "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", \
    "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191"
constexpr int o_reg(int q, int d) { return 16 * (4 * q + d); }
constexpr int q_reg(int q, int ks) { return 128 + 4 * (8 * q + ks); }

// s (VGPR) = k (VGPR) . q (a[QB:QB+3]) [+ s]
template <class Tr, int QB>
__device__ __forceinline__ void own_qk_first(f32x16 &s, typename Tr::mfma_vec k) {
    if constexpr (Tr::id == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(s) : "v"(k), "n"(QB), "n"(QB + 3) : SFA_AOWN);
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], 0" : "=&v"(s) : "v"(k), "n"(QB), "n"(QB + 3) : SFA_AOWN);
}
template <class Tr, int QB>
__device__ __forceinline__ void own_qk_first_c(f32x16 &s, typename Tr::mfma_vec k, const f32x16 &c) {
    // C operand = a VALU-written register tuple: two wait states in front (hazard (2))
    if constexpr (Tr::id == 1) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %4" : "=&v"(s) : "v"(k), "n"(QB), "n"(QB + 3), "v"(c) : SFA_AOWN);
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], %4" : "=&v"(s) : "v"(k), "n"(QB), "n"(QB + 3), "v"(c) : SFA_AOWN);
}
template <class Tr, int QB>
__device__ __forceinline__ void own_qk(f32x16 &s, typename Tr::mfma_vec k) {
    if constexpr (Tr::id == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(s) : "v"(k), "n"(QB), "n"(QB + 3) : SFA_AOWN);
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], %0" : "+v"(s) : "v"(k), "n"(QB), "n"(QB + 3) : SFA_AOWN);
}
// o (a[OB:OB+15]) += v (VGPR) . p (VGPR)
template <class Tr, int OB>
__device__ __forceinline__ void own_pv(typename Tr::mfma_vec v, typename Tr::mfma_vec pfrag) {
    if constexpr (Tr::id == 1) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(v), "v"(pfrag), "n"(OB), "n"(OB + 15) : SFA_AOWN);
    else asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(v), "v"(pfrag), "n"(OB), "n"(OB + 15) : SFA_AOWN);
}
// o (a[OB:OB+15]) = 0, by the matrix pipe: one instruction instead of sixteen v_accvgpr_write, in a pipe that idles
// during the epilogue anyway.  The next q-tile's P.V products then simply accumulate (no "first product" variant of the
// half-step).  z = four registers of zeros, possibly just written: two wait states (hazard (2)).
template <class Tr, int OB>
__device__ __forceinline__ void own_zero(typename Tr::mfma_vec z) {
    if constexpr (Tr::id == 1) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c1:%c2], %0, %0, 0" :: "v"(z), "n"(OB), "n"(OB + 15) : SFA_AOWN);
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c1:%c2], %0, %0, 0" :: "v"(z), "n"(OB), "n"(OB + 15) : SFA_AOWN);
}
// every MFMA issued so far has drained after 32 wait states (hazard (1)): in front of any VALU access to a0..a191
__device__ __forceinline__ void own_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: SFA_AOWN); }
template <int R>
__device__ __forceinline__ float own_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "n"(R) : SFA_AOWN);
    return x;
}
// a[R] *= f
template <int R>
__device__ __forceinline__ void own_scale(float f) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a%c2\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a%c2, %0" : "=&v"(t) : "v"(f), "n"(R) : SFA_AOWN);
}

// Per-wave online-softmax state of the two query blocks (O^T itself: the asm-owned registers above).
template <int D>
struct Acc {
    float msc[2];               // reference max the exponentials are taken against (log2 units); kNoKey while a row
                                // has seen no key yet (finite, so that -inf scores minus it stay -inf)
    float lsum[2];              // this lane's share of the running row sum
    float alpha[2];             // a rescale of O decided but not yet applied (see hstep); 1 = none
    uint32_t pk[2][8];          // P^T of the half-tile being consumed, packed: pk[q][4k .. 4k+3] = B operand of k-step k
    f32x16 cinit[2];            // prescaled flavour: -msc in all 16 registers (C operand of the first QK^T MFMA)
};
// What the epilogue of a finished q-tile needs besides O, once the running state belongs to the next one.
struct Fin {
    float lsum[2];
    float msc[2];
};

// O of query block q moves to a new reference max.  Rare.
template <int D>
__device__ __forceinline__ void rescale_o(int q, float alpha) {
    own_settle();                           // PV MFMAs of the previous half-step may be in flight
    static_for<16 * (D / 32)>([&](auto ic) {
        constexpr int r = decltype(ic)::value;
        if (q == 0) own_scale<o_reg(0, 0) + r>(alpha);
        else own_scale<o_reg(1, 0) + r>(alpha);
    });
    asm volatile("s_nop 1" ::: SFA_AOWN);   // v_accvgpr_write -> the next MFMA that reads it as C: two wait states
}

// ---- the element pipeline (see prefill_w4r2_kernel.hip for the measurements behind it) ----------------------
// A half-step is 32 GAPS -- one MFMA each -- and the softmax of a half-tile is cut into per-ELEMENT stages:
//     F  s = s * c2 - msc          (exact flavour only)
//     X  s = exp2(s)
//     A  lsum += s; every second element: pack the pair to 16 bit
// The 32 score registers a lane holds for a half-tile (2 query blocks x 16) are walked in the order their PV
// MFMAs need them: element i -> block i>>3 = (k-step, query block) in the order (0,q0) (0,q1) (1,q0) (1,q1),
// register 8*kstep + (i&7).  X of element i runs in gap i - 8, F one gap earlier, A one gap later: the first
// eight elements are exponentiated in the LAST eight gaps of the half-step that computed them.
// MFMA order: QK^T query-block-major (gaps 0-7 q0, 8-15 q1), then PV (gaps 16-19, 20-23, 24-27, 28-31).
// Row max of the NEW scores: q0 in gaps 9-16 (decision in gap 17), q1 in gaps 17-24 (decision in gap 25).
// LDS reads, one per gap: the 16 transposed V reads in gaps 0-15, the eight K fragments of the NEXT half-step
// in gaps 16-23.
// State at entry (and at exit, for sN): elements 0..7 exponentiated, 0..6 summed, pairs 0..2 packed, element 8
// scaled.
constexpr int kLead = 8;
constexpr float kNoKey = -1.0e30f;

__device__ __forceinline__ constexpr int el_q(int i) { return (i >> 3) & 1; }
__device__ __forceinline__ constexpr int el_r(int i) { return 8 * (i >> 4) + (i & 7); }

template <class Tr, int ORD, int I>
__device__ __forceinline__ void st_f(f32x16 (&s)[2], const float (&msc)[2], float c2) {
    if constexpr (ORD != 6 && I >= 0 && I < 32)
        asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(s[el_q(I)][el_r(I)]) : "s"(c2), "v"(msc[el_q(I)]));
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
// moves, move msc / lsum now and park the factor for O (applied at the entry of the next half-step).
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
    if (__any(__builtin_fmaf(mxl, c2, -kThr) > acc.msc[q])) {   // rare after the first tiles
        const float mx = half_max(mxl) * c2;                    // both lane halves hold the same query
        const float mnew = fmaxf(acc.msc[q], mx);
        const float al = fast_exp2(acc.msc[q] - mnew);          // (both finite: 1 if the row still has no key, 0 at its first)
        acc.msc[q] = mnew;
        acc.lsum[q] *= al;
        acc.alpha[q] = al;
        pend = 1;
    }
}
// The same point of the gap program when the new scores belong to the NEXT q-tile: the finished q-tile's row sum
// and reference leave for `fin` (every add under them is over: q0's last in gap 16, q1's in gap 24), and the new
// reference is set outright from the first half-tile's maximum.
template <class Tr, int D, int ORD>
__device__ __forceinline__ void fresh_start(Acc<D> &acc, Fin &fin, int q, f32x16 &s, float mxl, float c2) {
    fin.lsum[q] = acc.lsum[q];
    fin.msc[q] = acc.msc[q];
    const float mx = half_max(mxl);
    acc.lsum[q] = 0.f;
    if (ORD == 6) {
        const float m0 = (mx == ninf()) ? 0.f : mx;
        acc.msc[q] = m0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] -= m0; acc.cinit[q][r] = -m0; }
    } else {
        acc.msc[q] = fmaxf(mx * c2, kNoKey);
    }
}

// Apply the rescales of O decided during the previous half-step (wave-uniform, rare).
template <int D>
__device__ __forceinline__ void apply_pending(Acc<D> &acc, int &pend) {
    if (pend) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { rescale_o<D>(q, acc.alpha[q]); acc.alpha[q] = 1.0f; }
        pend = 0;
    }
}

// This lane's index, recomputed where it is needed: a per-lane constant computed at kernel entry lives in a VGPR through
// the half-step loop (which has none to spare -- hipcc spills it and waits vmcnt(0) for the reload, draining the LDS-DMA).
__device__ __forceinline__ int lane_now() {
    int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));             // (not to be merged with any other copy of it)
    return l;
}
// Keys beyond a row's last visible one get a score of -inf -- through the C operand of the block's FIRST QK^T MFMA:
// mask_tuple() builds the additive mask (0 / -inf per score register) in the wave-uniform, rare branch (diagonal and
// ragged tiles only) and the scores leave the matrix pipe masked.  Selecting on the finished scores instead makes them
// a new value on one path only, and hipcc reconciles the two paths with v_mov copies on the path that does NOT mask --
// every half-step (SQ_INSTS_VALU +15 %); left to hipcc the sixteen selects were ~80 instructions besides.
// Row of this lane in the block: qrow0 + (lane & 31) is its last visible key under the causal mask (qrow0 = the block's
// first row + Sk - Sq; huge without a mask), klast the last key there is, kbase the first of the 32 keys.
__device__ __forceinline__ f32x16 mask_tuple(int kbase, int qrow0, int klast) {
    const int l = lane_now();
    const int room = min(klast, qrow0 + (l & 31)) - kbase - 4 * (l >> 5);       // keys with offset <= room stay
    const float ni = ninf();
    float c[16];
    // register r holds key offset (r & 3) + 8 (r >> 2): 0 1 2 3 8 9 10 11 16 17 18 19 24 25 26 27
    asm volatile(
        "v_cmp_gt_i32_e32 vcc, 0, %16\n\tv_cndmask_b32_e32 %0, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 1, %16\n\tv_cndmask_b32_e32 %1, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 2, %16\n\tv_cndmask_b32_e32 %2, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 3, %16\n\tv_cndmask_b32_e32 %3, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 8, %16\n\tv_cndmask_b32_e32 %4, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 9, %16\n\tv_cndmask_b32_e32 %5, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 10, %16\n\tv_cndmask_b32_e32 %6, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 11, %16\n\tv_cndmask_b32_e32 %7, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 16, %16\n\tv_cndmask_b32_e32 %8, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 17, %16\n\tv_cndmask_b32_e32 %9, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 18, %16\n\tv_cndmask_b32_e32 %10, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 19, %16\n\tv_cndmask_b32_e32 %11, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 24, %16\n\tv_cndmask_b32_e32 %12, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 25, %16\n\tv_cndmask_b32_e32 %13, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 26, %16\n\tv_cndmask_b32_e32 %14, 0, %17, vcc\n\t"
        "v_cmp_gt_i32_e32 vcc, 27, %16\n\tv_cndmask_b32_e32 %15, 0, %17, vcc"
        : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6]), "=&v"(c[7]), "=&v"(c[8]),
          "=&v"(c[9]), "=&v"(c[10]), "=&v"(c[11]), "=&v"(c[12]), "=&v"(c[13]), "=&v"(c[14]), "=&v"(c[15])
        : "v"(room), "v"(ni)
        : "vcc");
    f32x16 cm;
#pragma unroll
    for (int r = 0; r < 16; ++r) cm[r] = c[r];
    return cm;
}

// One pipelined half-step of 32 gaps:
//   NEW 1: sN <- scores of the 32 keys whose K fragments are in kpre, both query blocks        (gaps 0-15)
//   NEW 2: the same, but those keys are the first half-tile of the NEXT q-tile (a[128:191] already hold its Q rows): the
//          decision stages start a fresh reference and move the finished q-tile's sums to fin
//   NEW 0: no new scores (this wave has passed its causal diagonal)
//   OLD 1: sO = scores of keys [32*HO, +32) of the tile whose V is at vbuf, in the entry state: finished,
//          O^T += V^T . P^T (gaps 16-31); FIRST: the k-step-0 products start O from zero (first half-tile of a q-tile)
//   OLD 0: nothing to consume (a wave joining the next q-tile after idling behind its diagonal)
//   and sN is left in the entry state for the next half-step.
// mask_n bit q: sN[q] holds keys that must be masked (diagonal / ragged tiles); kbase_n = their first key;
// qbase_n / klast: see mask_tuple (of the q-tile sN belongs to); diag0 = mask_tuple(0, 0, huge): the mask of a 32 x 32
// block on the causal diagonal; ninf16: sixteen times -inf.
// kpre in: the K fragments of this half-step; out: those of the next one (rows [32*PH, +32) of the tile at
// kbuf_pref).  hook(n): extra work for gap n (the LDS-DMA pieces of H2).
template <class Tr, int D, int ORD, int HO, int PH, int NEW, int OLD, int MK, class Hook = NoHook>
__device__ __forceinline__ void hstep(const lds_char *lds, unsigned k_e, unsigned v_e, int vbuf, int kbuf_pref,
                                      f32x16 (&sN)[2], f32x16 (&sO)[2],
                                      Acc<D> &acc, Fin &fin, int &pend, float c2, int mask_n, int kbase_n, int qbase_n,
                                      int klast, const f32x16 &diag0, const f32x16 &ninf16, typename Tr::mfma_vec (&kpre)[D / 16],
                                      const Hook &hook = Hook()) {
    using Vec = typename Tr::mfma_vec;
    constexpr int NKS = D / 16, NDB = D / 32;
    constexpr int RG = Img<D>::RG;
    constexpr bool PS = (ORD == 6);
    static_assert(NKS == 8 && NDB == 4, "the gap program below is written for head_dim 128");
    static_assert(NEW || OLD, "an idle half-step issues nothing");

    const lds_char *const vb_0 = lds + (v_e + vbuf), *const vb_1 = lds + ((v_e ^ 32) + vbuf);
    const lds_char *const kp_e = lds + (k_e + kbuf_pref), *const kp_o = lds + ((k_e ^ 32) + kbuf_pref);
    auto ld_kp = [&](int ks) __attribute__((always_inline)) -> Vec {
        return bitcast<Vec>(lds_read16(((ks & 1) ? kp_o : kp_e) + 4 * RG * PH + 512 * (ks >> 1)));
    };
    // transposed read e (0 / 1) of V fragment j (A operand of the PV MFMAs of d block j % 4, k-step j / 4)
    auto ld_vt = [&](int j, int e) __attribute__((always_inline)) -> u32x2 {
        const int d = j % NDB, s = 2 * HO + j / NDB;
        return bitcast<u32x2>(__builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_i16x4 *)((e ? vb_1 : vb_0) + RG * (2 * s + e) + 512 * d)));
    };

    if constexpr (OLD != 0) apply_pending<D>(acc, pend);

    Vec kf[NKS];
    u32x2 vlo[2 * NDB], vhi[2 * NDB];
    if constexpr (NEW != 0) {
#pragma unroll
        for (int i = 0; i < NKS; ++i) kf[i] = kpre[i];
        asm volatile("" :: "v"(kf[NKS - 1]));   // one wait for all eight K fragments (read >= 8 gaps ago)
    }
    float m0, m1;                           // lane max of the new scores, q0 / q1 (first written in gaps 9 / 17)
    static_for<32>([&](auto ic) {
        constexpr int n = decltype(ic)::value;
        // ---- the MFMA of this gap ----
        if constexpr (n < 16) {
            if constexpr (NEW != 0) {
                constexpr int q = n >> 3, ks = n & 7;
                if constexpr (ks == 0) {
                    // MK 0: the caller knows that no key of this half-tile needs masking -- the half-step loop of a q-tile's
                    //       inner tiles;
                    // MK 2 / 3: the caller knows WHICH mask each block takes -- the two half-steps in which a wave crosses
                    //       an aligned causal diagonal (rows and keys of its 64 x 64 corner start together): 2 = query block 0
                    //       on the diagonal, block 1 unmasked; 3 = block 0 fully masked, block 1 on the diagonal.  The mask
                    //       tuple is the MFMA's C operand directly: no branch, no copy (exact flavour only);
                    // MK 1: anything else (unaligned Sk - Sq, ragged tiles, the first scores of a q-tile), a few half-steps
                    //       per q-tile: the C operand is picked at run time.
                    constexpr bool MASK = MK == 1;
                    if constexpr (MK == 2 || MK == 3) {
                        static_assert(!(PS && NEW == 1) || MK < 2, "the static masks have no room for the prescaled reference");
                        if constexpr (MK == 2 && q == 1) own_qk_first<Tr, q_reg(q, 0)>(sN[q], kf[0]);
                        else own_qk_first_c<Tr, q_reg(q, 0)>(sN[q], kf[0], (MK == 3 && q == 0) ? ninf16 : diag0);
                    } else
                    if constexpr (MASK) {
                        // ONE MFMA statement whose C tuple is picked by copies (16 v_mov a path): four statements behind a
                        // four-way branch would do without them, but these half-steps run three times per wave and q-tile
                        const int qrow0 = qbase_n + 32 * q;
                        f32x16 cm;
                        if (!(mask_n & (1 << q))) {
                            if constexpr (PS && NEW == 1) {
                                cm = acc.cinit[q];
                            } else {
#pragma unroll
                                for (int r = 0; r < 16; ++r) cm[r] = 0.f;
                            }
                        } else if (kbase_n > min(klast, qrow0 + 31)) {          // no row of the block sees any of these keys
                            cm = ninf16;
                        } else if (!(PS && NEW == 1) && kbase_n == qrow0 && kbase_n + 31 <= klast) {
                            cm = diag0;                                         // the aligned causal diagonal
                        } else {
                            cm = mask_tuple(kbase_n, qrow0, klast);
                            if constexpr (PS && NEW == 1) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) cm[r] += acc.cinit[q][r];
                            }
                        }
                        own_qk_first_c<Tr, q_reg(q, 0)>(sN[q], kf[0], cm);
                    } else if constexpr (PS && NEW == 1) {
                        own_qk_first_c<Tr, q_reg(q, 0)>(sN[q], kf[0], acc.cinit[q]);
                    } else {
                        own_qk_first<Tr, q_reg(q, 0)>(sN[q], kf[0]);
                    }
                } else {
                    own_qk<Tr, q_reg(q, ks)>(sN[q], kf[ks]);
                }
            }
        } else if constexpr (OLD != 0) {
            constexpr int blk = (n - 16) >> 2, d = (n - 16) & 3, q = blk & 1, ks = blk >> 1, j = NDB * ks + d;
            u32x4 av, pv;
            av[0] = vlo[j][0]; av[1] = vlo[j][1]; av[2] = vhi[j][0]; av[3] = vhi[j][1];
            pv[0] = acc.pk[q][4 * ks + 0]; pv[1] = acc.pk[q][4 * ks + 1];
            pv[2] = acc.pk[q][4 * ks + 2]; pv[3] = acc.pk[q][4 * ks + 3];
            own_pv<Tr, o_reg(q, d)>(bitcast<Vec>(av), bitcast<Vec>(pv));
        }
        // ---- one LDS read ----
        if constexpr (n < 16) {
            if constexpr (OLD != 0) {
                if constexpr (n & 1) vhi[n >> 1] = ld_vt(n >> 1, 1);
                else vlo[n >> 1] = ld_vt(n >> 1, 0);
            }
        } else if constexpr (n < 16 + NKS) kpre[n - 16] = ld_kp(n - 16);
        // One s_waitcnt per batch of fragments instead of one per MFMA: naming the YOUNGEST read of a batch makes
        // hipcc wait for the whole batch here, and every batch was issued at least eight gaps ago.
        if constexpr (OLD != 0) {
            if constexpr (n == 15) asm volatile("" :: "v"(vhi[NDB - 1]));
            if constexpr (n == 23) asm volatile("" :: "v"(vhi[2 * NDB - 1]));
        }
        // ---- softmax stages of the half-tile being consumed ----
        if constexpr (OLD != 0) {
            if constexpr (n == 0) st_a<Tr, kLead - 1>(sO, acc.lsum, acc.pk);
            st_f<Tr, ORD, n + kLead + 1>(sO, acc.msc, c2);
            st_x<n + kLead>(sO);
            if constexpr (n >= 1) st_a<Tr, n + kLead - 1>(sO, acc.lsum, acc.pk);
        }
        // ---- row max of the new scores, and their lead stages ----
        if constexpr (NEW != 0) {
            // without a P.V side no MFMA separates q1's last QK^T MFMA (gap 15) from the first read of its result
            if constexpr (OLD == 0 && n == 16) settle(sN[1]);
            if constexpr (n == 9) st_max2(m0, sN[0][0], sN[0][1]);
            if constexpr (n > 9 && n <= 16) st_max3(m0, sN[0][2 * (n - 9)], sN[0][2 * (n - 9) + 1]);
            if constexpr (n == 17) {
                if constexpr (NEW == 2) fresh_start<Tr, D, ORD>(acc, fin, 0, sN[0], m0, c2);
                else decide<Tr, D, ORD>(acc, 0, sN[0], m0, c2, pend);
            }
            if constexpr (n == 17) st_max2(m1, sN[1][0], sN[1][1]);
            if constexpr (n > 17 && n <= 24) st_max3(m1, sN[1][2 * (n - 17)], sN[1][2 * (n - 17) + 1]);
            if constexpr (n == 25) {
                if constexpr (NEW == 2) fresh_start<Tr, D, ORD>(acc, fin, 1, sN[1], m1, c2);
                else decide<Tr, D, ORD>(acc, 1, sN[1], m1, c2, pend);
            }
            if constexpr (n >= 23) st_f<Tr, ORD, n - 23>(sN, acc.msc, c2);         // elements 0..8
            if constexpr (n >= 24) st_x<n - 24>(sN);                                 // elements 0..7
            if constexpr (n >= 25) st_a<Tr, n - 25>(sN, acc.lsum, acc.pk);           // elements 0..6
        }
        hook(n);
        SFA_FENCE();
    });
    if constexpr (NEW == 0) {               // the q-tile ends here for this wave: its sums stay where the epilogue looks
        fin.lsum[0] = acc.lsum[0]; fin.lsum[1] = acc.lsum[1];
        fin.msc[0] = acc.msc[0]; fin.msc[1] = acc.msc[1];
    }
}

}  // namespace w4

// Kernel arguments: what launch_w4_t derives from PrefillKernelParams, so that the kernel divides nothing per q-tile.
// The kernel reads them through a LAUNDERED kernarg pointer wherever a q-tile begins or ends (arg() below): hipcc
// otherwise loads every field at kernel entry and keeps ~60 scalars alive through the half-step loop, spilling the
// loop's own scalars to VGPR lanes (v_readlane / v_writelane in the MFMA gaps).
struct W4CurArgs {       // what a cursor step needs (read in ONE burst where a cursor moves to its next q-tile)
    int adv_hl, adv_i, adv_b, adv_h;    // one cursor step (= nslots units): heads, units, batches, heads mod Hq
    int U, Hq, bh_per_xcd, BH, nq, Sk, coff, G;     // U = units per head; G = query heads per K/V head
};
struct W4DescArgs {      // what the K / V buffer descriptors of a head need
    const uint16_t *k, *v;
    long long ks0, ks1, vs0, vs1;       // element strides: batch, head
    unsigned k_rowb, v_rowb;            // bytes between rows
    int k_extent, v_extent;             // bytes of one head's K / V rows
};
struct W4Args {
    W4CurArgs cur;
    W4DescArgs kv;
    const uint16_t *q;
    uint16_t *o;
    float *lse;
    long long qs0, qs1, os0, os1, os2;  // element strides: batch, head (and O's row)
    unsigned q_rowb;
    int Sq;
    float c2;                           // softmax scale * log2(e)
};

// Which items (q-tiles) a workgroup walks, in which order.  blockIdx & 7 labels the XCD (round-robin
// dispatch; a speed hint only), which owns heads [xcd * bh_per_xcd, +bh_per_xcd); its work list is
// head-major, U units per head -- causal: unit i = the q-tile pair (nq-1-i, i); full: unit i = q-tile i --
// and the XCD's workgroup `slot` takes units slot, slot + nslots, ...  All scalar, advanced by additions.
struct W4Cursor {
    int hl, i;          // head index inside the XCD's range, unit inside the head
    int sub;            // causal: 0 = the heavy q-tile of the pair, 1 = the light one
    int t, nt;          // tile inside the item (producer only), tiles of the item
    int b, h, qt;       // batch, head, q-tile
    int live;           // (int: a struct copy with padding bytes goes through scratch)
};

// DIAG (A/B library only): 256 = workgroup 8 stamps s_memtime at five points of each of its first 16 q-tiles -- and
// s_memrealtime around them, for the clock -- into the caller's lse buffer, which then carries no lse.
template <class Tr, int D, bool CAUSAL, int ORD, int DIAG = 0>
__global__ void __launch_bounds__(w4::kThreadsW4, 1)
prefill_w4_kernel(const W4Args args_by_value) {
    using namespace w4;
    using Vec = typename Tr::mfma_vec;
    constexpr int NQB = 2;
    constexpr bool PS = (ORD == 6);
    constexpr int NKS = D / 16, NDB = D / 32;
    constexpr int NJ = D / 64;                  // 128-byte column pieces per row
    using L = Img<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h2 = lane >> 5;
    (void)args_by_value;
    // the kernel arguments, re-read from the kernarg segment wherever they are needed (see W4Args)
    typedef const W4Args __attribute__((address_space(4))) *ArgPtr;
    auto arg = []() __attribute__((always_inline)) -> ArgPtr {
        ArgPtr a = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(a));
        return a;
    };
    // (field by field: a struct copy out of the constant address space does not compile on the host pass; hipcc merges the
    // scalar loads into two or three wide ones all the same)
    auto load_cur = [](ArgPtr a) __attribute__((always_inline)) -> W4CurArgs {
        W4CurArgs c;
        c.adv_hl = a->cur.adv_hl; c.adv_i = a->cur.adv_i; c.adv_b = a->cur.adv_b; c.adv_h = a->cur.adv_h;
        c.U = a->cur.U; c.Hq = a->cur.Hq; c.bh_per_xcd = a->cur.bh_per_xcd; c.BH = a->cur.BH; c.nq = a->cur.nq;
        c.Sk = a->cur.Sk; c.coff = a->cur.coff; c.G = a->cur.G;
        return c;
    };
    auto load_kv = [](ArgPtr a) __attribute__((always_inline)) -> W4DescArgs {
        W4DescArgs d;
        d.k = a->kv.k; d.v = a->kv.v; d.ks0 = a->kv.ks0; d.ks1 = a->kv.ks1; d.vs0 = a->kv.vs0; d.vs1 = a->kv.vs1;
        d.k_rowb = a->kv.k_rowb; d.v_rowb = a->kv.v_rowb; d.k_extent = a->kv.k_extent; d.v_extent = a->kv.v_extent;
        return d;
    };
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;

    auto item_tiles = [&](const W4CurArgs &a, int qt) __attribute__((always_inline)) -> int {
        int kv_end = a.Sk;
        if (CAUSAL) kv_end = min(a.Sk, qt * kRows + kRows + a.coff);
        return kv_end > 0 ? (kv_end + kKeys - 1) / kKeys : 0;
    };
    auto step_unit = [&](const W4CurArgs &a, W4Cursor &c) __attribute__((always_inline)) {
        c.i += a.adv_i;
        int carry = 0;
        if (c.i >= a.U) { c.i -= a.U; carry = 1; }
        c.hl += a.adv_hl + carry;
        c.b += a.adv_b;
        c.h += a.adv_h + carry;
        if (c.h >= a.Hq) { c.h -= a.Hq; ++c.b; }
    };
    // position the cursor on the first existing item at or after (hl, i, sub); skip_empty: also skip items
    // without any tile (causal rows that see no key)
    auto seek = [&](const W4CurArgs &a, W4Cursor &c, bool skip_empty) __attribute__((always_inline)) {
        for (;;) {
            if (c.hl >= a.bh_per_xcd || xcd * a.bh_per_xcd + c.hl >= a.BH) { c.live = 0; return; }
            const int heavy = CAUSAL ? a.nq - 1 - c.i : c.i;
            if (c.sub == 0 || (CAUSAL && heavy != c.i)) {
                c.qt = c.sub == 0 ? heavy : c.i;
                c.nt = item_tiles(a, c.qt);
                c.t = 0;
                if (!skip_empty || c.nt > 0) { c.live = 1; return; }
            }
            if (CAUSAL && c.sub == 0) { c.sub = 1; }
            else { c.sub = 0; step_unit(a, c); }
        }
    };
    auto next_item_a = [&](const W4CurArgs &a, W4Cursor &c, bool skip_empty) __attribute__((always_inline)) {
        if (CAUSAL && c.sub == 0) { c.sub = 1; }
        else { c.sub = 0; step_unit(a, c); }
        seek(a, c, skip_empty);
    };
    auto next_item = [&](W4Cursor &c, bool skip_empty) __attribute__((always_inline)) {
        const W4CurArgs ca = load_cur(arg());
        next_item_a(ca, c, skip_empty);
    };
    auto first_item = [&](W4Cursor &c, bool skip_empty) __attribute__((always_inline)) {
        const W4CurArgs a = load_cur(arg());
        c.hl = slot / a.U; c.i = slot - c.hl * a.U; c.sub = 0; c.t = 0; c.nt = 0; c.qt = 0; c.live = 0;
        const int bh = xcd * a.bh_per_xcd + c.hl;
        c.b = bh / a.Hq; c.h = bh - c.b * a.Hq;
        seek(a, c, skip_empty);
    };

    // ---- LDS-DMA producers ----
    // Wave w stages rows [16w, 16w+16) of every tile: row groups 2w (half 0) and 2w+1 (half 1), NJ pieces of
    // 8 rows x 128 B each.  Lane -> (sub-tile lane>>5, row (lane>>2)&7, slot lane&3) of its piece; the source
    // chunk is slot ^ ((row>>2)&3) so that LDS, written linearly, holds the swizzled image.
    const int r8 = (lane >> 2) & 7, dslot = lane & 3, dsub = lane >> 5;
    const unsigned k_rowb = arg()->kv.k_rowb, v_rowb = arg()->kv.v_rowb;
    const unsigned kvoff0 = (unsigned)r8 * k_rowb + 64u * dsub + 16u * (dslot ^ (r8 >> 2));
    const unsigned kvoff1 = (unsigned)(r8 + 8) * k_rowb + 64u * dsub + 16u * (dslot ^ (2 + (r8 >> 2)));
    const unsigned vvoff0 = (unsigned)r8 * v_rowb + 64u * dsub + 16u * (dslot ^ (r8 >> 2));
    const unsigned vvoff1 = (unsigned)(r8 + 8) * v_rowb + 64u * dsub + 16u * (dslot ^ (2 + (r8 >> 2)));
    const lds_char *const lds = (const lds_char *)smem;
    const unsigned lds0 = (unsigned)(uintptr_t)lds;         // LDS byte address of the dynamic segment
    // One head's K (or V) rows form a buffer of `extent` bytes (prefill_w4_serves keeps it below 2 GiB); the
    // descriptor a wave uses for a tile starts at ITS 16 rows of that tile and ends with the head, so rows
    // past the sequence end read as zeros.  Per tile the descriptor only moves by one tile's bytes.
    struct Desc { unsigned lo, hi; int left; };
    const int k_tileb = kKeys * (int)k_rowb, v_tileb = kKeys * (int)v_rowb;
    auto desc_at_head = [&](const W4DescArgs &a, int G, bool is_k, int b, int h) __attribute__((always_inline)) -> Desc {
        const int hk = G == 1 ? h : h / G;
        const uint16_t *head = is_k ? a.k + b * a.ks0 + hk * a.ks1 : a.v + b * a.vs0 + hk * a.vs1;
        const unsigned skip = 16u * wave * (is_k ? a.k_rowb : a.v_rowb);
        const unsigned long long base = (unsigned long long)(uintptr_t)head + skip;
        return Desc{(unsigned)base, (unsigned)(base >> 32), (is_k ? a.k_extent : a.v_extent) - (int)skip};
    };
    auto desc_advance = [&](Desc &d, int tileb) __attribute__((always_inline)) {
        const unsigned lo = d.lo + (unsigned)tileb;
        d.hi += lo < d.lo ? 1u : 0u;
        d.lo = lo;
        d.left -= tileb;
    };
    // the buffer descriptor of a tile; `live` false (the producer has run out of tiles): zero bytes, so the
    // pieces still issue -- no branch in the MFMA gaps -- and simply zero-fill a ring slot nobody will read
    auto make_srd = [&](const Desc &d, bool live) __attribute__((always_inline)) -> u32x4s {
        u32x4s srd;
        srd[0] = d.lo;
        srd[1] = d.hi & 0xffffu;
        srd[2] = live ? (unsigned)max(d.left, 0) : 0u;
        srd[3] = 0x00020000u;
        return srd;
    };
    // piece idx (0 .. 2*NJ-1) of this wave's share of a tile: row group idx / NJ, column piece idx % NJ
    auto issue_piece = [&](const u32x4s &srd, bool is_k, int ring_off, int idx) __attribute__((always_inline)) {
        const int half = idx / NJ, j = idx % NJ;
        const unsigned dst = lds0 + (is_k ? L::K_BASE : L::V_BASE) + ring_off + 2 * wave * L::RG;
        dma_piece(dst + half * L::RG + 1024 * j, is_k ? (half ? kvoff1 : kvoff0) : (half ? vvoff1 : vvoff0), srd, 128u * j);
    };

    // ---- this lane's LDS read bases (the odd twins are ^32) ----
    const int kx = (l31 >> 2) & 3;
    const unsigned k_e = L::K_BASE + L::RG * (l31 >> 3) + 64 * (l31 & 7) + 16 * (h2 ^ kx);
    const int vy = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    const unsigned v_e = L::V_BASE + 64 * (4 * h2 + ((lane & 15) >> 2)) + 16 * (vy ^ h2) + 8 * (lane & 1);

    const float c2 = arg()->c2;

    // ---- cursors: the compute, and the DMA producer running ahead of it.  The K producer is three stream
    // positions ahead of the compute, the V producer two: V re-issues the tile K issued one call earlier,
    // from the descriptor K's call left behind for it (vpend). ----
    W4Cursor pc;
    first_item(pc, true);
    Desc kd = {0, 0, 0}, vd = {0, 0, 0}, vpend = {0, 0, 0};
    int vpend_live = 0;
    if (pc.live) {
        const ArgPtr a = arg();
        const W4DescArgs da = load_kv(a);
        const int G = a->cur.G;
        kd = desc_at_head(da, G, true, pc.b, pc.h);
        vd = desc_at_head(da, G, false, pc.b, pc.h);
    }
    int kring_p = 0, vring_p = 0;               // ring byte offsets the producers write next
    auto ring_next = [](int x) __attribute__((always_inline)) -> int { return x == (kRing - 1) * L::TILE ? 0 : x + L::TILE; };
    // one piece per call (spread over the MFMA gaps of H2): V pieces first, then K pieces
    u32x4s piece_srd = {0, 0, 0, 0};            // descriptor of the tile whose pieces are being dealt out
    auto produce_v_piece = [&](int idx) __attribute__((always_inline)) {
        if (idx == 0) piece_srd = make_srd(vpend, vpend_live != 0);
        issue_piece(piece_srd, false, vring_p, idx);
        if (idx == 2 * NJ - 1) vring_p = kring_p;
    };
    auto k_advance = [&]() __attribute__((always_inline)) {
        vpend = vd;
        if (++pc.t < pc.nt) {
            desc_advance(kd, k_tileb);
            desc_advance(vd, v_tileb);
        } else {
            // the producer moves to the next q-tile: ONE burst of scalar loads for everything the step and the two
            // descriptors need, one s_waitcnt (this runs in an MFMA gap of all four waves at once: tools/w4_events.py)
            const ArgPtr a = arg();
            const W4CurArgs ca = load_cur(a);
            const W4DescArgs da = load_kv(a);
            next_item_a(ca, pc, true);
            if (pc.live) { kd = desc_at_head(da, ca.G, true, pc.b, pc.h); vd = desc_at_head(da, ca.G, false, pc.b, pc.h); }
        }
    };
    auto produce_k_piece = [&](int idx) __attribute__((always_inline)) {
        if (idx == 0) piece_srd = make_srd(kd, pc.live != 0);
        issue_piece(piece_srd, true, kring_p, idx);
        if (idx == 2 * NJ - 1) {
            vpend_live = pc.live;
            if (pc.live) k_advance();
            kring_p = ring_next(kring_p);
        }
    };
    // the same tile by tile, outside a half-step (prologue, idle steps)
    auto produce_v = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int idx = 0; idx < 2 * NJ; ++idx) produce_v_piece(idx);
    };
    auto produce_k = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int idx = 0; idx < 2 * NJ; ++idx) produce_k_piece(idx);
    };
    auto dma_hook = [&](int n) __attribute__((always_inline)) {                            // one piece per gap: gaps 0..7 of every H2
        if (n < 2 * NJ) produce_v_piece(n);
        else if (n < 4 * NJ) produce_k_piece(n - 2 * NJ);
    };

    // Vector-memory operations this wave has issued SINCE the last K/V piece and that are certain to have been
    // issued (the Q request: 16 pieces; the O stores of a wave whose 64 rows all exist: 16).  vmcnt retires in
    // order, so the barrier of a step -- which needs the pieces issued behind the previous barrier -- may leave
    // that many operations in flight.
    int young = 0;
    auto wait_and_sync = [&]() __attribute__((always_inline)) {
        if (young < 8) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");         // (the usual case first: one branch)
        else if (young < 16) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else if (young < 24) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
        else if (young < 32) asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(32)\n\ts_barrier" ::: "memory");
        young = 0;                                          // (the K/V pieces of the step follow)
    };

    // stream prologue: K(0), V(0), K(1); the half-step that scores the first q-tile's first half-tile stands in for
    // "H2 of step -1" and issues V(1), K(2) like every other H2.  The compute's ring position starts one slot back.
    produce_k(); produce_v(); produce_k();
    int kcur = (kRing - 1) * L::TILE, vcur = (kRing - 1) * L::TILE;     // ring byte offsets of the compute's current tile

    // Q rows reach the accumulator file through LDS: 16 LDS-DMA pieces in the K image into a wave-private 64-row
    // image, read back like K fragments.  Rows past Sq read as zeros.
    auto load_q = [&](int b, int h, int qt) __attribute__((always_inline)) {
        const ArgPtr a = arg();
        const int row0 = qt * kRows + 64 * wave;
        const unsigned rowb = a->q_rowb;
        const int ln = lane_now(), qr8 = (ln >> 2) & 7, qslot = ln & 3, qsub = ln >> 5;
        const unsigned qvoff0 = (unsigned)qr8 * rowb + 64u * qsub + 16u * (qslot ^ (qr8 >> 2));
        const unsigned qvoff1 = (unsigned)(qr8 + 8) * rowb + 64u * qsub + 16u * (qslot ^ (2 + (qr8 >> 2)));
        const unsigned long long base = (unsigned long long)(uintptr_t)(a->q + b * a->qs0 + h * a->qs1) + (unsigned long long)row0 * rowb;
        u32x4s srd;
        srd[0] = (unsigned)base;
        srd[1] = (unsigned)(base >> 32) & 0xffffu;
        srd[2] = row0 < a->Sq ? (unsigned)(a->Sq - 1 - row0) * rowb + 2u * D : 0u;
        srd[3] = 0x00020000u;
        const unsigned dst = lds0 + L::Q_BASE + wave * L::TILE;
#pragma unroll
        for (int rg = 0; rg < 8; ++rg)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                dma_piece(dst + rg * L::RG + 1024 * j, (rg & 1) ? qvoff1 : qvoff0, srd, 128u * j + 16u * (rg >> 1) * rowb);
        young += 8 * NJ;
    };
    // The Q rows go from the wave's image into a[128:191] in two parts: sixteen reads, and the wait for them in front of
    // the first MFMA that names the new rows.  A wave issues the reads as early as it can -- its last QK^T MFMA on the old
    // rows is behind it and the image has landed: in the gaps of its last half-step that still scores (qf_hook), in an idle
    // step behind its causal diagonal, or, failing both (q-tiles of one or two tiles, the first of the list), at the top
    // of the next q-tile, where every wave would otherwise stand for the LDS latency of its whole image at once.
    auto q_addr = [&](unsigned &qe, unsigned &qo) __attribute__((always_inline)) {
        const int ln = lane_now(), ql31 = ln & 31, qh2 = ln >> 5;
        const unsigned q_e = L::Q_BASE + L::RG * (ql31 >> 3) + 64 * (ql31 & 7) + 16 * (qh2 ^ ((ql31 >> 2) & 3));
        qe = lds0 + q_e + wave * L::TILE;
        qo = lds0 + (q_e ^ 32) + wave * L::TILE;
    };
    auto fetch_q_piece = [&](auto ic, unsigned qe, unsigned qo) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, q = i / NKS, ks = i % NKS;
        const unsigned addr = (ks & 1) ? qo : qe;
        asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" :: "v"(addr), "n"(q_reg(q, ks)), "n"(q_reg(q, ks) + 3),
                     "n"(4 * L::RG * q + 512 * (ks >> 1)) : SFA_AOWN, "memory");
    };
    auto fetch_q_issue = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the pieces (and whatever this wave issued since)
        unsigned qe, qo;
        q_addr(qe, qo);
        static_for<NQB * NKS>([&](auto ic) { fetch_q_piece(ic, qe, qo); });
    };
    auto fetch_q_done = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: SFA_AOWN, "memory");
        if (PS) {           // prescaled flavour: fold scale * log2(e) into Q once per q-tile
            static_for<4 * NQB * NKS>([&](auto ic) {
                constexpr int r = 128 + decltype(ic)::value;
                const uint32_t w = bitcast<uint32_t>(own_read<r>());
                const uint32_t w2 = Tr::pack2(Tr::lo_f32(w) * c2, Tr::hi_f32(w) * c2);
                asm volatile("v_accvgpr_write_b32 a%c1, %0" :: "v"(w2), "n"(r) : SFA_AOWN);
            });
            asm volatile("s_nop 1" ::: SFA_AOWN);   // v_accvgpr_write -> MFMA operand: two wait states
        }
    };
    // In the gaps: gap 15 waits for the image -- the request (16 pieces) is followed by exactly the eight K/V pieces of one
    // H2 wherever qf_on is set (see the call sites), and vmcnt retires in order -- gaps 16..31 issue one read each.  The
    // last MFMA that read the old rows issued in gap 15 at the latest; a read's data arrives an LDS latency later.
    bool qf_on = false;                 // wave-uniform: this half-step fetches
    bool q_in = false;                  // the next q-tile's rows are in (or on their way into) a[128:191]
    unsigned qf_e = 0, qf_o = 0;
    auto qf_hook = [&](int n) __attribute__((always_inline)) {
        if (n == 15) {
            q_addr(qf_e, qf_o);
            if (qf_on) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        static_for<NQB * NKS>([&](auto ic) {
            if (n == 16 + decltype(ic)::value) {
                if (qf_on) fetch_q_piece(ic, qf_e, qf_o);
            }
        });
    };
    auto dma_qf_hook = [&](int n) __attribute__((always_inline)) { dma_hook(n); qf_hook(n); };

    // ---- what a wave needs to know about its rows of a q-tile ----
    struct ItemW {
        int wq0;            // this wave's first query row
        int qbase;          // mask_tuple's qrow0 of query block 0: wq0 + Sk - Sq under the causal mask
        int klast;          // Sk - 1
        int ntw;            // tiles this wave computes on (wave-uniform; the others it idles through)
        int whole[NQB];     // key index up to which a 32-key half-tile is visible in full to query block q
        int t_mask;         // first step whose half-steps score keys that may need masking (those of tiles t_mask .. )
        int aligned;        // the wave's last tile is a whole 64 x 64 corner on the causal diagonal (and not its first tile)
    };
    auto item_w_of = [&](int wv, int coff, int Sk, int qt, int nt, ItemW &w) __attribute__((always_inline)) {
        w.wq0 = qt * kRows + 64 * wv;                       // causal: key j visible iff j <= i + coff
        w.qbase = CAUSAL ? w.wq0 + coff : (1 << 29);
        w.klast = Sk - 1;
        w.ntw = nt;
        if (CAUSAL) w.ntw = (w.wq0 + 63 + coff >= 0) ? min(nt, (w.wq0 + 63 + coff) / kKeys + 1) : 0;
#pragma unroll
        for (int q = 0; q < NQB; ++q) w.whole[q] = CAUSAL ? min(w.wq0 + 32 * q + coff - 31, Sk - 32) : Sk - 32;
        // half-tile j (keys 32 j ..) needs masking for a block iff 32 j > whole[q]; H1(t) scores half-tile 2t+1, H2(t) 2t+2
        const int wmin = min(w.whole[0], w.whole[1]);
        const int jm = wmin < 0 ? 0 : wmin / 32 + 1;
        w.t_mask = max(0, (jm - 1) >> 1);
        // rows and keys of the wave's corner start together (wq0 + coff a multiple of 64), the corner is the wave's last
        // tile, lies inside the keys, and H2 of the step before it exists in this q-tile
        const int d0 = w.wq0 + coff;
        w.aligned = CAUSAL && d0 >= 64 && (d0 & 63) == 0 && d0 + 63 <= Sk - 1 && w.ntw == (d0 >> 6) + 1;
    };
    // bit q set: the 32 keys starting at kbase need masking for query block q (wave-uniform)
    auto mask_bits = [&](const ItemW &w, int kbase) __attribute__((always_inline)) -> int {
        int m = 0;
#pragma unroll
        for (int q = 0; q < NQB; ++q)
            if (kbase > w.whole[q]) m |= 1 << q;
        return m;
    };

    // DIAG 256: 0 q-tile top (in front of the Q fetch), 1 behind the barrier, 2 behind the seam half-step and the previous
    // q-tile's epilogue, 3 behind the full steps, 4 in front of the next top; [5] = ntw | nt << 32, [6] = [0]..[4] in
    // s_memrealtime ticks (100 MHz), [7] = this wave's exposed epilogue (cycles, the seam path only)
    unsigned long long its[5] = {0, 0, 0, 0, 0}, rt0 = 0, rt1 = 0, epi = 0;
    int item_no = 0;
    auto istamp = [&](int which) __attribute__((always_inline)) {
        if constexpr ((DIAG & 256) != 0) {
            if (blockIdx.x == 8) {
                unsigned long long tm, rt;
                asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm), "=s"(rt) :: "memory");
                if (which == 0) { its[0] = tm; rt0 = rt; }
                if (which == 1) its[1] = tm;
                if (which == 2) its[2] = tm;
                if (which == 3) its[3] = tm;
                if (which == 4) { its[4] = tm; rt1 = rt; }
                if (which == 5) epi = tm;
                if (which == 6) epi = tm - epi;
            }
        }
    };
    // DIAG 1: workgroup 8 logs (kind, step, time) events of its first four q-tiles, every wave for itself, into the lse
    // buffer as u64[wave][512]: kind 1 H1 done, 2 barrier passed, 3 H2 done, 4 epilogue block done, 5 Q fetched
    int ev_n = 0;
    auto ev = [&](int kind, int t) __attribute__((always_inline)) {
        if constexpr ((DIAG & 1) != 0) {
            if (blockIdx.x == 8 && item_no < 4 && ev_n < 512) {
                unsigned long long tm;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) :: "memory");
                if (lane == 0)
                    reinterpret_cast<unsigned long long *>(arg()->lse)[wave * 512 + ev_n] =
                        (tm & 0xffffffffffffull) | ((unsigned long long)(kind | (t << 4) | (item_no << 12)) << 48);
                ++ev_n;
            }
        }
    };
    auto istore = [&](int ntw, int nt) __attribute__((always_inline)) {
        if constexpr ((DIAG & 1) != 0) ++item_no;
        if constexpr ((DIAG & 256) != 0) {
            if (blockIdx.x == 8 && item_no < 16 && lane == 0) {
                unsigned long long *dst = reinterpret_cast<unsigned long long *>(arg()->lse) + (wave * 16 + item_no) * 8;
                dst[0] = its[0]; dst[1] = its[1]; dst[2] = its[2]; dst[3] = its[3]; dst[4] = its[4];
                dst[5] = (unsigned long long)(unsigned)ntw | ((unsigned long long)(unsigned)nt << 32);
                dst[6] = rt1 - rt0;
                dst[7] = epi;
            }
            ++item_no;
        }
    };

    // the additive mask of a 32 x 32 block on the aligned causal diagonal (rows and keys start together): 16 registers
    // every wave keeps for the whole kernel (it has ~60 to spare), so that the three masked blocks a wave meets per q-tile
    // cost their MFMA's C operand and nothing else
    const f32x16 diag0 = mask_tuple(0, 0, 1 << 29);
    f32x16 ninf16;                              // ... and the mask of a block none of whose keys any row sees
#pragma unroll
    for (int r = 0; r < 16; ++r) ninf16[r] = ninf();
    asm volatile("" : "+v"(ninf16));            // (one tuple kept, not sixteen v_mov wherever it is used)
    {                                           // O starts from zero; every epilogue leaves it so for the next q-tile
        const Vec z = bitcast<Vec>(make_uint4(0u, 0u, 0u, 0u));
        static_for<2 * NDB>([&](auto ic) { own_zero<Tr, 16 * decltype(ic)::value>(z); });
    }
    Acc<D> acc;
    Fin fin = {{0.f, 0.f}, {0.f, 0.f}};
    int pend = 0;                                           // a rescale of O is parked in acc.alpha (wave-uniform)
    f32x16 sA[NQB], sB[NQB];
    Vec kpre[NKS];
#pragma unroll
    for (int q = 0; q < NQB; ++q) acc.alpha[q] = 1.0f;

    // ---- epilogue of one 32-row query block of a wave: normalise, convert, store O[row][:] (and the log-sum-exp) ----
    // ~1.8 k cycles of VALU issue per block (64 accumulator reads, 64 multiplies, 32 converts, 16 half swaps, 8 stores):
    // a wave that finishes a q-tile early stores one block behind its last half-step and the other one step later,
    // so that it is never late at a barrier the working waves are waiting at.
    struct EpiArgs { int Sq; uint16_t *obase; long long os2; float *lse_p; long long lse_row0; };
    auto load_epi = [&](int b, int h) __attribute__((always_inline)) -> EpiArgs {        // one burst of scalar loads per epilogue
        const ArgPtr a = arg();
        const int Sq = a->Sq;
        return EpiArgs{Sq, a->o + b * a->os0 + h * a->os1, a->os2, a->lse, ((long long)b * a->cur.Hq + h) * Sq};
    };
    auto epilogue_q = [&](const EpiArgs &ea, const ItemW &w, int q, bool have_o) __attribute__((always_inline)) {
        const int Sq = ea.Sq;
        const long long os2 = ea.os2;
        float *const lse_p = ea.lse_p;
        const int ln = lane_now(), l31 = ln & 31, h2 = ln >> 5;
        const int qrow = w.wq0 + 32 * q + l31;
        float ltot = 0.f;
        if (have_o) {
            own_settle();
            ltot = half_sum(fin.lsum[q]);
        }
        const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
        f32x2 inv2;
        inv2[0] = inv; inv2[1] = inv;
        if (qrow < Sq) {
            uint16_t *orow = ea.obase + (long long)qrow * os2;
            if (have_o) {
                static_for<2 * NDB>([&](auto ic) {
                    constexpr int d = decltype(ic)::value >> 1, g = 2 * (decltype(ic)::value & 1);
                    // two accumulator registers -> one register pair -> v_pk_mul_f32 (no MFMA runs beside the epilogue, so the
                    // packed op issues at once: tools/micro/pk_f32.hip) -> one v_cvt_pk
                    auto rd2 = [&](auto rc) __attribute__((always_inline)) -> uint32_t {
                        constexpr int r = decltype(rc)::value;
                        f32x2 pr;
                        pr[0] = q == 0 ? own_read<o_reg(0, d) + r>() : own_read<o_reg(1, d) + r>();
                        pr[1] = q == 0 ? own_read<o_reg(0, d) + r + 1>() : own_read<o_reg(1, d) + r + 1>();
                        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pr) : "v"(inv2));
                        return Tr::pack2(pr[0], pr[1]);
                    };
                    using std::integral_constant;
                    const uint32_t ax = rd2(integral_constant<int, 4 * g + 0>{});
                    const uint32_t ay = rd2(integral_constant<int, 4 * g + 2>{});
                    const uint32_t bx = rd2(integral_constant<int, 4 * g + 4>{});
                    const uint32_t by = rd2(integral_constant<int, 4 * g + 6>{});
                    const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                    const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                    // lower lanes: [own g | upper's g] = columns 8g..8g+7; upper lanes: [lower's g+1 | own g+1]
                    *reinterpret_cast<uint4 *>(orow + 32 * d + 8 * g + 8 * h2) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                });
            } else {
#pragma unroll
                for (int d = 0; d < NDB; ++d)
#pragma unroll
                    for (int g = 0; g < 4; g += 2)
                        *reinterpret_cast<uint4 *>(orow + 32 * d + 8 * g + 8 * h2) = make_uint4(0u, 0u, 0u, 0u);
            }
            if (!(DIAG & 257) && lse_p && h2 == 0) {
                const float lse = ltot > 0.f ? (fin.msc[q] + __log2f(ltot)) * kLn2 : ninf();
                lse_p[ea.lse_row0 + qrow] = lse;
            }
        }
        if (have_o) {               // the block's O starts the next q-tile from zero (all lanes: outside the row predicate)
            const Vec z = bitcast<Vec>(make_uint4(0u, 0u, 0u, 0u));
            static_for<NDB>([&](auto ic) {
                constexpr int d = decltype(ic)::value;
                if (q == 0) own_zero<Tr, o_reg(0, d)>(z);
                else own_zero<Tr, o_reg(1, d)>(z);
            });
        }
        if (w.wq0 + 32 * q + 31 < Sq) young += 8;           // every lane stored: 8 row stores at least
    };
    auto epilogue = [&](int b, int h, const ItemW &w, bool have_o) __attribute__((always_inline)) {
        const EpiArgs ea = load_epi(b, h);
        epilogue_q(ea, w, 0, have_o);
        epilogue_q(ea, w, 1, have_o);
    };
    // q-tiles without any key (causal rows in front of the first key; Sq > Sk only): O = 0, lse = -inf.  They own no
    // stream position, so they are dealt with up front; the pipeline below walks the q-tiles that have keys.
    if (CAUSAL && arg()->cur.coff < 0) {
        W4Cursor c0;
        first_item(c0, false);
        while (c0.live) {
            if (c0.nt == 0) {
                ItemW w0;
                const ArgPtr a0 = arg();
                item_w_of(wave, a0->cur.coff, a0->cur.Sk, c0.qt, 0, w0);
                epilogue(c0.b, c0.h, w0, false);
            }
            next_item(c0, false);
        }
    }
    W4Cursor cc;
    first_item(cc, true);
    if (cc.live) {
        load_q(cc.b, cc.h, cc.qt);
        // What the loop carries from one q-tile to the next: the previous q-tile's identity for its epilogue, and
        // whether this wave worked up to its last tile (then the previous q-tile's last half-tile is still to be
        // consumed: sB, O, the sums).
        int pb = 0, ph = 0;
        ItemW pw = {0, 0, 0, 0, {0, 0}, 0, 0};
        bool prev_full = false;
        bool epi_pending = false;                           // query block 1 of q-tile (pb, ph, pw) is still to be stored
        ItemW cw;                                           // this wave's view of the current q-tile
        bool full_wave;
        for (;;) {
            {
                const ArgPtr a = arg();
                item_w_of(wave, a->cur.coff, a->cur.Sk, cc.qt, cc.nt, cw);
            }
            // ======== the second half of the previous q-tile's last step, which scores this q-tile's first half-tile ========
            // (the very first time: nothing to consume, the stream's "step -1")
            // The last QK^T MFMA on the previous q-tile's Q rows is behind us: bring in this q-tile's.
            istamp(0);
            if (!q_in) fetch_q_issue();
            fetch_q_done();
            q_in = false;
            ev(5, 0);
            wait_and_sync();
            ev(2, 63);
            istamp(1);
            {
                const int k1 = ring_next(kcur);
                if (prev_full) {
                    hstep<Tr, D, ORD, 1, 1, 2, 1, 1>(lds, k_e, v_e, vcur, k1, sA, sB, acc, fin, pend, c2,
                                                          mask_bits(cw, 0), 0, cw.qbase, cw.klast, diag0, ninf16, kpre, dma_hook);
                    istamp(5);
                    ev(3, 63);
                    epilogue(pb, ph, pw, true);
                    ev(4, 63);
                    istamp(6);
                } else {                                    // idled behind the previous q-tile's diagonal: join
                    const lds_char *const kb_e = lds + (k_e + k1), *const kb_o = lds + ((k_e ^ 32) + k1);
#pragma unroll
                    for (int i = 0; i < NKS; ++i)
                        kpre[i] = bitcast<Vec>(lds_read16(((i & 1) ? kb_o : kb_e) + 512 * (i >> 1)));
                    hstep<Tr, D, ORD, 1, 1, 2, 0, 1>(lds, k_e, v_e, vcur, k1, sA, sB, acc, fin, pend, c2,
                                                          mask_bits(cw, 0), 0, cw.qbase, cw.klast, diag0, ninf16, kpre, dma_hook);
                    ev(3, 63);
                    if (epi_pending) { epilogue_q(load_epi(pb, ph), pw, 1, true); epi_pending = false; ev(4, 63); }
                }
                kcur = k1;
                vcur = ring_next(vcur);
            }

            istamp(2);
            // ======== q-tile cc: everything but the second half of its last step ========
            const int nt = cc.nt, ntw = cw.ntw;
            const int b = cc.b, h = cc.h;
            W4Cursor nx = cc;
            next_item(nx, true);
            const bool chained = nx.live != 0;
            // the next q-tile's Q rows are requested behind H2 of this step (-1: right here): the pieces then have two
            // steps and a half to land, and the image they overwrite has just been read
            const int qreq_t = chained ? max(nt - 3, -1) : -2;
            if (qreq_t == -1) load_q(nx.b, nx.h, nx.qt);
            const bool qf_ok = qreq_t >= 0;                 // the request is at least two barriers old where the fetch may go
            if (ntw == 0) epilogue(b, h, cw, false);        // rows that see no key at all

            // A wave's steps of a q-tile come in three phases -- full steps, its last tile, idle steps behind its causal
            // diagonal -- written one after the other rather than as one loop with a choice per step: every half-step
            // rewrites ~200 registers of state, and a path that skips one makes hipcc carry copies of them.
            full_wave = !CAUSAL || ntw == nt;               // this wave works up to the q-tile's last tile
            int t = 0;
            auto step_done = [&]() __attribute__((always_inline)) {                        // behind H2(t)
                if (t == qreq_t) load_q(nx.b, nx.h, nx.qt);
                kcur = ring_next(kcur);
                vcur = ring_next(vcur);
                ++t;
            };
            // H1(t): S(B_t) = K(t)[32:64] Q^T  ||  softmax(A_t), O += P(A_t) V(t)[0:32]
            // barrier(t): K(t+2), V(t+1) visible; the slots of K(t), V(t-1) free
            // H2(t): S(A_t+1) = K(t+1)[0:32] Q^T  ||  softmax(B_t), O += P(B_t) V(t)[32:64]  ||  LDS-DMA of K(t+3), V(t+2)
            // ---- phase A: full steps -- the inner tiles: no key of theirs needs masking, and the half-steps carry no
            // mask code ----
            {
                const int t_inner = min(ntw - 1, cw.t_mask);
                while (t < t_inner) {
                    const int k1 = ring_next(kcur);
                    hstep<Tr, D, ORD, 0, 0, 1, 1, 0>(lds, k_e, v_e, vcur, k1, sB, sA, acc, fin, pend, c2,
                                                          0, 0, 0, 0, diag0, ninf16, kpre);
                    ev(1, t);
                    wait_and_sync();
                    ev(2, t);
                    hstep<Tr, D, ORD, 1, 1, 1, 1, 0>(lds, k_e, v_e, vcur, k1, sA, sB, acc, fin, pend, c2,
                                                          0, 0, 0, 0, diag0, ninf16, kpre, dma_hook);
                    ev(3, t);
                    step_done();
                }
            }
            istamp(3);
            // ---- phase B: the tiles at the causal diagonal / the ragged end of the keys (full steps still), and the first
            // half of this wave's last tile; an early wave finishes the q-tile here ----
            if (ntw > 0) {
                if (!PS && cw.aligned) {
                    // the wave crosses an aligned causal diagonal in its last tile: the three masked blocks -- query block 0
                    // against the tile's first half (scored by H2 of the step before), both blocks against its second half
                    // -- take their masks as C operands fixed at compile time.  (t == ntw - 2 here.)
                    {
                        const int k1 = ring_next(kcur);
                        hstep<Tr, D, ORD, 0, 0, 1, 1, 0>(lds, k_e, v_e, vcur, k1, sB, sA, acc, fin, pend, c2,
                                                          0, 0, 0, 0, diag0, ninf16, kpre);
                        ev(1, t);
                        wait_and_sync();
                        ev(2, t);
                        hstep<Tr, D, ORD, 1, 1, 1, 1, PS ? 1 : 2>(lds, k_e, v_e, vcur, k1, sA, sB, acc, fin, pend, c2,
                                                                   1, t * kKeys + 64, cw.qbase, cw.klast, diag0, ninf16, kpre, dma_hook);
                        ev(3, t);
                        step_done();
                    }
                    qf_on = qf_ok && ntw == nt;             // (behind the request: H1(nt-2), H2(nt-2) with its eight pieces)
                    hstep<Tr, D, ORD, 0, 0, 1, 1, PS ? 1 : 3>(lds, k_e, v_e, vcur, ring_next(kcur), sB, sA, acc, fin, pend, c2,
                                                               3, t * kKeys + 32, cw.qbase, cw.klast, diag0, ninf16, kpre, qf_hook);
                    if (qf_on) { q_in = true; qf_on = false; }
                    ev(1, t);
                } else {
                    for (;;) {
                        const int k1 = ring_next(kcur);
                        const int kbase = t * kKeys;
                        qf_on = qf_ok && ntw == nt && t + 1 == nt;
                        hstep<Tr, D, ORD, 0, 0, 1, 1, 1>(lds, k_e, v_e, vcur, k1, sB, sA, acc, fin, pend, c2,
                                                          mask_bits(cw, kbase + 32), kbase + 32, cw.qbase, cw.klast, diag0, ninf16, kpre, qf_hook);
                        if (qf_on) { q_in = true; qf_on = false; }
                        ev(1, t);
                        if (t + 1 == ntw) break;            // the wave's last tile: its second half is the seam's, or ends the q-tile
                        wait_and_sync();
                        ev(2, t);
                        hstep<Tr, D, ORD, 1, 1, 1, 1, 1>(lds, k_e, v_e, vcur, k1, sA, sB, acc, fin, pend, c2,
                                                          mask_bits(cw, kbase + 64), kbase + 64, cw.qbase, cw.klast, diag0, ninf16, kpre, dma_hook);
                        ev(3, t);
                        step_done();
                    }
                }
                if (!full_wave) {
                    wait_and_sync();
                    ev(2, t);
                    qf_on = qf_ok && ntw + 1 == nt;         // (this is H2(nt-2): the request, H1(nt-2), this half-step's pieces)
                    hstep<Tr, D, ORD, 1, 1, 0, 1, 1>(lds, k_e, v_e, vcur, ring_next(kcur), sA, sB, acc, fin, pend, c2,
                                                         0, 0, cw.qbase, cw.klast, diag0, ninf16, kpre, dma_qf_hook);
                    if (qf_on) { q_in = true; qf_on = false; }
                    ev(3, t);
                    // this wave's rows are stored in the time it would otherwise idle: one block now (the working waves
                    // reach the next barrier one half-step from here), the other in the next step -- an idle one, or the
                    // half-step that joins the next q-tile while the wave that owns the diagonal's end runs its own epilogue
                    epilogue_q(load_epi(b, h), cw, 0, true);
                    ev(4, t);
                    epi_pending = true;
                    step_done();
                }
            }
            // ---- phase C: idle steps behind the diagonal: keep staging for the others ----
            if (!full_wave) {
                while (t + 1 < nt) {
                    wait_and_sync();
                    ev(2, t);
                    produce_v();
                    produce_k();
                    ev(3, t);
                    if (epi_pending) { epilogue_q(load_epi(b, h), cw, 1, true); epi_pending = false; ev(4, t); }
                    if (qf_ok && t == nt - 2) { fetch_q_issue(); q_in = true; }     // (requested behind step nt-3)
                    step_done();
                }
            }
            // here every wave stands in front of the last barrier of q-tile cc
            istamp(4);
            istore(ntw, nt);
            if (!chained) break;
            prev_full = full_wave;
            pb = b; ph = h; pw = cw;
            cc = nx;
        }
        // ======== the last q-tile of the list: nothing to score behind it ========
        wait_and_sync();
        if (full_wave) {
            hstep<Tr, D, ORD, 1, 1, 0, 1, 1>(lds, k_e, v_e, vcur, ring_next(kcur), sA, sB, acc, fin, pend, c2,
                                                  0, 0, cw.qbase, cw.klast, diag0, ninf16, kpre, dma_hook);
            epilogue(cc.b, cc.h, cw, true);
        } else if (epi_pending) {
            epilogue_q(load_epi(cc.b, cc.h), cw, 1, true);
        }
    }
    // drain: the last steps' pieces must have landed before the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <class Tr, int D, int ORD, int DIAG = 0>
int launch_w4_t(const PrefillKernelParams &p, bool causal, hipStream_t stream) {
    using namespace w4;
    const int lds = Img<D>::TOTAL;
    // one workgroup per CU, fewer when the XCD lists are shorter than 32 units
    const int nq = (p.Sq + kRows - 1) / kRows;
    const int U = causal ? (nq + 1) / 2 : nq;
    const long long units_xcd = (long long)p.bh_per_xcd * U;
    const int nslots = (int)(units_xcd < 32 ? units_xcd : 32);
    W4Args a;
    a.q = p.q; a.o = p.o; a.lse = p.lse;
    a.qs0 = p.qs[0]; a.qs1 = p.qs[1]; a.os0 = p.os[0]; a.os1 = p.os[1]; a.os2 = p.os[2];
    a.q_rowb = (unsigned)(2 * p.qs[2]);
    a.Sq = p.Sq;
    a.c2 = p.scale_log2;
    a.kv.k = p.k; a.kv.v = p.v;
    a.kv.ks0 = p.ks[0]; a.kv.ks1 = p.ks[1]; a.kv.vs0 = p.vs[0]; a.kv.vs1 = p.vs[1];
    a.kv.k_rowb = (unsigned)(2 * p.ks[2]); a.kv.v_rowb = (unsigned)(2 * p.vs[2]);
    a.kv.k_extent = (int)((long long)(p.Sk - 1) * a.kv.k_rowb + 2 * D);
    a.kv.v_extent = (int)((long long)(p.Sk - 1) * a.kv.v_rowb + 2 * D);
    a.cur.U = U; a.cur.Hq = p.Hq; a.cur.bh_per_xcd = p.bh_per_xcd; a.cur.BH = p.B * p.Hq; a.cur.nq = nq;
    a.cur.Sk = p.Sk; a.cur.coff = p.Sk - p.Sq; a.cur.G = p.Hq / p.Hkv;
    a.cur.adv_hl = nslots / U; a.cur.adv_i = nslots - a.cur.adv_hl * U;
    a.cur.adv_b = a.cur.adv_hl / p.Hq; a.cur.adv_h = a.cur.adv_hl - a.cur.adv_b * p.Hq;
    dim3 grid(8u * nslots), block(kThreadsW4);
    static DynLdsAttr attr_c, attr_f;
    if (const int rc = causal ? attr_c.ensure(reinterpret_cast<const void *>(&prefill_w4_kernel<Tr, D, true, ORD, DIAG>), lds, "prefill_w4_kernel")
                              : attr_f.ensure(reinterpret_cast<const void *>(&prefill_w4_kernel<Tr, D, false, ORD, DIAG>), lds, "prefill_w4_kernel"))
        return rc;
    if (causal) hipLaunchKernelGGL((prefill_w4_kernel<Tr, D, true, ORD, DIAG>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((prefill_w4_kernel<Tr, D, false, ORD, DIAG>), grid, block, lds, stream, a);
    return check_launch("prefill_w4_kernel");
}

}  // namespace

#ifdef SFA_W4_PART
// One flavour per translation unit (prefill_w4_kernel_p1..3.hip include this file with SFA_W4_PART set): the four flavours
// compile side by side instead of one after the other (each takes about two minutes).
#if SFA_W4_PART == 1
int launch_prefill_w4_fp16_exact(const PrefillKernelParams &p, bool causal, hipStream_t stream) { return launch_w4_t<Fp16, 128, 2>(p, causal, stream); }
#elif SFA_W4_PART == 2
int launch_prefill_w4_fp16_prescaled(const PrefillKernelParams &p, bool causal, hipStream_t stream) { return launch_w4_t<Fp16, 128, 6>(p, causal, stream); }
#elif SFA_W4_PART == 3
int launch_prefill_w4_bf16_prescaled(const PrefillKernelParams &p, bool causal, hipStream_t stream) { return launch_w4_t<Bf16, 128, 6>(p, causal, stream); }
#else
#error "SFA_W4_PART is 1, 2 or 3"
#endif
#else   // the main translation unit: bf16 exact (the headline kernel), the diagnostics of the A/B library, the entry point

// The K / V / Q rows of one head are addressed through 32-bit buffer descriptors: a head whose rows span 2 GiB or
// more, or a row stride of 16 MiB or more, is served by the 8-wave kernel instead (prefill_dispatch.hip).
bool prefill_w4_serves(const PrefillKernelParams &p, int head_dim) {
    if (head_dim != 128) return false;
    const long long k_ext = (long long)(p.Sk - 1) * 2 * p.ks[2] + 2 * head_dim, v_ext = (long long)(p.Sk - 1) * 2 * p.vs[2] + 2 * head_dim;
    const long long q_ext = (long long)(p.Sq - 1) * 2 * p.qs[2] + 2 * head_dim;
    return k_ext < (1ll << 31) && v_ext < (1ll << 31) && q_ext < (1ll << 31) && p.ks[2] * 2 < (1ll << 24) && p.vs[2] * 2 < (1ll << 24) &&
           p.qs[2] * 2 < (1ll << 24);
}

// force: 0 = flavour by policy (exact unless the caller opted into fast_scale), 1 = prescaled, 2 = exact
int launch_prefill_w4(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream, int force) {
    if (dtype != SFA_DTYPE_FP16 && dtype != SFA_DTYPE_BF16)
        return fail(SFA_ERR_BAD_DTYPE, "sfa_prefill_fwd: dtype %d is not fp16(0)/bf16(1)", dtype);
    if (head_dim != 128)
        return fail(SFA_ERR_UNSUPPORTED_HEAD_DIM, "sfa_prefill_fwd: the 4-wave kernel serves head_dim 128 (got %d)", head_dim);
    if (!prefill_w4_serves(p, head_dim))
        return fail(SFA_ERR_BAD_SHAPE, "sfa_prefill_fwd: one head's Q/K/V rows span more than 2 GiB");
    const bool prescaled = force == 0 ? p.fast_scale != 0 : force == 1;
    (void)prescaled;
#ifdef SFA_WITH_VARIANTS        // the q-tile stamping build (tools/w4_item_stamps.py): the A/B library only
    if (force == 3) return launch_w4_t<Bf16, 128, 2, 256>(p, causal, stream);
    if (force == 4) return launch_w4_t<Bf16, 128, 2, 1>(p, causal, stream);         // the event log (tools/w4_events.py)
#endif
#ifdef SFA_W4_DEV       // development builds: one flavour, one dtype (seconds instead of minutes to compile)
    return launch_w4_t<Bf16, 128, 2>(p, causal, stream);
#else
    if (dtype == SFA_DTYPE_FP16)
        return prescaled ? launch_prefill_w4_fp16_prescaled(p, causal, stream) : launch_prefill_w4_fp16_exact(p, causal, stream);
    return prescaled ? launch_prefill_w4_bf16_prescaled(p, causal, stream) : launch_w4_t<Bf16, 128, 2>(p, causal, stream);
#endif
}
#endif  // SFA_W4_PART

}  // namespace sfa
