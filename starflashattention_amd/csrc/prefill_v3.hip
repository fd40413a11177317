// Fused attention forward (prefill) for gfx950 (MI355X), generation 3: half-tile software pipeline,
// explicitly slot-ordered, no bubbles between half-steps.
// bf16 / fp16, head_dim 64 / 128, causal or full, MHA or GQA.  The reference has no prefill
// kernel; this is the entry point BASELINE.json's headline metric is quoted on (SURVEY.md
// section 8(a) row A-new).
//
// MI355X design (MFMA-bound: AI = 1024 FLOP/B at S=4096, D=128):
//   * workgroup = 8 waves = 256 query rows of one (batch, head); wave w owns rows
//     [32w, 32w+32).  K/V tiles of 64 keys are staged once per workgroup into LDS
//     (register-staged: global loads in flight for a whole tile time, then ds_write; K double-,
//     V triple-buffered; ONE barrier per tile) and shared by all 8 waves.
//   * S^T = K . Q^T with v_mfma_f32_32x32x16 (A = K rows from LDS by ds_read_b128, B = Q^T held
//     in registers for the whole kernel): the 32x32 accumulator has the QUERY on the lane and
//     keys in registers, so the online-softmax row max / row sum are in-lane loops plus one
//     v_permlane32_swap -- no LDS, no ds_bpermute.
//   * O^T += V^T . P^T: the S^T accumulator, exponentiated and converted to 16 bit in place, is
//     already the B operand of the second MFMA (it sums over the accumulator's row index);
//     A = V^T comes from the row-major V tile through ds_read_b64_tr_b16 (hardware transpose).
//     The O^T accumulator again has the query on the lane: the rescale is one scalar per lane.
//   * Software pipeline at HALF-tile (32-key) granularity inside each wave (MFMA and VALU are
//     separate pipes).  Two 16-register score accumulators A (keys 0-31 of a tile) and B (keys
//     32-63) alternate roles; every basic block pairs 16 MFMAs with one half-tile of VALU:
//        H1(t): QK^T(B_t)     || exp/convert(A_t),   then  PV(A_t) || rowmax(B_t)
//        H2(t): QK^T(A_{t+1}) || exp/convert(B_t),   then  PV(B_t) || rowmax(A_{t+1})
//     Only 32 score registers are live (a full-tile pipeline needs 64 and spills at 2 waves/SIMD).
//   * Every half-step is written as 16 (8 for D=64) SLOTS in program order -- one MFMA, the LDS
//     fragment reads that feed the MFMA two slots later, and a slice of the softmax VALU work --
//     fenced with sched_barrier(0) so hipcc keeps that order.  This bounds the fragments in
//     flight (3 K + 3 V fragments = 24 VGPRs; the compiler's own clustering hoisted 50+ and
//     spilled, and a spill reload's vmcnt(0) then drained the in-flight K/V staging loads).
//   * No serial section between half-steps: the row max of the scores a half-step is about to
//     exponentiate is finished (v_permlane32_swap, wave-uniform rescale decision) in its slot 0,
//     next to the first QK^T MFMA, and the first two K fragments of a half-step are read from LDS
//     during the last slots of the previous one.
//   * Lazy rescale: O and the row sum are rescaled only when some row max in the wave grew by
//     more than 2^8 over the reference max (wave-uniform branch, almost never taken after the
//     first tiles).  exp2 arguments stay <= 8, so P <= 256: bf16/fp16 keep the same RELATIVE
//     precision and the fp32 accumulators have ample headroom.
//   * LDS images: K rows XOR-swizzled for conflict-free ds_read_b128 of the A operand; V rows
//     XOR-swizzled for conflict-free transposed reads (cdna_hip_programming.md T2 / T10).
//   * blockIdx -> (head, q-tile) is XCD-aware (prefill_common.h).
#include "prefill_common.h"

namespace sfa {

namespace {

using namespace prefill;

constexpr float kRescaleThr = 8.0f;     // log2 units

// LDS images, PADDED rows (not XOR-swizzled): every read address is then one lane-constant base
// plus a compile-time immediate, so the whole kernel needs two LDS address registers instead
// of ~16 (the XOR images of v0 made hipcc spill address VGPRs at 2 waves/SIMD).
//   K rows: 2*D + 16 bytes.  ds_read_b128 lane groups read 16 rows (distinct mod 16) at one chunk:
//           slot = (17*row + ch) mod 16 (D=128), (9*row + ch) mod 16 (D=64) -> conflict-free.
//   V rows: 2*D + 64 bytes.  a 32-lane half of ds_read_b64_tr_b16 reads 4 consecutive rows x 64
//           contiguous bytes: 320q mod 256 = 64q (D=128), 192q mod 256 = {0,192,128,64} (D=64)
//           -> the four rows tile the 256-byte bank row, conflict-free.
template <int D> struct Lds {
    static constexpr int KS = 2 * D + 16;           // K row stride (bytes)
    static constexpr int VS = 2 * D + 64;           // V row stride
    static constexpr int KTILE = kBN * KS;
    static constexpr int VTILE = kBN * VS;
    static constexpr int V_BASE = 2 * KTILE;        // K[2] then V[3]
    static constexpr int TOTAL = 2 * KTILE + 3 * VTILE;
    static_assert(3 * VTILE < 65536 && 2 * KTILE < 65536, "ds immediates are 16 bit");
};

// key of register r = kbase + (r&3) + 8*(r>>2) + 4*h2
__device__ __forceinline__ void mask_half(f32x16 &s, int kbase, int h2, int lim) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (kbase + (r & 3) + 8 * (r >> 2) + 4 * h2 > lim) s[r] = ninf();
}

__device__ __forceinline__ float lane_rowmax(const f32x16 &s) {
    float m0 = fmaxf(s[0], s[4]), m1 = fmaxf(s[1], s[5]), m2 = fmaxf(s[2], s[6]), m3 = fmaxf(s[3], s[7]);
    m0 = fmaxf(m0, fmaxf(s[8], s[12]));
    m1 = fmaxf(m1, fmaxf(s[9], s[13]));
    m2 = fmaxf(m2, fmaxf(s[10], s[14]));
    m3 = fmaxf(m3, fmaxf(s[11], s[15]));
    return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}

#define SFA_FENCE() __builtin_amdgcn_sched_barrier(0)

// One pipelined half-step in explicit slot order:
//   sN <- scores of K rows [32*HN, +32) of the tile at kb      (DO_QK; NKS MFMAs)
//   sO  = scores of keys [32*HO, +32) of the tile whose V is at vb: row max finished (slot 0),
//         exponentiated in place, packed to 16 bit, O^T += V^T . P^T      (NPV MFMAs)
// kb / vb / kb_pref already include this lane's read base (Lds<D> comment).
//   kpre[PF]  in: first PF K fragments of this half-step (read from LDS earlier);
//             out (PREF): first PF fragments of the next half-step, rows [32*(1-HN), +32) at kb_pref
//   PF        how many slots ahead of its MFMA a fragment is read
//   mxO       in: this lane's max over the 16 scores in sO (before masking)
//   mxN       out: this lane's max over the 16 new scores
template <class Tr, int D, bool CAUSAL, int PF, int HN, int HO, bool DO_QK, bool PREF, int ABL = 0>
__device__ __forceinline__ void h_block(const char *kb, const char *vb, const char *kb_pref,
                                        const typename Tr::mfma_vec (&qf)[D / 16], f32x16 &sN, f32x16 &sO,
                                        f32x16 (&o)[D / 32], float c2, float &msc, float &lsum,
                                        float mxO, float &mxN, bool mask_o, int kbase_o, int h2, int lim,
                                        typename Tr::mfma_vec (&kpre)[PF]) {
    using Vec = typename Tr::mfma_vec;
    constexpr int NKS = D / 16, NDB = D / 32, NPV = 2 * NDB;
    constexpr int KS = Lds<D>::KS, VS = Lds<D>::VS;
    constexpr int EP = 16 / NPV;            // elements per early PV slot     (elements 8..15)
    constexpr int EM = 32 / NPV;            // new scores max-ed per late PV slot

    auto ld_k = [&](int ks) -> Vec {
        if (ABL & 16) return kpre[0];       // timing-only ablation: no K fragment reads
        return bitcast<Vec>(*reinterpret_cast<const uint4 *>(kb + KS * 32 * HN + 32 * ks));
    };
    auto ld_v = [&](int j) -> Vec {         // A operand of PV MFMA j: d block j % NDB, k-step j / NDB
        if (ABL & 16) return kpre[1];       // timing-only ablation: no V fragment reads
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
    // ---- slot 0: first QK MFMA next to the finish of sO's row max ----
    if (DO_QK) {
#pragma unroll
        for (int i = 0; i < PF; ++i) kf[i] = kpre[i];
        if (PF < NKS) kf[PF] = ld_k(PF); else vf[PF - NKS] = ld_v(PF - NKS);
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        sN = Tr::mfma32(kf[0], qf[0], z);
    } else {
#pragma unroll
        for (int i = 0; i < PF; ++i) vf[i] = ld_v(i);
    }
    {
        float mxl = mxO;
        if (mask_o) {                                   // wave-uniform, diagonal / ragged tiles only
            mask_half(sO, kbase_o, h2, lim);
            mxl = lane_rowmax(sO);
        }
        const float mx = half_max(mxl) * c2;            // both lane halves hold the same query
        if (__any(mx > msc + kRescaleThr)) {            // rare after the first tiles
            const float mnew = fmaxf(msc, mx);
            const float alpha = (mnew == ninf()) ? 1.0f : fast_exp2(msc - mnew);
            msc = mnew;
            lsum *= alpha;
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
    }
    const float msafe = (msc == ninf()) ? 0.f : msc;
    SFA_FENCE();

    uint32_t pk[8];                         // P^T packed: pk[4k .. 4k+3] is the B operand of k-step k
    float rs0 = 0.f, rs1 = 0.f;
    auto soft1 = [&](int e) {               // one element; packs when the pair is complete
        if (ABL & 8) {                      // timing-only ablation: no softmax VALU at all
            if (e & 1) pk[e >> 1] = bitcast<uint32_t>(sO[e]);
            return;
        }
        if (ABL & 2) sO[e] = fmaf(sO[e], c2, -msafe);      // timing-only ablation: no v_exp
        else sO[e] = fast_exp2(fmaf(sO[e], c2, -msafe));
        if (e & 1) { rs1 += sO[e]; pk[e >> 1] = Tr::pack2(sO[e - 1], sO[e]); } else { rs0 += sO[e]; }
    };
    auto soft2 = [&](int e) { soft1(e); soft1(e + 1); };

    if (DO_QK) {
#pragma unroll
        for (int i = 1; i < NKS; ++i) {     // elements 0..7 spread over slots 1..NKS-1
            if (i + PF < NKS) kf[i + PF] = ld_k(i + PF); else vf[i + PF - NKS] = ld_v(i + PF - NKS);
            sN = Tr::mfma32(kf[i], qf[i], sN);
#pragma unroll
            for (int e = (i - 1) * 8 / (NKS - 1); e < i * 8 / (NKS - 1); ++e) soft1(e);
            SFA_FENCE();
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; e += 2) soft2(e);
        SFA_FENCE();
    }
    float m0 = ninf(), m1 = ninf();
#pragma unroll
    for (int j = 0; j < NPV; ++j) {
        if (j + PF < NPV) {
            vf[j + PF] = ld_v(j + PF);
        } else if (PREF) {                  // last PF slots: first K fragments of the next half-step
            kpre[j + PF - NPV] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(
                kb_pref + KS * 32 * (1 - HN) + 32 * (j + PF - NPV)));
        }
        uint4 w;
        w.x = pk[4 * (j / NDB) + 0]; w.y = pk[4 * (j / NDB) + 1];
        w.z = pk[4 * (j / NDB) + 2]; w.w = pk[4 * (j / NDB) + 3];
        o[j % NDB] = Tr::mfma32(vf[j], bitcast<Vec>(w), o[j % NDB]);
        if (j < NPV / 2) {
#pragma unroll
            for (int e = 0; e < EP; e += 2) soft2(8 + EP * j + e);
        } else if (DO_QK) {
#pragma unroll
            for (int e = 0; e < EM; e += 2) {
                m0 = fmaxf(m0, sN[EM * (j - NPV / 2) + e]);
                m1 = fmaxf(m1, sN[EM * (j - NPV / 2) + e + 1]);
            }
        }
        SFA_FENCE();
    }
    lsum += rs0 + rs1;
    mxN = fmaxf(m0, m1);
}

template <class Tr, int D, bool CAUSAL, int PF, int ABL = 0>
__global__ void __launch_bounds__(kThreads, 2)
prefill_kernel_v3(const PrefillKernelParams p) {
    using Vec = typename Tr::mfma_vec;
    constexpr int NKS = D / 16;                 // k-steps of Q.K^T
    constexpr int NDB = D / 32;                 // 32-wide d blocks of O^T
    constexpr int CPR = D / 8;                  // 16-B chunks per row
    constexpr int NLD = kBN * CPR / kThreads;   // chunks staged per thread per tile (2 or 1)
    using L = Lds<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const BlockCoord bc = block_coord(p);
    if (bc.bh >= p.B * p.Hq) return;
    const int b = bc.bh / p.Hq, h = bc.bh % p.Hq;
    const int hk = h / (p.Hq / p.Hkv);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform
    const int l31 = lane & 31, h2 = lane >> 5;
    const int q0 = bc.qt * kBM;
    const int wq0 = q0 + 32 * wave;             // this wave's first query row
    const int qrow = wq0 + l31;
    const int coff = p.Sk - p.Sq;               // causal: key j visible iff j <= i + coff

    // ---- Q^T fragments (B operand): lane holds Q[qrow][16ks + 8*h2 .. +8] ----
    Vec qf[NKS];
    {
        const uint16_t *qp = p.q + b * p.qs[0] + h * p.qs[1] + (long long)min(qrow, p.Sq - 1) * p.qs[2] + 8 * h2;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            qf[ks] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(qp + 16 * ks));
    }

    // tiles the workgroup walks / tiles this wave computes on (both wave-uniform)
    int kv_end = p.Sk;
    if (CAUSAL) kv_end = min(p.Sk, q0 + kBM + coff);
    const int nt = kv_end > 0 ? (kv_end + kBN - 1) / kBN : 0;
    int ntw = nt;
    if (CAUSAL) ntw = (wq0 + 31 + coff >= 0) ? min(nt, (wq0 + 31 + coff) / kBN + 1) : 0;
    const int lim = CAUSAL ? min(p.Sk - 1, qrow + coff) : p.Sk - 1;       // last visible key of this row

    // ---- staging: thread owns chunk(s) c = tid (+512) of every tile ----
    const int st_row0 = tid / CPR, st_ch = tid % CPR;
    const int st_row1 = (tid + kThreads) / CPR;                 // NLD == 2 only
    const uint16_t *kg = p.k + b * p.ks[0] + hk * p.ks[1] + st_ch * 8;
    const uint16_t *vg = p.v + b * p.vs[0] + hk * p.vs[1] + st_ch * 8;
    char *const k_w = smem + L::KS * st_row0 + 16 * st_ch;                 // row1 = row0 + 512/CPR
    char *const v_w = smem + L::V_BASE + L::VS * st_row0 + 16 * st_ch;
    constexpr int ROW1 = kThreads / CPR;
    uint4 kr0, kr1, vr0, vr1;       // plain scalars: arrays of these ended up in scratch
    kr0 = kr1 = vr0 = vr1 = make_uint4(0, 0, 0, 0);

#define SFA_LOAD_KV(KT)                                                                             \
    do {                                                                                            \
        const long long r0_ = min((KT) * kBN + st_row0, p.Sk - 1);                                   \
        kr0 = *reinterpret_cast<const uint4 *>(kg + r0_ * p.ks[2]);                                  \
        vr0 = *reinterpret_cast<const uint4 *>(vg + r0_ * p.vs[2]);                                  \
        if (NLD > 1) {                                                                              \
            const long long r1_ = min((KT) * kBN + st_row1, p.Sk - 1);                               \
            kr1 = *reinterpret_cast<const uint4 *>(kg + r1_ * p.ks[2]);                              \
            vr1 = *reinterpret_cast<const uint4 *>(vg + r1_ * p.vs[2]);                              \
        }                                                                                           \
    } while (0)
#define SFA_STORE_KV(KBUF, VBUF)                                                                    \
    do {                                                                                            \
        *reinterpret_cast<uint4 *>(k_w + (KBUF)) = kr0;         /* KBUF/VBUF: byte offsets */       \
        *reinterpret_cast<uint4 *>(v_w + (VBUF)) = vr0;                                             \
        if (NLD > 1) {                                                                              \
            *reinterpret_cast<uint4 *>(k_w + (KBUF) + ROW1 * L::KS) = kr1;                           \
            *reinterpret_cast<uint4 *>(v_w + (VBUF) + ROW1 * L::VS) = vr1;                           \
        }                                                                                           \
    } while (0)

    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float msc = ninf();     // reference max the exponentials are taken against (log2 units)
    float lsum = 0.f;       // this lane's share of the running row sum
    const float c2 = p.scale_log2;
    // the two LDS read bases of this lane (everything else is an immediate)
    const char *const k_rd = smem + L::KS * l31 + 16 * h2;                 // K row l31, chunk h2
    const char *const v_rd = smem + L::V_BASE + L::VS * (4 * h2 + ((lane & 15) >> 2)) +
                             32 * ((lane >> 4) & 1) + 16 * ((lane & 3) >> 1) + 8 * (lane & 1);

#define SFA_NEEDS_MASK(KBASE) ((CAUSAL && ((KBASE) + 31 > wq0 + coff)) || ((KBASE) + 32 > p.Sk))

    // ---- prologue: tile 0 into LDS, tile 1 in flight, scores of the first half-tile ----
    f32x16 sA, sB;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sA[r] = 0.f; sB[r] = 0.f; }
    if (nt > 0) {
        SFA_LOAD_KV(0);
        SFA_STORE_KV(0, 0);
    }
    __syncthreads();
    SFA_LOAD_KV(1);
    float mxA = ninf(), mxB = ninf();       // lane-local maxima of the pending score half-tiles
    Vec kpre[PF];                           // first PF K fragments of the next half-step
#pragma unroll
    for (int i = 0; i < PF; ++i) kpre[i] = bitcast<Vec>(make_uint4(0, 0, 0, 0));
    if (ntw > 0) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            sA = Tr::mfma32(bitcast<Vec>(*reinterpret_cast<const uint4 *>(k_rd + 32 * ks)), qf[ks], sA);
#pragma unroll
        for (int i = 0; i < PF; ++i)
            kpre[i] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(k_rd + L::KS * 32 + 32 * i));
        mxA = lane_rowmax(sA);
    }

    // Tile t lives in K buffer t % 2 and V buffer t % 3 (offsets kept as scalars, added to the two
    // lane bases once per step).  Buffer safety with ONE barrier per tile: K(t+1) overwrites
    // K(t-1), last read in H1(t-1), i.e. before barrier(t-1); V(t+1) overwrites V(t-2), last
    // read in H2(t-2), i.e. before barrier(t-1) as well.  Loads/stores of tiles past the end are
    // row-clamped and land in buffers nobody reads again.
    int kcur = 0, vcur = 0;         // byte offsets of tile t's buffers
    // Stage tile t+1 (loaded one step ago), sync, read the first K fragments of H2, put tile t+2
    // in flight.
#define SFA_STAGE_AND_SYNC(T, WITH_KF)                                                              \
    do {                                                                                            \
        const int vnext_ = (vcur == 2 * L::VTILE) ? 0 : vcur + L::VTILE;                            \
        if (!(ABL & 4)) SFA_STORE_KV(kcur ^ L::KTILE, vnext_);                                      \
        if (!(ABL & 1)) __syncthreads();                                                            \
        if (WITH_KF) {                                                                              \
            _Pragma("unroll") for (int i_ = 0; i_ < PF; ++i_)                                       \
                kpre[i_] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(                           \
                    k_rd + (kcur ^ L::KTILE) + 32 * i_));                                           \
            SFA_FENCE();    /* LDS reads first: the address math of the loads below covers them */   \
        }                                                                                           \
        if (!(ABL & 4)) SFA_LOAD_KV((T) + 2);                                                       \
        SFA_FENCE();                                                                                \
    } while (0)
#define SFA_ADVANCE()                                                                               \
    do {                                                                                            \
        kcur ^= L::KTILE;                                                                           \
        vcur = (vcur == 2 * L::VTILE) ? 0 : vcur + L::VTILE;                                        \
    } while (0)

    // FULL steps: this wave needs tile t+1 as well.
    //   H1(t): QK^T(B_t)     || max,exp(A_t),  PV(A_t) || lane max(B_t)
    //   H2(t): QK^T(A_{t+1}) || max,exp(B_t),  PV(B_t) || lane max(A_{t+1})
    int t = 0;
    for (; t + 1 < ntw; ++t) {
        const char *kb = k_rd + kcur, *vb = v_rd + vcur;
        h_block<Tr, D, CAUSAL, PF, 1, 0, true, false, ABL>(kb, vb, kb, qf, sB, sA, o, c2, msc, lsum, mxA, mxB,
                                                  SFA_NEEDS_MASK(t * kBN), t * kBN, h2, lim, kpre);
        SFA_STAGE_AND_SYNC(t, true);
        const char *kbn = k_rd + (kcur ^ L::KTILE);
        h_block<Tr, D, CAUSAL, PF, 0, 1, true, true, ABL>(kbn, vb, kbn, qf, sA, sB, o, c2, msc, lsum, mxB, mxA,
                                                 SFA_NEEDS_MASK(t * kBN + 32), t * kBN + 32, h2, lim, kpre);
        SFA_ADVANCE();
    }
    // TAIL step: this wave's last tile (no next scores to compute).
    if (t < ntw) {
        const char *kb = k_rd + kcur, *vb = v_rd + vcur;
        h_block<Tr, D, CAUSAL, PF, 1, 0, true, false, ABL>(kb, vb, kb, qf, sB, sA, o, c2, msc, lsum, mxA, mxB,
                                                  SFA_NEEDS_MASK(t * kBN), t * kBN, h2, lim, kpre);
        SFA_STAGE_AND_SYNC(t, false);
        h_block<Tr, D, CAUSAL, PF, 0, 1, false, false, ABL>(kb, vb, kb, qf, sA, sB, o, c2, msc, lsum, mxB, mxA,
                                                   SFA_NEEDS_MASK(t * kBN + 32), t * kBN + 32, h2, lim, kpre);
        SFA_ADVANCE();
        ++t;
    }
    // idle steps (causal: tiles beyond this wave's diagonal): keep staging for the other waves.
    for (; t < nt; ++t) {
        SFA_STAGE_AND_SYNC(t, false);
        SFA_ADVANCE();
    }
#undef SFA_STAGE_AND_SYNC
#undef SFA_ADVANCE
#undef SFA_NEEDS_MASK

    // ---- epilogue: normalise, convert, store O[qrow][:] (lane holds 4 consecutive d per group) ----
    const float ltot = half_sum(lsum);
    const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
    if (qrow < p.Sq) {
        uint16_t *op = p.o + b * p.os[0] + h * p.os[1] + (long long)qrow * p.os[2] + 4 * h2;
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = Tr::pack2(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv);
                w.y = Tr::pack2(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
                *reinterpret_cast<uint2 *>(op + 32 * d + 8 * g) = w;
            }
        }
        if (p.lse && h2 == 0) {
            const float lse = ltot > 0.f ? (msc + __log2f(ltot)) * kLn2 : ninf();
            p.lse[((long long)b * p.Hq + h) * p.Sq + qrow] = lse;
        }
    }
#undef SFA_LOAD_KV
#undef SFA_STORE_KV
}

template <class Tr, int D, int PF>
int launch_t(const PrefillKernelParams &p, bool causal, hipStream_t stream) {
    const size_t lds = Lds<D>::TOTAL;      // K[2] + V[3], padded rows
    dim3 grid(8u * p.bh_per_xcd * p.nq_tiles), block(kThreads);
    static bool attr_set = false;       // idempotent; a race only repeats the call
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&prefill_kernel_v3<Tr, D, true, PF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&prefill_kernel_v3<Tr, D, false, PF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (causal) {
        hipLaunchKernelGGL((prefill_kernel_v3<Tr, D, true, PF>), grid, block, lds, stream, p);
    } else {
        hipLaunchKernelGGL((prefill_kernel_v3<Tr, D, false, PF>), grid, block, lds, stream, p);
    }
    return check_launch("prefill_kernel_v3");
}

template <int PF>
int launch_pf(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream) {
    if (dtype == SFA_DTYPE_FP16) {
        if (head_dim == 128) return launch_t<Fp16, 128, PF>(p, causal, stream);
        if (head_dim == 64) return launch_t<Fp16, 64, PF>(p, causal, stream);
    } else if (dtype == SFA_DTYPE_BF16) {
        if (head_dim == 128) return launch_t<Bf16, 128, PF>(p, causal, stream);
        if (head_dim == 64) return launch_t<Bf16, 64, PF>(p, causal, stream);
    } else {
        return fail(SFA_ERR_BAD_DTYPE, "sfa_prefill_fwd: dtype %d is not fp16(0)/bf16(1)", dtype);
    }
    return fail(SFA_ERR_UNSUPPORTED_HEAD_DIM, "sfa_prefill_fwd: head_dim %d not in {64, 128}", head_dim);
}

// timing-only ablations of the headline configuration (bf16, D=128); results are WRONG by design
template <int ABL, int PF = 2>
int launch_abl(const PrefillKernelParams &p, bool causal, hipStream_t stream) {
    const size_t lds = Lds<128>::TOTAL;
    dim3 grid(8u * p.bh_per_xcd * p.nq_tiles), block(kThreads);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&prefill_kernel_v3<Bf16, 128, true, PF, ABL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&prefill_kernel_v3<Bf16, 128, false, PF, ABL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (causal) hipLaunchKernelGGL((prefill_kernel_v3<Bf16, 128, true, PF, ABL>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((prefill_kernel_v3<Bf16, 128, false, PF, ABL>), grid, block, lds, stream, p);
    return check_launch("prefill_kernel_v3 (ablation)");
}

}  // namespace

int launch_prefill_ablation(const PrefillKernelParams &p, int abl, int dtype, int head_dim, bool causal,
                            hipStream_t stream) {
    if (dtype != SFA_DTYPE_BF16 || head_dim != 128)
        return fail(SFA_ERR_BAD_SHAPE, "ablation builds exist for bf16, head_dim 128 only");
    switch (abl) {
        case 1: return launch_abl<1>(p, causal, stream);
        case 2: return launch_abl<2>(p, causal, stream);
        case 4: return launch_abl<4>(p, causal, stream);
        case 5: return launch_abl<5>(p, causal, stream);
        case 8: return launch_abl<8>(p, causal, stream);
        case 13: return launch_abl<13>(p, causal, stream);
        case 29: return launch_abl<29>(p, causal, stream);          // + no LDS fragment reads
        case 213: return launch_abl<13, 4>(p, causal, stream);      // ablation 13 at PF = 4
        case 313: return launch_abl<13, 6>(p, causal, stream);      // ablation 13 at PF = 6
        case 200: return launch_abl<0, 4>(p, causal, stream);       // full kernel at PF = 4 (correct)
        case 300: return launch_abl<0, 6>(p, causal, stream);       // full kernel at PF = 6 (correct)
        default: return fail(SFA_ERR_BAD_SHAPE, "no ablation build %d", abl);
    }
}

int launch_prefill_v3(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream) {
    return launch_pf<2>(p, dtype, head_dim, causal, stream);
}
int launch_prefill_v4(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream) {
    return launch_pf<3>(p, dtype, head_dim, causal, stream);
}

}  // namespace sfa
