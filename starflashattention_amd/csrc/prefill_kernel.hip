// Fused attention forward (prefill) for gfx950 (MI355X), the product kernel: half-tile software
// pipeline, explicitly slot-ordered, NQB query blocks per wave.
// bf16 / fp16, head_dim 64 / 128, causal or full, MHA or GQA.  The reference has no prefill
// kernel; this is the entry point BASELINE.json's headline metric is quoted on (SURVEY.md
// section 8(a) row A-new).
//
// MI355X design (MFMA-bound: AI = 1024 FLOP/B at S=4096, D=128):
//   * workgroup = 256 query rows of one (batch, head).  NQB = 1 (shipped): 8 waves of 32 rows, two
//     waves per SIMD.  NQB = 2 (4 waves x 64 rows, one per SIMD, 512-register file, every K / V
//     fragment feeding two MFMAs) is kept as a template parameter: it halves the LDS operand
//     traffic, but as scheduled by hipcc 7.2 it measured 687 vs 868 TFLOPS (needs
//     -mllvm -amdgpu-mfma-vgpr-form=1 to avoid ~2000 v_accvgpr copies; still 336 of them) -- it
//     wants hand-placed AGPR ownership, which is a later round's work.
//   * K/V tiles of 64 keys are staged once per workgroup into LDS and shared by all waves
//     (register-staged: global loads in flight for a whole tile time, then ds_write; K and V
//     triple-buffered, K running two tiles ahead of the compute and V one; ONE barrier per tile, and
//     nothing right behind that barrier depends on what it publishes).  The loads and ds_writes
//     ride inside the MFMA slots of the half-steps instead of bunching up around the barrier.
//   * S^T = K . Q^T with v_mfma_f32_32x32x16 (A = K rows from LDS by ds_read_b128, B = Q^T held
//     in registers for the whole kernel): the 32x32 accumulator has the QUERY on the lane and
//     keys in registers, so the online-softmax row max / row sum are in-lane loops plus one
//     v_permlane32_swap -- no LDS, no ds_bpermute.
//   * O^T += V^T . P^T: the S^T accumulator, exponentiated and converted to 16 bit in place, is
//     already the B operand of the second MFMA (it sums over the accumulator's row index);
//     A = V^T comes from the row-major V tile through ds_read_b64_tr_b16 (hardware transpose).
//     The O^T accumulator again has the query on the lane: the rescale is one scalar per lane.
//   * Software pipeline at HALF-tile (32-key) granularity inside each wave (MFMA and VALU are
//     separate pipes).  Per query block two 16-register score accumulators A (keys 0-31 of a
//     tile) and B (keys 32-63) alternate roles:
//        H1(t): QK^T(B_t)     || max,exp(A_t),   PV(A_t) || lane max(B_t)
//        H2(t): QK^T(A_{t+1}) || max,exp(B_t),   PV(B_t) || lane max(A_{t+1})
//   * Every half-step is written as SLOTS in program order -- one LDS fragment, the NQB MFMAs
//     it feeds, the fragment read PF slots ahead, and a slice of the softmax VALU work -- fenced
//     with sched_barrier(0) so hipcc keeps that order (its own clustering hoisted 50+ fragment
//     registers, spilled, and a spill reload's vmcnt(0) drained the in-flight staging loads).
//     The softmax slices are themselves staged across slots (scale/subtract, v_exp, sum+pack of a
//     pair sit in three consecutive slots) so no VALU instruction issues right behind the one it
//     depends on (+3 %).
//   * Lazy rescale: O and the row sum are rescaled only when some row max in the wave grew by
//     more than 2^8 over the reference max (wave-uniform branch, almost never taken after the
//     first tiles).  exp2 arguments stay <= 8, so P <= 256: bf16/fp16 keep the same RELATIVE
//     precision and the fp32 accumulators have ample headroom.
//   * LDS images use PADDED rows (prefill_common.h would XOR-swizzle): every read address is one
//     lane-constant base plus a compile-time immediate -- two LDS address registers in total --
//     and SQ_LDS_BANK_CONFLICT measures 0.
//   * blockIdx -> (head, q-tile) is XCD-aware (prefill_common.h).
#include "prefill_core.h"

namespace sfa {

namespace {

using namespace prefill;

template <class Tr, int D, bool CAUSAL, int NQB, int PF, int ORD, int DIAG>
__global__ void __launch_bounds__(kThreads / NQB, 2 / NQB)
prefill_kernel(const PrefillKernelParams p) {
    using Vec = typename Tr::mfma_vec;
    constexpr int THREADS = kThreads / NQB;     // 512 (8 waves) or 256 (4 waves)
    constexpr int WROWS = 32 * NQB;             // query rows per wave
    constexpr int NKS = D / 16;                 // k-steps of Q.K^T
    constexpr int NDB = D / 32;                 // 32-wide d blocks of O^T
    constexpr int CPR = D / 8;                  // 16-B chunks per row
    constexpr int NLD = kBN * CPR / THREADS;    // chunks staged per thread per tile (1, 2 or 4)
    constexpr int ROWSTEP = THREADS / CPR;      // row distance between a thread's chunks
    using L = Lds<D>;
    static_assert(NLD >= 1 && NLD <= 4, "staging registers are named kr0..kr3");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const BlockCoord bc = block_coord(p);
    if (bc.bh >= p.B * p.Hq) return;
    const int b = bc.bh / p.Hq, h = bc.bh % p.Hq;
    const int hk = h / (p.Hq / p.Hkv);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform
    const int l31 = lane & 31, h2 = lane >> 5;
    const int q0 = bc.qt * kBM;
    const int wq0 = q0 + WROWS * wave;          // this wave's first query row
    const int coff = p.Sk - p.Sq;               // causal: key j visible iff j <= i + coff

    // ---- Q^T fragments (B operand): lane holds Q[row][16ks + 8*h2 .. +8] of each query block ----
    Vec qf[NQB][NKS];
    int lim[NQB];                               // last visible key of this lane's row, per block
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
        const int qrow = wq0 + 32 * q + l31;
        const uint16_t *qp = p.q + b * p.qs[0] + h * p.qs[1] + (long long)min(qrow, p.Sq - 1) * p.qs[2] + 8 * h2;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            qf[q][ks] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(qp + 16 * ks));
        lim[q] = CAUSAL ? min(p.Sk - 1, qrow + coff) : p.Sk - 1;
    }
    // tiles the workgroup walks / tiles this wave computes on (both wave-uniform)
    int kv_end = p.Sk;
    if (CAUSAL) kv_end = min(p.Sk, q0 + kBM + coff);
    const int nt = kv_end > 0 ? (kv_end + kBN - 1) / kBN : 0;
    int ntw = nt;
    if (CAUSAL) ntw = (wq0 + WROWS - 1 + coff >= 0) ? min(nt, (wq0 + WROWS - 1 + coff) / kBN + 1) : 0;

    // ---- staging: thread owns chunks (row st_row + i*ROWSTEP, chunk st_ch), i < NLD, of every tile ----
    const int st_row = tid / CPR, st_ch = tid % CPR;
    const char *const kg = reinterpret_cast<const char *>(p.k + b * p.ks[0] + hk * p.ks[1]);   // uniform
    const char *const vg = reinterpret_cast<const char *>(p.v + b * p.vs[0] + hk * p.vs[1]);
    const long long k_tile_bytes = 2ll * kBN * p.ks[2], v_tile_bytes = 2ll * kBN * p.vs[2];
    const unsigned k_rowb = (unsigned)(2 * p.ks[2]), v_rowb = (unsigned)(2 * p.vs[2]);
    char *const k_w = smem + L::KS * st_row + 16 * st_ch;
    char *const v_w = smem + L::V_BASE + L::VS * st_row + 16 * st_ch;
    uint4 kr0, kr1, kr2, kr3, vr0, vr1, vr2, vr3;   // plain scalars: arrays of these ended up in scratch
    kr0 = kr1 = kr2 = kr3 = vr0 = vr1 = vr2 = vr3 = make_uint4(0, 0, 0, 0);

    // Load this thread's chunks of tile KT.  Whole tiles: uniform tile base + 32-bit lane offset
    // (no per-load 64-bit VALU math).  The ragged last tile (and the row-clamped dummy tiles the
    // pipeline requests past the end) clamp every row to Sk-1.
#define SFA_LD1(DST, G, ROWB, TB, I, KT)                                                            \
    DST = *reinterpret_cast<const uint4 *>(                                                         \
        (G) + (KT) * (TB) + (unsigned)(st_row + (I) * ROWSTEP) * (ROWB) + 16u * st_ch)
#define SFA_LD1C(DST, G, ROWB, I, KT)                                                               \
    DST = *reinterpret_cast<const uint4 *>(                                                         \
        (G) + (long long)min((KT) * kBN + st_row + (I) * ROWSTEP, p.Sk - 1) * (ROWB) + 16u * st_ch)
#define SFA_LOAD_ONE(KR0, KR1, KR2, KR3, G, ROWB, TB, KT)                                            \
    do {                                                                                            \
        const int kt_ = (KT);                                                                       \
        if ((kt_ + 1) * kBN <= p.Sk) {                                                              \
            SFA_LD1(KR0, G, ROWB, TB, 0, kt_);                                                      \
            if (NLD > 1) SFA_LD1(KR1, G, ROWB, TB, 1, kt_);                                         \
            if (NLD > 2) SFA_LD1(KR2, G, ROWB, TB, 2, kt_);                                         \
            if (NLD > 3) SFA_LD1(KR3, G, ROWB, TB, 3, kt_);                                         \
        } else {                                                                                    \
            SFA_LD1C(KR0, G, ROWB, 0, kt_);                                                         \
            if (NLD > 1) SFA_LD1C(KR1, G, ROWB, 1, kt_);                                            \
            if (NLD > 2) SFA_LD1C(KR2, G, ROWB, 2, kt_);                                            \
            if (NLD > 3) SFA_LD1C(KR3, G, ROWB, 3, kt_);                                            \
        }                                                                                           \
    } while (0)
#define SFA_LOAD_K(KT) SFA_LOAD_ONE(kr0, kr1, kr2, kr3, kg, k_rowb, k_tile_bytes, KT)
#define SFA_LOAD_V(KT) SFA_LOAD_ONE(vr0, vr1, vr2, vr3, vg, v_rowb, v_tile_bytes, KT)
#define SFA_STORE_ONE(W, RS, BUF, R0, R1, R2, R3)                                                   \
    do {                                                                                            \
        *reinterpret_cast<uint4 *>((W) + (BUF)) = R0;           /* BUF: byte offset */              \
        if (NLD > 1) *reinterpret_cast<uint4 *>((W) + (BUF) + ROWSTEP * (RS)) = R1;                  \
        if (NLD > 2) *reinterpret_cast<uint4 *>((W) + (BUF) + 2 * ROWSTEP * (RS)) = R2;              \
        if (NLD > 3) *reinterpret_cast<uint4 *>((W) + (BUF) + 3 * ROWSTEP * (RS)) = R3;              \
    } while (0)
#define SFA_STORE_K(KBUF) SFA_STORE_ONE(k_w, L::KS, KBUF, kr0, kr1, kr2, kr3)
#define SFA_STORE_V(VBUF) SFA_STORE_ONE(v_w, L::VS, VBUF, vr0, vr1, vr2, vr3)

    // The same staging, one chunk at a time (op n < 2*NLD: even = K chunk n/2, odd = V chunk n/2), so
    // the FULL steps can issue the loads inside H2's first QK slots and the ds_writes inside H1's
    // PV slots instead of bunching them around the barrier.  load_op is the whole-tile fast path
    // only (uniform tile base + lane offset, no clamping); ragged tiles take SFA_LOAD_K / _V.
    constexpr int NOPS = 2 * NLD;
    auto load_op = [&](int n, int kt_k, int kt_v) {
#define SFA_LDOP(N, KR, VR, I)                                                                      \
        if (n == (N)) SFA_LD1(KR, kg, k_rowb, k_tile_bytes, I, kt_k);                               \
        if (n == (N) + 1) SFA_LD1(VR, vg, v_rowb, v_tile_bytes, I, kt_v);
        SFA_LDOP(0, kr0, vr0, 0)
        if (NLD > 1) { SFA_LDOP(2, kr1, vr1, 1) }
        if (NLD > 2) { SFA_LDOP(4, kr2, vr2, 2) }
        if (NLD > 3) { SFA_LDOP(6, kr3, vr3, 3) }
#undef SFA_LDOP
    };
    auto store_op = [&](int n, int kbuf, int vbuf) {
#define SFA_STOP(N, KR, VR, I)                                                                      \
        if (n == (N)) *reinterpret_cast<uint4 *>(k_w + kbuf + (I) * ROWSTEP * L::KS) = KR;           \
        if (n == (N) + 1) *reinterpret_cast<uint4 *>(v_w + vbuf + (I) * ROWSTEP * L::VS) = VR;
        SFA_STOP(0, kr0, vr0, 0)
        if (NLD > 1) { SFA_STOP(2, kr1, vr1, 1) }
        if (NLD > 2) { SFA_STOP(4, kr2, vr2, 2) }
        if (NLD > 3) { SFA_STOP(6, kr3, vr3, 3) }
#undef SFA_STOP
    };

    Acc<D, NQB> acc;
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc.o[q][d][r] = 0.f;
        acc.msc[q] = ninf();
        acc.lsum[q] = 0.f;
    }
    const float c2 = p.scale_log2;
    // the two LDS read bases of this lane (everything else is an immediate)
    const char *const k_rd = smem + L::KS * l31 + 16 * h2;                 // K row l31, chunk h2
    const char *const v_rd = smem + L::V_BASE + L::VS * (4 * h2 + ((lane & 15) >> 2)) +
                             32 * ((lane >> 4) & 1) + 16 * ((lane & 3) >> 1) + 8 * (lane & 1);

    // bit q set: the 32 keys starting at KBASE need masking for query block q (wave-uniform)
    auto mask_bits = [&](int kbase) -> int {
        int m = 0;
#pragma unroll
        for (int q = 0; q < NQB; ++q)
            if ((CAUSAL && (kbase + 31 > wq0 + 32 * q + coff)) || (kbase + 32 > p.Sk)) m |= 1 << q;
        return m;
    };

    // ---- prologue: tile 0 into LDS, tile 1 in flight, scores of the first half-tile ----
    f32x16 sA[NQB], sB[NQB];
    float mxA[NQB], mxB[NQB];                   // lane-local maxima of the pending score half-tiles
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { sA[q][r] = 0.f; sB[q][r] = 0.f; }
        mxA[q] = ninf();
        mxB[q] = ninf();
    }
    // K runs TWO tiles ahead of the compute, V one: tile t lives in K buffer t % 3 and V buffer
    // t % 3.  Step t stores K(t+2) and V(t+1) (inside H1's PV slots), syncs once, and loads K(t+3)
    // and V(t+2) from global memory (inside H2's QK slots).  Because K(t+1) has been visible since
    // barrier(t-1), the first K fragments of H2(t) are read during the last slots of H1(t): nothing
    // right behind the barrier depends on what it publishes.
    // Buffer safety with ONE barrier per tile: K(t+2) overwrites K(t-1), last read in H1(t-1), and
    // V(t+1) overwrites V(t-2), last read in H2(t-2) -- both before barrier(t-1).  Loads/stores of
    // tiles past the end are row-clamped and land in buffers nobody reads again.
    // All prologue loads (Q, K(0), V(0), K(1)) are in flight together before anything waits.
    uint4 kx0, kx1, kx2, kx3;                   // K(1), prologue only
    kx0 = kx1 = kx2 = kx3 = make_uint4(0, 0, 0, 0);
    if (nt > 0) {
        SFA_LOAD_K(0);
        SFA_LOAD_V(0);
        SFA_LOAD_ONE(kx0, kx1, kx2, kx3, kg, k_rowb, k_tile_bytes, 1);
    }
    // Launder the Q fragments through an empty asm: hipcc waits for their global loads HERE and
    // afterwards no longer ties these registers to the VM counter (its loop-carried scoreboard
    // otherwise keeps a stale vmcnt(N) in front of every QK^T MFMA).
#pragma unroll
    for (int q = 0; q < NQB; ++q)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[q][ks]));

    if (nt > 0) {
        SFA_STORE_K(0);
        SFA_STORE_V(0);
        SFA_STORE_ONE(k_w, L::KS, L::KTILE, kx0, kx1, kx2, kx3);
    }
    __syncthreads();
    SFA_LOAD_K(2);
    SFA_LOAD_V(1);
    Vec kpre[PF];                               // first PF K fragments of the next half-step
#pragma unroll
    for (int i = 0; i < PF; ++i) kpre[i] = bitcast<Vec>(make_uint4(0, 0, 0, 0));
    if (ntw > 0) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const Vec a = bitcast<Vec>(*reinterpret_cast<const uint4 *>(k_rd + 32 * ks));
#pragma unroll
            for (int q = 0; q < NQB; ++q) sA[q] = Tr::mfma32(a, qf[q][ks], sA[q]);
        }
#pragma unroll
        for (int i = 0; i < PF; ++i)
            kpre[i] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(k_rd + L::KS * 32 + 32 * i));
#pragma unroll
        for (int q = 0; q < NQB; ++q) mxA[q] = lane_rowmax(sA[q]);
    }

    int kcur = 0, vcur = 0;         // byte offsets of tile t's K and V buffers
#define SFA_NEXT3(X, TILE) (((X) == 2 * (TILE)) ? 0 : (X) + (TILE))
    // non-overlapped form of the staging (TAIL and idle steps): store, sync, load
#define SFA_STAGE_AND_SYNC(T)                                                                       \
    do {                                                                                            \
        const int k1_ = SFA_NEXT3(kcur, L::KTILE);                                                  \
        SFA_STORE_K(SFA_NEXT3(k1_, L::KTILE));                                                      \
        SFA_STORE_V(SFA_NEXT3(vcur, L::VTILE));                                                     \
        __syncthreads();                                                                            \
        SFA_LOAD_K((T) + 3);                                                                        \
        SFA_LOAD_V((T) + 2);                                                                        \
        SFA_FENCE();                                                                                \
    } while (0)
#define SFA_ADVANCE()                                                                               \
    do {                                                                                            \
        kcur = SFA_NEXT3(kcur, L::KTILE);                                                           \
        vcur = SFA_NEXT3(vcur, L::VTILE);                                                           \
    } while (0)

    // FULL steps: this wave needs tile t+1 as well.  Staging is spread under the MFMAs: the ds_writes
    // ride in the PV slots of H1(t), the global loads in the QK slots of H2(t).
    constexpr int NPV_ = 2 * NDB;
    // DIAG & 2 (diagnostic build only, never the product): one workgroup stamps s_memtime at four
    // points of steps 8..15 into p.lse (as u64[wave][step][4]); the stamp drains lgkmcnt, so read
    // SHARES from it, not absolute speed (cdna_hip_programming.md section 7, In-kernel stamps).
    auto stamp = [&](int step, int which) {
        if ((DIAG & 2) && blockIdx.x == 8 && step >= 8 && step < 16 && p.lse) {
            unsigned long long tm;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) :: "memory");
            if (lane == 0)
                reinterpret_cast<unsigned long long *>(p.lse)[(wave * 8 + (step - 8)) * 4 + which] = tm;
        }
    };
    int t = 0;
    // H1(t): QK^T(B_t) || exp(A_t), PV(A_t); prefetches the first fragments of K(t+1)'s A half
#define SFA_H1_FULL()                                                                               \
    const int k1 = SFA_NEXT3(kcur, L::KTILE), k2 = SFA_NEXT3(k1, L::KTILE);                         \
    const int v1 = SFA_NEXT3(vcur, L::VTILE);                                                       \
    const char *kb = k_rd + kcur, *vb = v_rd + vcur, *kb1 = k_rd + k1;                              \
    auto st_hook = [&](int j) {         /* NOPS stores spread evenly over the NPV slots */           \
        _Pragma("unroll")                                                                           \
        for (int n = j * NOPS / NPV_; n < (j + 1) * NOPS / NPV_; ++n) store_op(n, k2, v1);          \
    };                                                                                              \
    h_block<Tr, D, NQB, PF, ORD, 1, 0, true, true>(kb, vb, kb1, qf, sB, sA, acc, c2, mxA, mxB,      \
                                              mask_bits(t * kBN), t * kBN, h2, lim, kpre, NoHook(), st_hook)
    // steady state: K(t+3) and V(t+2) are whole tiles, their loads ride in H2's QK slots
    const int t_fast_end = (DIAG & 32) ? 0 : min(ntw - 1, p.Sk / kBN - 3);    // DIAG & 32: loads never ride in slots
    for (; t < t_fast_end; ++t) {
        stamp(t, 0);
        SFA_H1_FULL();
        stamp(t, 1);
        __syncthreads();
        stamp(t, 2);
        const int tk = t + 3, tv = t + 2;
        auto ld_hook = [&](int i) {         // NOPS loads spread evenly over QK slots 1..NKS-1
#pragma unroll
            for (int n = (i - 1) * NOPS / (NKS - 1); n < i * NOPS / (NKS - 1); ++n) load_op(n, tk, tv);
        };
        // H2(t): QK^T(A_{t+1}) || exp(B_t), PV(B_t); prefetches K(t+1)'s B half for H1(t+1)
        h_block<Tr, D, NQB, PF, ORD, 0, 1, true, true>(kb1, vb, kb1, qf, sA, sB, acc, c2, mxB, mxA,
                                                  mask_bits(t * kBN + 32), t * kBN + 32, h2, lim, kpre, ld_hook);
        stamp(t, 3);
        SFA_ADVANCE();
    }
    // last FULL steps: the tiles to load are ragged or past the end -> clamped loads up front
    for (; t + 1 < ntw; ++t) {
        SFA_H1_FULL();
        __syncthreads();
        SFA_LOAD_K(t + 3);
        SFA_LOAD_V(t + 2);
        SFA_FENCE();
        h_block<Tr, D, NQB, PF, ORD, 0, 1, true, true>(kb1, vb, kb1, qf, sA, sB, acc, c2, mxB, mxA,
                                                  mask_bits(t * kBN + 32), t * kBN + 32, h2, lim, kpre);
        SFA_ADVANCE();
    }
#undef SFA_H1_FULL
    // TAIL step: this wave's last tile (no next scores to compute).
    if (t < ntw) {
        const char *kb = k_rd + kcur, *vb = v_rd + vcur;
        h_block<Tr, D, NQB, PF, ORD, 1, 0, true, false>(kb, vb, kb, qf, sB, sA, acc, c2, mxA, mxB,
                                                   mask_bits(t * kBN), t * kBN, h2, lim, kpre);
        SFA_STAGE_AND_SYNC(t);
        h_block<Tr, D, NQB, PF, ORD, 0, 1, false, false>(kb, vb, kb, qf, sA, sB, acc, c2, mxB, mxA,
                                                    mask_bits(t * kBN + 32), t * kBN + 32, h2, lim, kpre);
        SFA_ADVANCE();
        ++t;
    }
    // idle steps (causal: tiles beyond this wave's diagonal): keep staging for the other waves.
    for (; t < nt; ++t) {
        SFA_STAGE_AND_SYNC(t);
        SFA_ADVANCE();
    }
#undef SFA_NEXT3
#undef SFA_STAGE_AND_SYNC
#undef SFA_ADVANCE

    // ---- epilogue: normalise, convert, store O[row][:] (lane holds 4 consecutive d per group) ----
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
        const int qrow = wq0 + 32 * q + l31;
        const float ltot = half_sum(acc.lsum[q]);
        const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
        if (qrow < p.Sq) {
            uint16_t *op = p.o + b * p.os[0] + h * p.os[1] + (long long)qrow * p.os[2] + 4 * h2;
#pragma unroll
            for (int d = 0; d < NDB; ++d) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 w;
                    w.x = Tr::pack2(acc.o[q][d][4 * g + 0] * inv, acc.o[q][d][4 * g + 1] * inv);
                    w.y = Tr::pack2(acc.o[q][d][4 * g + 2] * inv, acc.o[q][d][4 * g + 3] * inv);
                    *reinterpret_cast<uint2 *>(op + 32 * d + 8 * g) = w;
                }
            }
            if (!(DIAG & 2) && p.lse && h2 == 0) {
                const float lse = ltot > 0.f ? (acc.msc[q] + __log2f(ltot)) * kLn2 : ninf();
                p.lse[((long long)b * p.Hq + h) * p.Sq + qrow] = lse;
            }
        }
    }
#undef SFA_LOAD_K
#undef SFA_LOAD_V
#undef SFA_LOAD_ONE
#undef SFA_STORE_K
#undef SFA_STORE_V
#undef SFA_STORE_ONE
#undef SFA_LD1
#undef SFA_LD1C
}

template <class Tr, int D, int NQB, int PF, int ORD, int DIAG>
int launch_t(const PrefillKernelParams &p, bool causal, hipStream_t stream) {
    const size_t lds = Lds<D>::TOTAL;      // K[3] + V[3], padded rows
    dim3 grid(8u * p.bh_per_xcd * p.nq_tiles), block(kThreads / NQB);
    static bool attr_set = false;       // idempotent; a race only repeats the call
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&prefill_kernel<Tr, D, true, NQB, PF, ORD, DIAG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&prefill_kernel<Tr, D, false, NQB, PF, ORD, DIAG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (causal) {
        hipLaunchKernelGGL((prefill_kernel<Tr, D, true, NQB, PF, ORD, DIAG>), grid, block, lds, stream, p);
    } else {
        hipLaunchKernelGGL((prefill_kernel<Tr, D, false, NQB, PF, ORD, DIAG>), grid, block, lds, stream, p);
    }
    return check_launch("prefill_kernel");
}

template <int NQB, int PF, int ORD, int DIAG>
int launch_cfg(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream) {
    if (dtype == SFA_DTYPE_FP16) {
        if (head_dim == 128) return launch_t<Fp16, 128, NQB, PF, ORD, DIAG>(p, causal, stream);
        if (head_dim == 64) return launch_t<Fp16, 64, NQB, PF, ORD, DIAG>(p, causal, stream);
    } else if (dtype == SFA_DTYPE_BF16) {
        if (head_dim == 128) return launch_t<Bf16, 128, NQB, PF, ORD, DIAG>(p, causal, stream);
        if (head_dim == 64) return launch_t<Bf16, 64, NQB, PF, ORD, DIAG>(p, causal, stream);
    } else {
        return fail(SFA_ERR_BAD_DTYPE, "sfa_prefill_fwd: dtype %d is not fp16(0)/bf16(1)", dtype);
    }
    return fail(SFA_ERR_UNSUPPORTED_HEAD_DIM, "sfa_prefill_fwd: head_dim %d not in {64, 128}", head_dim);
}

}  // namespace

int launch_prefill_main(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream) {
    return launch_cfg<1, 2, 2, 0>(p, dtype, head_dim, causal, stream);      // NQB = 1, prefetch distance 2, staged softmax
}
// diagnostic / A-B variants for tools/prefill_ab.py and tools/prefill_stamps.py (SFA_PREFILL_IMPL = 2..4)
int launch_prefill_variant(int which, const PrefillKernelParams &p, int dtype, int head_dim, bool causal,
                           hipStream_t stream) {
    if (which == 2) return launch_cfg<1, 2, 0, 0>(p, dtype, head_dim, causal, stream);     // un-staged softmax slices
    if (which == 3) return launch_cfg<1, 2, 2, 32>(p, dtype, head_dim, causal, stream);    // loads bunched after the barrier
    return launch_cfg<1, 2, 2, 2>(p, dtype, head_dim, causal, stream);                     // in-kernel stamps -> lse buffer
}

}  // namespace sfa
