// Fused attention forward (prefill) for gfx950 (MI355X), the product kernel: half-tile software
// pipeline, explicitly slot-ordered, NQB query blocks per wave.
// bf16 / fp16, head_dim 64 / 128, causal or full, MHA or GQA.  The reference has no prefill
// kernel; this is the entry point BASELINE.json's headline metric is quoted on (SURVEY.md
// section 8(a) row A-new).
//
// MI355X design (MFMA-bound: AI = 1024 FLOP/B at S=4096, D=128):
//   * q-tile = 256 query rows of one (batch, head), 8 waves of 32 rows, two waves per SIMD.
//     (h_block in prefill_core.h is generic in NQB, query blocks per wave; NQB = 2 -- 4 waves x 64
//     rows, one per SIMD, every K / V fragment feeding two MFMAs -- halves the LDS operand traffic
//     but as scheduled by hipcc 7.2 measured 687 vs 868 TFLOPS even with
//     -mllvm -amdgpu-mfma-vgpr-form=1; it wants hand-placed AGPR ownership: DESIGN.md.)
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
//   * Two numeric flavours (template ORD): 2 = exact scale (scores are the fp32 QK^T, one FMA by
//     scale*log2(e) in front of v_exp; the default) and 6 = prescaled Q
//     (Q * scale*log2(e) rounded to 16 bit once per q-tile, the first QK^T MFMA of a half-step starts
//     from C = -reference max, so scores leave the MFMA as exp2 arguments and the scale/subtract pass
//     is gone: +5 %; opt-in, sfa_prefill_args.fast_scale).  launch_prefill_main picks.
//   * Where the K/V tiles of a stream position live is computed once per step on the scalar unit
//     (tile_src), not inside every staging load.
//   * LDS images use PADDED rows (prefill_common.h would XOR-swizzle): every read address is one
//     lane-constant base plus a compile-time immediate -- two LDS address registers in total --
//     and SQ_LDS_BANK_CONFLICT measures 0.
//   * One workgroup processes a PAIR of q-tiles of the same (batch, head): the heaviest remaining
//     one and the lightest (qt = n-1-i and i), so under the causal mask every workgroup does the same
//     4(n+1) tile steps.  The K/V tile stream simply continues from the first q-tile's tiles into
//     the second's (same K/V, indices restart), so the second q-tile has no staging prologue, no
//     dispatch gap, and its Q rows are loaded under the first one's last step and epilogue.
//     (Workgroup-level stamps: prologue 5.4 us + epilogue 2.2 us + ~3.7 us dispatch gap per workgroup
//     against ~75 us of tile steps at S=4096 causal -- tools/prefill_wg_stamps.py.)
//   * blockIdx -> (head, q-tile pair) is XCD-aware (prefill_common.h).
#include <cstdlib>

#include "prefill_core.h"

namespace sfa {

namespace {

using namespace prefill;

template <class Tr, int D, bool CAUSAL, int PF, int ORD, int DIAG>
__global__ void __launch_bounds__(kThreads, 2)
prefill_kernel(const PrefillKernelParams p) {
    using Vec = typename Tr::mfma_vec;
    constexpr int NQB = 1;
    constexpr bool PS = (ORD == 6);            // prescaled Q, scores leave the MFMA as exp2 arguments
    constexpr int PSO = PS ? 32 : 0;            // prescaled: a half-step finishes the max of its NEW scores
    constexpr int NKS = D / 16;                 // k-steps of Q.K^T
    constexpr int NDB = D / 32;                 // 32-wide d blocks of O^T
    constexpr int NPV_ = 2 * NDB;
    constexpr int CPR = D / 8;                  // 16-B chunks per row
    constexpr int NLD = kBN * CPR / kThreads;   // chunks staged per thread per tile (2 or 1)
    constexpr int ROWSTEP = kThreads / CPR;     // row distance between a thread's chunks
    using L = Lds<D>;
    static_assert(NLD >= 1 && NLD <= 2, "staging registers are named kr0, kr1");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- which (batch, head) and which q-tiles ----
    // n = ceil(Sq/256) q-tiles form ceil(n/2) balanced pairs (n-1-i, i); a workgroup owns
    // p.pairs_per_wg (1 or 2) of them: slot j of p.nq_tiles slots per head takes pairs j and
    // j + nq_tiles.  Items (q-tiles) are walked heavy, light, heavy, light.
    const BlockCoord bc = block_coord(p);       // .qt = slot index
    if (bc.bh >= p.B * p.Hq) return;
    const int nq = (p.Sq + kBM - 1) / kBM;
    const int npairs = (nq + 1) / 2;
    constexpr int MAX_ITEMS = 4;
    auto item_qt = [&](int it) -> int {         // q-tile of item `it`, or -1 if the item does not exist
        const int pr = bc.qt + (it >> 1) * p.nq_tiles;
        if ((it >> 1) >= p.pairs_per_wg || pr >= npairs) return -1;
        const int heavy = nq - 1 - pr;
        if (it & 1) return heavy == pr ? -1 : pr;
        return heavy;
    };
    // DIAG & 2 (diagnostic build only): workgroup-level stamps (start, loop entry of item 0, end of
    // item 0's loop, end) as u64 pairs (s_memtime, s_memrealtime) behind the per-step stamps in p.lse
    auto wg_stamp = [&](int which) {
        if ((DIAG & 2) && p.lse && blockIdx.x < 2048 && threadIdx.x == 0) {
            unsigned long long tm, rt;
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm), "=s"(rt) :: "memory");
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(p.lse) + 1024 + (blockIdx.x * 4 + which) * 2;
            dst[0] = tm;
            dst[1] = rt;
        }
    };
    wg_stamp(0);
    const int b = bc.bh / p.Hq, h = bc.bh % p.Hq;
    const int hk = h / (p.Hq / p.Hkv);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform
    const int l31 = lane & 31, h2 = lane >> 5;
    const int coff = p.Sk - p.Sq;               // causal: key j visible iff j <= i + coff

    // tiles of the K/V stream each item walks (workgroup-uniform)
    auto item_tiles = [&](int qt) -> int {
        int kv_end = p.Sk;
        if (CAUSAL) kv_end = min(p.Sk, qt * kBM + kBM + coff);
        return kv_end > 0 ? (kv_end + kBN - 1) / kBN : 0;
    };
    auto item_nt = [&](int it) -> int { const int qt = item_qt(it); return qt < 0 ? 0 : item_tiles(qt); };
    // stream position where each item's tiles start (cumulative)
    const int cs1 = item_nt(0), cs2 = cs1 + item_nt(1), cs3 = cs2 + item_nt(2);
    const int nt_all = cs3 + item_nt(3);         // length of the tile stream

    // ---- staging: thread owns chunks (row st_row + i*ROWSTEP, chunk st_ch), i < NLD, of every tile ----
    const int st_row = tid / CPR, st_ch = tid % CPR;
    const char *const kg = reinterpret_cast<const char *>(p.k + b * p.ks[0] + hk * p.ks[1]);   // uniform
    const char *const vg = reinterpret_cast<const char *>(p.v + b * p.vs[0] + hk * p.vs[1]);
    const long long k_tile_bytes = 2ll * kBN * p.ks[2], v_tile_bytes = 2ll * kBN * p.vs[2];
    const unsigned k_rowb = (unsigned)(2 * p.ks[2]), v_rowb = (unsigned)(2 * p.vs[2]);
    char *const k_w = smem + L::KS * st_row + 16 * st_ch;
    char *const v_w = smem + L::V_BASE + L::VS * st_row + 16 * st_ch;
    uint4 kr0, kr1, vr0, vr1;       // plain scalars: arrays of these ended up in scratch
    kr0 = kr1 = vr0 = vr1 = make_uint4(0, 0, 0, 0);

    // Branch-free staging loads.  Stream position -> K/V tile: the items' tiles back to back (indices
    // restart at 0 for every item, same K/V); positions past the end re-read the last tile of the sequence (never
    // used).  The one ragged tile (Sk % 64 != 0) swaps in a row-clamped lane offset with a v_cndmask.
    const int n_kv_tiles = (p.Sk + kBN - 1) / kBN;
    const int ragged_tile = (p.Sk % kBN) ? n_kv_tiles - 1 : -1;
    auto tile_of = [&](int pos) -> int {        // scalar; selects, no branches
        int base = pos >= cs1 ? cs1 : 0;
        base = pos >= cs2 ? cs2 : base;
        base = pos >= cs3 ? cs3 : base;
        return min(pos - base, n_kv_tiles - 1);
    };
    const int row0_ = st_row, row1_ = st_row + ROWSTEP;
    const int last0_ = p.Sk - 1 - (n_kv_tiles - 1) * kBN;                  // last valid row of the last tile
    const unsigned ow_k0 = (unsigned)row0_ * k_rowb + 16u * st_ch, ow_k1 = (unsigned)row1_ * k_rowb + 16u * st_ch;
    const unsigned ow_v0 = (unsigned)row0_ * v_rowb + 16u * st_ch, ow_v1 = (unsigned)row1_ * v_rowb + 16u * st_ch;
    const unsigned or_k0 = (unsigned)min(row0_, last0_) * k_rowb + 16u * st_ch;
    const unsigned or_k1 = (unsigned)min(row1_, last0_) * k_rowb + 16u * st_ch;
    const unsigned or_v0 = (unsigned)min(row0_, last0_) * v_rowb + 16u * st_ch;
    const unsigned or_v1 = (unsigned)min(row1_, last0_) * v_rowb + 16u * st_ch;
    constexpr int NOPS = 2 * NLD;   // op n: even = K chunk n/2, odd = V chunk n/2
    // Where the K and V tiles of two stream positions live: computed ONCE per step (scalar unit, a
    // dozen instructions), not inside every load -- per-load index math put 20 scalar branches and
    // ~100 SALU instructions per step into the MFMA slots.
    struct TileSrc { const char *k, *v; bool rk, rv; };
    auto tile_src = [&](int pos_k, int pos_v) -> TileSrc {
        const int tk = tile_of(pos_k), tv = tile_of(pos_v);
        return TileSrc{kg + tk * k_tile_bytes, vg + tv * v_tile_bytes, tk == ragged_tile, tv == ragged_tile};
    };
    auto load_op = [&](int n, const TileSrc &ts) {
        if (n == 0) kr0 = *reinterpret_cast<const uint4 *>(ts.k + (ts.rk ? or_k0 : ow_k0));
        if (n == 1) vr0 = *reinterpret_cast<const uint4 *>(ts.v + (ts.rv ? or_v0 : ow_v0));
        if (NLD > 1 && n == 2) kr1 = *reinterpret_cast<const uint4 *>(ts.k + (ts.rk ? or_k1 : ow_k1));
        if (NLD > 1 && n == 3) vr1 = *reinterpret_cast<const uint4 *>(ts.v + (ts.rv ? or_v1 : ow_v1));
    };
    auto store_op = [&](int n, int kbuf, int vbuf) {
        if (n == 0) *reinterpret_cast<uint4 *>(k_w + kbuf) = kr0;
        if (n == 1) *reinterpret_cast<uint4 *>(v_w + vbuf) = vr0;
        if (NLD > 1 && n == 2) *reinterpret_cast<uint4 *>(k_w + kbuf + ROWSTEP * L::KS) = kr1;
        if (NLD > 1 && n == 3) *reinterpret_cast<uint4 *>(v_w + vbuf + ROWSTEP * L::VS) = vr1;
    };

    const float c2 = p.scale_log2;
    // the two LDS read bases of this lane (everything else is an immediate)
    const char *const k_rd = smem + L::KS * l31 + 16 * h2;                 // K row l31, chunk h2
    const char *const v_rd = smem + L::V_BASE + L::VS * (4 * h2 + ((lane & 15) >> 2)) +
                             32 * ((lane >> 4) & 1) + 16 * ((lane & 3) >> 1) + 8 * (lane & 1);

    // ---- Q^T fragments (B operand): lane holds Q[row][16ks + 8*h2 .. +8] ----
    Vec qf[NQB][NKS];
    auto load_q = [&](int qt) {
        const int qrow = qt * kBM + 32 * wave + l31;
        const uint16_t *qp = p.q + b * p.qs[0] + h * p.qs[1] + (long long)min(qrow, p.Sq - 1) * p.qs[2] + 8 * h2;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            qf[0][ks] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(qp + 16 * ks));
    };
    // Launder the Q fragments through an empty asm: hipcc waits for their global loads HERE and
    // afterwards no longer ties these registers to the VM counter (its loop-carried scoreboard
    // otherwise keeps a stale vmcnt(N) in front of every QK^T MFMA).
    auto launder_q = [&]() {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[0][ks]));
    };
    auto prescale_q = [&]() {       // prescaled mode: fold scale * log2(e) into Q once per q-tile
        if (!PS) return;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 w = bitcast<u32x4>(qf[0][ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = Tr::pack2(Tr::lo_f32(w[i]) * c2, Tr::hi_f32(w[i]) * c2);
            qf[0][ks] = bitcast<Vec>(w);
        }
    };

    // ---- staging prologue: stream positions 0 and 1 into LDS, 2 (K) and 1 (V) in flight.
    // K runs TWO tiles ahead of the compute, V one: position t lives in K buffer t % 3 and V buffer
    // t % 3.  Step t stores K(t+2) and V(t+1) (inside H1's PV slots), syncs once, and loads K(t+3)
    // and V(t+2) (inside H2's QK slots).  Because K(t+1) has been visible since barrier(t-1), the
    // first K fragments of H2(t) are read during the last slots of H1(t): nothing right behind the
    // barrier depends on what it publishes.  Buffer safety with ONE barrier per tile: K(t+2)
    // overwrites K(t-1), last read in H1(t-1), and V(t+1) overwrites V(t-2), last read in H2(t-2)
    // -- both before barrier(t-1).
    load_q(item_qt(0));
    uint4 kx0, kx1;                             // K(1), prologue only
    kx0 = kx1 = make_uint4(0, 0, 0, 0);
    if (nt_all > 0) {
        const TileSrc ts0 = tile_src(0, 0);
#pragma unroll
        for (int n = 0; n < NOPS; ++n) load_op(n, ts0);
        const int t1 = tile_of(1);
        const bool r1 = t1 == ragged_tile;
        kx0 = *reinterpret_cast<const uint4 *>(kg + t1 * k_tile_bytes + (r1 ? or_k0 : ow_k0));
        if (NLD > 1) kx1 = *reinterpret_cast<const uint4 *>(kg + t1 * k_tile_bytes + (r1 ? or_k1 : ow_k1));
    }
    launder_q();
    if (nt_all > 0) {
#pragma unroll
        for (int n = 0; n < NOPS; ++n) store_op(n, 0, 0);
        *reinterpret_cast<uint4 *>(k_w + L::KTILE) = kx0;
        if (NLD > 1) *reinterpret_cast<uint4 *>(k_w + L::KTILE + ROWSTEP * L::KS) = kx1;
    }
    __syncthreads();
    {
        const TileSrc ts1 = tile_src(2, 1);
#pragma unroll
        for (int n = 0; n < NOPS; ++n) load_op(n, ts1);
    }

    int kcur = 0, vcur = 0;         // byte offsets of the K and V buffers of stream position t
    int t = 0;                      // stream position
#define SFA_NEXT3(X, TILE) (((X) == 2 * (TILE)) ? 0 : (X) + (TILE))
#define SFA_ADVANCE()                                                                               \
    do {                                                                                            \
        kcur = SFA_NEXT3(kcur, L::KTILE);                                                           \
        vcur = SFA_NEXT3(vcur, L::VTILE);                                                           \
    } while (0)
    // non-overlapped form of the staging (TAIL and idle steps): store, sync, load
#define SFA_STAGE_AND_SYNC(T)                                                                       \
    do {                                                                                            \
        const int k1_ = SFA_NEXT3(kcur, L::KTILE);                                                  \
        _Pragma("unroll") for (int n_ = 0; n_ < NOPS; ++n_)                                         \
            store_op(n_, SFA_NEXT3(k1_, L::KTILE), SFA_NEXT3(vcur, L::VTILE));                      \
        __syncthreads();                                                                            \
        const TileSrc ts_ = tile_src((T) + 3, (T) + 2);                                             \
        _Pragma("unroll") for (int n_ = 0; n_ < NOPS; ++n_) load_op(n_, ts_);                       \
        SFA_FENCE();                                                                                \
    } while (0)

    // DIAG & 2: one workgroup stamps s_memtime at four points of steps 8..15 into p.lse (as
    // u64[wave][step][4]); the stamp drains lgkmcnt, so read SHARES from it, not absolute speed.
    auto stamp = [&](int step, int which) {
        if ((DIAG & 2) && blockIdx.x == 8 && step >= 8 && step < 16 && p.lse) {
            unsigned long long tm;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) :: "memory");
            if (lane == 0)
                reinterpret_cast<unsigned long long *>(p.lse)[(wave * 8 + (step - 8)) * 4 + which] = tm;
        }
    };

    for (int item = 0; item < MAX_ITEMS; ++item) {
        const int qt = item_qt(item);
        if (qt < 0) continue;
        const int tbase = item == 0 ? 0 : item == 1 ? cs1 : item == 2 ? cs2 : cs3;    // stream position of tile 0
        const int nt = item_tiles(qt);                      // tiles the workgroup walks for this item
        int qt_next = -1;                                   // the next item that exists, if any
        for (int j = item + 1; j < MAX_ITEMS && qt_next < 0; ++j) qt_next = item_qt(j);
        const int q0 = qt * kBM;
        const int wq0 = q0 + 32 * wave;                     // this wave's first query row
        const int qrow = wq0 + l31;
        int ntw = nt;                                       // tiles this wave computes on (wave-uniform)
        if (CAUSAL) ntw = (wq0 + 31 + coff >= 0) ? min(nt, (wq0 + 31 + coff) / kBN + 1) : 0;
        int lim[NQB];                                       // last visible key of this lane's row
        lim[0] = CAUSAL ? min(p.Sk - 1, qrow + coff) : p.Sk - 1;
        // bit 0 set: the 32 keys starting at KBASE need masking for this wave's rows (wave-uniform)
        auto mask_bits = [&](int kbase) -> int {
            return ((CAUSAL && (kbase + 31 > wq0 + coff)) || (kbase + 32 > p.Sk)) ? 1 : 0;
        };
        const int tend = tbase + nt, twend = tbase + ntw;   // stream positions

        // Q of items > 0 was requested behind the previous item's TAIL step.  Unconditional (a no-op
        // asm for item 0): under `if (item > 0)` hipcc still saw the first-iteration path as "Q loads
        // in flight" and put a vmcnt wait in front of every QK^T MFMA of the main loop.
        launder_q();
        prescale_q();
        Acc<D, NQB> acc;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc.o[0][d][r] = 0.f;
        acc.msc[0] = PS ? 0.f : ninf();
        acc.lsum[0] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.cinit[0][r] = 0.f;

        // ---- scores of the first half-tile, first fragments of the second ----
        f32x16 sA[NQB], sB[NQB];
        float mxA[NQB] = {ninf()}, mxB[NQB] = {ninf()};     // lane-local maxima of the pending half-tiles
#pragma unroll
        for (int r = 0; r < 16; ++r) { sA[0][r] = 0.f; sB[0][r] = 0.f; }
        Vec kpre[PF];                                       // first PF K fragments of the next half-step
#pragma unroll
        for (int i = 0; i < PF; ++i) kpre[i] = bitcast<Vec>(make_uint4(0, 0, 0, 0));
        if (ntw > 0) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const Vec a = bitcast<Vec>(*reinterpret_cast<const uint4 *>(k_rd + kcur + 32 * ks));
                sA[0] = Tr::mfma32(a, qf[0][ks], sA[0]);
            }
#pragma unroll
            for (int i = 0; i < PF; ++i)
                kpre[i] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(k_rd + kcur + L::KS * 32 + 32 * i));
            mxA[0] = lane_rowmax(sA[0]);
            if (PS) {               // the first half-tile sets the reference outright (scores may sit far below 0)
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
        if (item == 0) wg_stamp(1);

        // ---- FULL steps: this wave needs the next tile as well.  Staging is spread under the MFMAs:
        // the ds_writes ride in the PV slots of H1, the global loads in the QK slots of H2.
        //   H1(t): QK^T(B_t)     || max,exp(A_t),   PV(A_t) || lane max(B_t)
        //   H2(t): QK^T(A_{t+1}) || max,exp(B_t),   PV(B_t) || lane max(A_{t+1})
        for (; t + 1 < twend; ++t) {
            stamp(t, 0);
            const int k1 = SFA_NEXT3(kcur, L::KTILE), k2 = SFA_NEXT3(k1, L::KTILE);
            const int v1 = SFA_NEXT3(vcur, L::VTILE);
            const char *kb = k_rd + kcur, *vb = v_rd + vcur, *kb1 = k_rd + k1;
            const int kbase = (t - tbase) * kBN;
            auto st_hook = [&](int j) {         // NOPS stores spread evenly over the NPV slots
#pragma unroll
                for (int n = j * NOPS / NPV_; n < (j + 1) * NOPS / NPV_; ++n) store_op(n, k2, v1);
            };
            // (mask, first key) describe the half-tile whose row max the half-step finishes: sO, or in
            // prescaled mode sN
            h_block<Tr, D, NQB, PF, ORD, 1, 0, true, true>(kb, vb, kb1, qf, sB, sA, acc, c2, mxA, mxB,
                                                      mask_bits(kbase + PSO), kbase + PSO, h2, lim, kpre, NoHook(), st_hook);
            stamp(t, 1);
            __syncthreads();
            stamp(t, 2);
            const TileSrc ts = tile_src(t + 3, t + 2);
            auto ld_hook = [&](int i) {         // NOPS loads spread evenly over QK slots 1..NKS-1
#pragma unroll
                for (int n = (i - 1) * NOPS / (NKS - 1); n < i * NOPS / (NKS - 1); ++n) load_op(n, ts);
            };
            h_block<Tr, D, NQB, PF, ORD, 0, 1, true, true>(kb1, vb, kb1, qf, sA, sB, acc, c2, mxB, mxA,
                                                      mask_bits(kbase + 32 + PSO), kbase + 32 + PSO, h2, lim, kpre, ld_hook);
            stamp(t, 3);
            SFA_ADVANCE();
        }
        // ---- TAIL step: this wave's last tile (its second half computes no new scores) ----
        // The Q fragments are dead after it: the next item's rows are requested right behind it, under
        // the idle steps and the epilogue.
        if (t < twend) {
            const char *kb = k_rd + kcur, *vb = v_rd + vcur;
            const int kbase = (t - tbase) * kBN;
            h_block<Tr, D, NQB, PF, ORD, 1, 0, true, false>(kb, vb, kb, qf, sB, sA, acc, c2, mxA, mxB,
                                                       mask_bits(kbase + PSO), kbase + PSO, h2, lim, kpre);
            SFA_STAGE_AND_SYNC(t);
            h_block<Tr, D, NQB, PF, ORD, 0, 1, false, false>(kb, vb, kb, qf, sA, sB, acc, c2, mxB, mxA,
                                                        mask_bits(kbase + 32), kbase + 32, h2, lim, kpre);
            SFA_ADVANCE();
            ++t;
        }
        if (qt_next >= 0) load_q(qt_next);
        // ---- idle steps (causal: tiles beyond this wave's diagonal): keep staging for the others ----
        for (; t < tend; ++t) {
            SFA_STAGE_AND_SYNC(t);
            SFA_ADVANCE();
        }
        if (item == 0) wg_stamp(2);

        // ---- epilogue: normalise, convert, store O[row][:] (lane holds 4 consecutive d per group) ----
        const float ltot = half_sum(acc.lsum[0]);
        const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
        if (qrow < p.Sq) {
            uint16_t *orow = p.o + b * p.os[0] + h * p.os[1] + (long long)qrow * p.os[2];
            store_o_row<Tr, D>(orow, acc.o[0], inv, h2);
            if (!(DIAG & 2) && p.lse && h2 == 0) {
                const float lse = ltot > 0.f ? (acc.msc[0] + __log2f(ltot)) * kLn2 : ninf();
                p.lse[((long long)b * p.Hq + h) * p.Sq + qrow] = lse;
            }
        }
    }
    wg_stamp(3);
#undef SFA_NEXT3
#undef SFA_ADVANCE
#undef SFA_STAGE_AND_SYNC
}

template <class Tr, int D, int PF, int ORD, int DIAG>
int launch_t(const PrefillKernelParams &p_in, bool causal, hipStream_t stream) {
    PrefillKernelParams p = p_in;
    const int nq = (p.Sq + kBM - 1) / kBM;
    const int npairs = (nq + 1) / 2;           // balanced q-tile pairs (n-1-i, i)
    // One pair per workgroup.  Two pairs (half the dispatches and staging prologues again) are
    // supported and tested, but measured 920 vs 941 TFLOPS on the headline shape, so they stay opt-in.
    p.pairs_per_wg = 1;
    if (const int v = g_knobs.prefill_pairs.load(std::memory_order_relaxed); v == 1 || v == 2)
        p.pairs_per_wg = v;                     // tests / A-B runs
    p.nq_tiles = (npairs + p.pairs_per_wg - 1) / p.pairs_per_wg;      // workgroup slots per head
    const size_t lds = Lds<D>::TOTAL;          // K[3] + V[3], padded rows
    dim3 grid(8u * p.bh_per_xcd * p.nq_tiles), block(kThreads);
    static DynLdsAttr attr_c, attr_f;
    if (const int rc = causal ? attr_c.ensure(reinterpret_cast<const void *>(&prefill_kernel<Tr, D, true, PF, ORD, DIAG>),
                                              (int)lds, "prefill_kernel")
                              : attr_f.ensure(reinterpret_cast<const void *>(&prefill_kernel<Tr, D, false, PF, ORD, DIAG>),
                                              (int)lds, "prefill_kernel"))
        return rc;
    if (causal) {
        hipLaunchKernelGGL((prefill_kernel<Tr, D, true, PF, ORD, DIAG>), grid, block, lds, stream, p);
    } else {
        hipLaunchKernelGGL((prefill_kernel<Tr, D, false, PF, ORD, DIAG>), grid, block, lds, stream, p);
    }
    return check_launch("prefill_kernel");
}

template <int PF, int ORD, int DIAG>
int launch_cfg(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream) {
    if (dtype == SFA_DTYPE_FP16) {
        if (head_dim == 128) return launch_t<Fp16, 128, PF, ORD, DIAG>(p, causal, stream);
        if (head_dim == 64) return launch_t<Fp16, 64, PF, ORD, DIAG>(p, causal, stream);
    } else if (dtype == SFA_DTYPE_BF16) {
        if (head_dim == 128) return launch_t<Bf16, 128, PF, ORD, DIAG>(p, causal, stream);
        if (head_dim == 64) return launch_t<Bf16, 64, PF, ORD, DIAG>(p, causal, stream);
    } else {
        return fail(SFA_ERR_BAD_DTYPE, "sfa_prefill_fwd: dtype %d is not fp16(0)/bf16(1)", dtype);
    }
    return fail(SFA_ERR_UNSUPPORTED_HEAD_DIM, "sfa_prefill_fwd: head_dim %d not in {64, 128}", head_dim);
}

}  // namespace

// Exact-scale kernel (ORD 2) by default; the prescaled-Q kernel (ORD 6, +5 %) only when the caller opted in
// (sfa_prefill_args.fast_scale) and wants no log-sum-exp: its scores carry Q*scale rounded to 16 bit --
// O within one 16-bit rounding of the exact kernel's for unit-variance data, but the score error grows
// with the logits (tests/test_prefill_gpu.py::test_prefill_extreme_logits).
int launch_prefill_main(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream) {
    if (p.fast_scale) return launch_cfg<2, 6, 0>(p, dtype, head_dim, causal, stream);
    return launch_cfg<2, 2, 0>(p, dtype, head_dim, causal, stream);     // prefetch distance 2, staged softmax
}
// forced / diagnostic variants for the tests, tools/prefill_ab.py and tools/prefill_*stamps.py
int launch_prefill_variant(int which, const PrefillKernelParams &p, int dtype, int head_dim, bool causal,
                           hipStream_t stream) {
#ifdef SFA_WITH_VARIANTS      // diagnostics: the A/B library only
    if (which == 2) return launch_cfg<2, 0, 0>(p, dtype, head_dim, causal, stream);     // un-staged softmax slices
    if (which == 4) return launch_cfg<2, 2, 2>(p, dtype, head_dim, causal, stream);     // in-kernel stamps -> lse buffer
#endif
    if (which == 3) return launch_cfg<2, 6, 0>(p, dtype, head_dim, causal, stream);     // prescaled Q, forced
    if (which == 10) return launch_cfg<2, 2, 0>(p, dtype, head_dim, causal, stream);    // exact scale, forced
    return fail(SFA_ERR_BAD_SHAPE, "prefill_impl %d needs the A/B build of the library (build_lib(variants=True))", which);
}

}  // namespace sfa
