// Fused attention forward (prefill) for gfx950 on v_mfma_f32_16x16x32: the structure of
// prefill_kernel.hip (paired 256-row q-tiles, 64-key tiles triple-buffered in padded LDS images,
// half-tile software pipeline in explicit slots, lazy rescale, staging spread under the MFMAs) with the
// fragment maps of prefill_core16.h.  Scores leave the MFMA already relative to the reference max
// (C operand = -max); EXACT keeps Q as given and multiplies by scale*log2(e) in front of v_exp, the
// prescaled flavour folds that factor into Q once per q-tile.
#include <cstdlib>

#include "prefill_core16.h"

namespace sfa {

namespace {

using namespace prefill;

template <class Tr, int D, bool CAUSAL, bool EXACT, int PF>
__global__ void __launch_bounds__(kThreads, 2)
prefill_kernel16(const PrefillKernelParams p) {
    using Vec = typename Tr::mfma_vec;
    constexpr int NKS = D / 32;                 // k-steps of one Q.K^T accumulator
    constexpr int NSL = 2 * NKS;                // QK slots of a half-step
    constexpr int NDT = D / 16;                 // 16-wide d tiles of O^T = PV slots of a half-step
    constexpr int NPV_ = NDT;
    constexpr int CPR = D / 8;                  // 16-B chunks per row
    constexpr int NLD = kBN * CPR / kThreads;   // chunks staged per thread per tile (2 or 1)
    constexpr int ROWSTEP = kThreads / CPR;     // row distance between a thread's chunks
    using L = Lds16<D>;
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
    const int b = bc.bh / p.Hq, h = bc.bh % p.Hq;
    const int hk = h / (p.Hq / p.Hkv);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform
    const int c16 = lane & 15, g = lane >> 4;
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
    const char *const k_rd = smem + L::KS * c16 + 16 * g;                  // K row c16, chunk g
    const char *const v_rd = smem + L::V_BASE + L::VS * (4 * g + (c16 >> 2)) + 8 * (lane & 3);

    // ---- Q^T fragments (B operand): lane holds Q[row 16qb + c16][32ks + 8g .. +8] ----
    Vec qf[2][NKS];
    auto load_q = [&](int qt) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qrow = qt * kBM + 32 * wave + 16 * qb + c16;
            const uint16_t *qp = p.q + b * p.qs[0] + h * p.qs[1] + (long long)min(qrow, p.Sq - 1) * p.qs[2] + 8 * g;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                qf[qb][ks] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(qp + 32 * ks));
        }
    };
    // (launder: see prefill_kernel.hip)
    auto launder_q = [&]() {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[qb][ks]));
    };
    auto prescale_q = [&]() {       // prescaled flavour: fold scale * log2(e) into Q once per q-tile
        if (EXACT) return;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                u32x4 w = bitcast<u32x4>(qf[qb][ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = Tr::pack2(Tr::lo_f32(w[i]) * c2, Tr::hi_f32(w[i]) * c2);
                qf[qb][ks] = bitcast<Vec>(w);
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

    for (int item = 0; item < MAX_ITEMS; ++item) {
        const int qt = item_qt(item);
        if (qt < 0) continue;
        const int tbase = item == 0 ? 0 : item == 1 ? cs1 : item == 2 ? cs2 : cs3;    // stream position of tile 0
        const int nt = item_tiles(qt);                      // tiles the workgroup walks for this item
        int qt_next = -1;                                   // the next item that exists, if any
        for (int j = item + 1; j < MAX_ITEMS && qt_next < 0; ++j) qt_next = item_qt(j);
        const int q0 = qt * kBM;
        const int wq0 = q0 + 32 * wave;                     // this wave's first query row
        int ntw = nt;                                       // tiles this wave computes on (wave-uniform)
        if (CAUSAL) ntw = (wq0 + 31 + coff >= 0) ? min(nt, (wq0 + 31 + coff) / kBN + 1) : 0;
        int lim[2];                                         // last visible key of this lane's two rows
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) lim[qb] = CAUSAL ? min(p.Sk - 1, wq0 + 16 * qb + c16 + coff) : p.Sk - 1;
        // bit 0 set: the 32 keys starting at KBASE need masking for this wave's rows (wave-uniform)
        auto mask_bits = [&](int kbase) -> int {
            return ((CAUSAL && (kbase + 31 > wq0 + coff)) || (kbase + 32 > p.Sk)) ? 1 : 0;
        };
        const int tend = tbase + nt, twend = tbase + ntw;   // stream positions

        launder_q();                                        // unconditional: see prefill_kernel.hip
        prescale_q();
        Acc16<D> acc;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc.o[qb][dt][r] = 0.f;
            acc.msc[qb] = 0.f;
            acc.lsum[qb] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.cinit[qb][r] = 0.f;
        }

        // ---- scores of the first half-tile, first fragments of the second ----
        f32x4 sA[2][2], sB[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { sA[qb][kt][r] = 0.f; sB[qb][kt][r] = 0.f; }
        Vec kpre[PF];                                       // first PF K fragments of the next half-step
#pragma unroll
        for (int i = 0; i < PF; ++i) kpre[i] = bitcast<Vec>(make_uint4(0, 0, 0, 0));
        if (ntw > 0) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const Vec a = bitcast<Vec>(*reinterpret_cast<const uint4 *>(k_rd + kcur + L::KS * 16 * kt + 64 * ks));
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) sA[qb][kt] = Mfma16<Tr>::run(a, qf[qb][ks], sA[qb][kt]);
                }
#pragma unroll
            for (int i = 0; i < PF; ++i)
                kpre[i] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(
                    k_rd + kcur + L::KS * (32 + 16 * (i & 1)) + 64 * (i >> 1)));
            // the first half-tile sets the reference outright (scores may sit far below 0)
            if (mask_bits(0)) {
                mask_half16(sA[0], 0, g, lim[0]);
                mask_half16(sA[1], 0, g, lim[1]);
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float mx = quad_max(lane_rowmax16(sA[qb]));
                const float m0 = (mx == ninf()) ? 0.f : mx;
                acc.msc[qb] = m0;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sA[qb][kt][r] -= m0;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc.cinit[qb][r] = -m0;
            }
        }

        // ---- FULL steps: this wave needs the next tile as well.  Staging is spread under the MFMAs:
        // the ds_writes ride in the PV slots of H1, the global loads in the QK slots of H2.
        //   H1(t): QK^T(B_t)     || exp(A_t),   PV(A_t) || lane max(B_t), finish max(B_t)
        //   H2(t): QK^T(A_{t+1}) || exp(B_t),   PV(B_t) || lane max(A_{t+1}), finish max(A_{t+1})
        for (; t + 1 < twend; ++t) {
            const int k1 = SFA_NEXT3(kcur, L::KTILE), k2 = SFA_NEXT3(k1, L::KTILE);
            const int v1 = SFA_NEXT3(vcur, L::VTILE);
            const char *kb = k_rd + kcur, *vb = v_rd + vcur, *kb1 = k_rd + k1;
            const int kbase = (t - tbase) * kBN;
            auto st_hook = [&](int j) {         // NOPS stores spread evenly over the PV slots
#pragma unroll
                for (int n = j * NOPS / NPV_; n < (j + 1) * NOPS / NPV_; ++n) store_op(n, k2, v1);
            };
            h_block16<Tr, D, EXACT, PF, 1, 0, true, true>(kb, vb, kb1, qf, sB, sA, acc, c2, mask_bits(kbase + 32),
                                                          kbase + 32, g, lim, kpre, NoHook(), st_hook);
            __syncthreads();
            const TileSrc ts = tile_src(t + 3, t + 2);
            auto ld_hook = [&](int i) {         // NOPS loads spread evenly over QK slots 1..NSL-1
#pragma unroll
                for (int n = (i - 1) * NOPS / (NSL - 1); n < i * NOPS / (NSL - 1); ++n) load_op(n, ts);
            };
            h_block16<Tr, D, EXACT, PF, 0, 1, true, true>(kb1, vb, kb1, qf, sA, sB, acc, c2, mask_bits(kbase + 64),
                                                          kbase + 64, g, lim, kpre, ld_hook);
            SFA_ADVANCE();
        }
        // ---- TAIL step: this wave's last tile (its second half computes no new scores) ----
        // The Q fragments are dead after it: the next item's rows are requested right behind it, under
        // the idle steps and the epilogue.
        if (t < twend) {
            const char *kb = k_rd + kcur, *vb = v_rd + vcur;
            const int kbase = (t - tbase) * kBN;
            h_block16<Tr, D, EXACT, PF, 1, 0, true, false>(kb, vb, kb, qf, sB, sA, acc, c2, mask_bits(kbase + 32),
                                                           kbase + 32, g, lim, kpre);
            SFA_STAGE_AND_SYNC(t);
            h_block16<Tr, D, EXACT, PF, 0, 1, false, false>(kb, vb, kb, qf, sA, sB, acc, c2, 0, 0, g, lim, kpre);
            SFA_ADVANCE();
            ++t;
        }
        if (qt_next >= 0) load_q(qt_next);
        // ---- idle steps (causal: tiles beyond this wave's diagonal): keep staging for the others ----
        for (; t < tend; ++t) {
            SFA_STAGE_AND_SYNC(t);
            SFA_ADVANCE();
        }

        // ---- epilogue: normalise, convert, store.  Lane holds d 16dt + 4g .. +4 of rows 16qb + c16 ----
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float ltot = quad_sum(acc.lsum[qb]);
            const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
            const int qrow = wq0 + 16 * qb + c16;
            if (qrow < p.Sq) {
                uint16_t *orow = p.o + b * p.os[0] + h * p.os[1] + (long long)qrow * p.os[2] + 4 * g;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    uint2 w;
                    w.x = Tr::pack2(acc.o[qb][dt][0] * inv, acc.o[qb][dt][1] * inv);
                    w.y = Tr::pack2(acc.o[qb][dt][2] * inv, acc.o[qb][dt][3] * inv);
                    *reinterpret_cast<uint2 *>(orow + 16 * dt) = w;
                }
                if (p.lse && g == 0) {
                    const float mlog2 = EXACT ? acc.msc[qb] * c2 : acc.msc[qb];
                    const float lse = ltot > 0.f ? (mlog2 + __log2f(ltot)) * kLn2 : ninf();
                    p.lse[((long long)b * p.Hq + h) * p.Sq + qrow] = lse;
                }
            }
        }
    }
#undef SFA_NEXT3
#undef SFA_ADVANCE
#undef SFA_STAGE_AND_SYNC
}


template <class Tr, int D, bool EXACT>
int launch_t(const PrefillKernelParams &p_in, bool causal, hipStream_t stream) {
    PrefillKernelParams p = p_in;
    const int nq = (p.Sq + kBM - 1) / kBM;
    const int npairs = (nq + 1) / 2;           // balanced q-tile pairs (n-1-i, i)
    p.pairs_per_wg = 1;
    if (const int v = g_knobs.prefill_pairs.load(std::memory_order_relaxed); v == 1 || v == 2)
        p.pairs_per_wg = v;                     // tests / A-B runs
    p.nq_tiles = (npairs + p.pairs_per_wg - 1) / p.pairs_per_wg;      // workgroup slots per head
    const size_t lds = Lds16<D>::TOTAL;
    dim3 grid(8u * p.bh_per_xcd * p.nq_tiles), block(kThreads);
    static DynLdsAttr attr_c, attr_f;
    if (const int rc = causal ? attr_c.ensure(reinterpret_cast<const void *>(&prefill_kernel16<Tr, D, true, EXACT, 2>),
                                              (int)lds, "prefill_kernel16")
                              : attr_f.ensure(reinterpret_cast<const void *>(&prefill_kernel16<Tr, D, false, EXACT, 2>),
                                              (int)lds, "prefill_kernel16"))
        return rc;
    if (causal) {
        hipLaunchKernelGGL((prefill_kernel16<Tr, D, true, EXACT, 2>), grid, block, lds, stream, p);
    } else {
        hipLaunchKernelGGL((prefill_kernel16<Tr, D, false, EXACT, 2>), grid, block, lds, stream, p);
    }
    return check_launch("prefill_kernel16");
}

template <bool EXACT>
int launch_flavour(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream) {
    if (dtype == SFA_DTYPE_FP16) {
        if (head_dim == 128) return launch_t<Fp16, 128, EXACT>(p, causal, stream);
        if (head_dim == 64) return launch_t<Fp16, 64, EXACT>(p, causal, stream);
    } else if (dtype == SFA_DTYPE_BF16) {
        if (head_dim == 128) return launch_t<Bf16, 128, EXACT>(p, causal, stream);
        if (head_dim == 64) return launch_t<Bf16, 64, EXACT>(p, causal, stream);
    } else {
        return fail(SFA_ERR_BAD_DTYPE, "sfa_prefill_fwd: dtype %d is not fp16(0)/bf16(1)", dtype);
    }
    return fail(SFA_ERR_UNSUPPORTED_HEAD_DIM, "sfa_prefill_fwd: head_dim %d not in {64, 128}", head_dim);
}

}  // namespace

// force: 0 = by policy (exact scale unless the caller opted into the prescaled-Q flavour),
//        1 = prescaled, 2 = exact
int launch_prefill_x16(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream,
                       int force) {
    const bool exact = force == 0 ? p.fast_scale == 0 : force == 2;
    if (exact) return launch_flavour<true>(p, dtype, head_dim, causal, stream);
    return launch_flavour<false>(p, dtype, head_dim, causal, stream);
}

}  // namespace sfa
