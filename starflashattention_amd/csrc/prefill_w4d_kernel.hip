// Fused attention forward (prefill) for head_dim 256 on gfx950: the one-wave-per-SIMD, persistent structure of
// prefill_w4_kernel.hip (read its header first) re-cut for rows twice as long.  bf16 / fp16, causal or full, MHA or
// GQA, any strides, exact-scale numerics (fast_scale is accepted and ignored).  SURVEY.md 8(f-3): "hdim 64/256";
// the reference's head_dim is a runtime field (src/params.h:38).
//
// What changes against head_dim 128:
//   * O^T of ONE 32-row query block is already 8 blocks x 16 = 128 accumulator registers and Q^T 64: a wave owns 32
//     query rows, a workgroup (4 waves, one per SIMD) a 128-row q-tile.  Every K / V fragment read from LDS feeds one
//     MFMA (LDS reads per MFMA 1.5, against 0.75 -- the price of the long rows).
//   * K / V tiles are 32 keys (16 KiB in the same 8-row-group image, RG = 4096): the LDS budget is the same 160 KiB --
//     3-deep K and V rings (96 KiB) + a wave-private 32-row image of the next q-tile's Q rows (4 x 16 KiB).
//   * a tile is ONE step of 32 MFMA gaps, one barrier each:
//         step(t):  S(t+1) = K(t+1) Q^T   (gaps 0-15, K fragments read during step t-1)
//                   softmax stages of S(t) (one element per gap: 16 per lane and tile), V(t)^T reads (two per gap)
//                   O^T += V(t)^T P(t)^T  (gaps 16-31: k-step 0 x 8 d-blocks, k-step 1 x 8 d-blocks)
//                   row max of S(t+1) (gaps 17-24), lazy-rescale decision (25), its first stages (26-31)
//                   K(t+2) fragment reads (gaps 16-31), the LDS-DMA pieces of V(t+2), K(t+4) (gaps 16-23)
//     barrier(t) makes K(t+2) and V(t) visible; the slots of K(t+1) and V(t-1) are free by then.  A step is shorter than
//     an HBM round trip, so the producer runs K FOUR and V two stream positions ahead and a barrier waits only for the
//     pieces issued two steps back (s_waitcnt vmcnt(8)).
// Replaces the compiler-scheduled prefill_d256_kernel.hip (9 VALU per MFMA executed, MFMA busy 21 %).
#include "prefill_w4_common.h"

namespace sfa {

namespace {

using namespace prefill;

namespace w4d {

using namespace w4c;

constexpr int kD = 256;
constexpr int kRows = 128;          // query rows per workgroup (q-tile): 32 per wave
constexpr int kKeys = 32;           // keys per K/V tile = per step
constexpr int kLead = 5;            // elements of the next tile exponentiated at the end of the step that scored it (gaps 27-31)
constexpr int NKS = kD / 16, NDB = kD / 32, NJ = kD / 64;

template <int RING> struct Img {
    static constexpr int RG = 512 * (kD / 32);      // bytes of one 8-row group (4096)
    static constexpr int TILE = 4 * RG;             // 32 rows
    static constexpr int K_BASE = 0;
    static constexpr int V_BASE = RING * TILE;
    static constexpr int Q_BASE = 2 * RING * TILE;  // the Q rows of the next q-tile: one 32-row image per wave
    static constexpr int TOTAL = Q_BASE + 4 * TILE;
};

// Per-wave online-softmax state of its query block.
struct Acc {
    f32x16 o[NDB];              // O^T accumulators (AGPRs)
    float msc;                  // reference max the exponentials are taken against (log2 units)
    float msafe;                // msc, or 0 while a row has seen no key yet
    float thr;                  // msc + kThr: the lazy-rescale trigger
    float lsum;                 // this lane's share of the running row sum
    float alpha;                // a rescale of O decided but not yet applied; 1 = none
    uint32_t pk[8];             // P^T of the tile being consumed, packed: pk[4k .. 4k+3] = B operand of k-step k
};

__device__ __forceinline__ void rescale_o(Acc &acc, float alpha) {
#pragma unroll
    for (int d = 0; d < NDB; ++d) settle_acc(acc.o[d]);            // PV MFMAs of the previous step may be in flight
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.o[d][r] *= alpha;
}

// ---- the element stages (inline asm: they keep their place among the asm MFMAs) ----
template <int I>
__device__ __forceinline__ void st_f(f32x16 &s, float msafe, float c2) {
    if constexpr (I >= 0 && I < 16) asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(s[I]) : "s"(c2), "v"(msafe));
}
template <int I>
__device__ __forceinline__ void st_x(f32x16 &s) {
    if constexpr (I >= 0 && I < 16) asm volatile("v_exp_f32 %0, %0" : "+v"(s[I]));
}
template <class Tr, int I>
__device__ __forceinline__ void st_a(f32x16 &s, float &lsum, uint32_t (&pk)[8]) {
    if constexpr (I >= 0 && I < 16) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(lsum) : "v"(s[I]));
        if constexpr (I & 1) {
            if constexpr (Tr::id == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[I >> 1]) : "v"(s[I - 1]), "v"(s[I]));
            else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[I >> 1]) : "v"(s[I - 1]), "v"(s[I]));
        }
    }
}

// The reference max against freshly computed (masked if need be) raw scores: decide, and if it moves, move
// msc / thr / msafe / lsum now and park the factor for O (applied at the entry of the next step).
__device__ __forceinline__ void decide(Acc &acc, float mxl, float c2, int &pend) {
    if (__any(mxl * c2 > acc.thr)) {                            // rare after the first tiles
        const float mx = half_max(mxl) * c2;                    // both lane halves hold the same query
        const float mnew = fmaxf(acc.msc, mx);
        const float al = (mnew == ninf()) ? 1.0f : fast_exp2(acc.msc - mnew);
        acc.msc = mnew;
        acc.thr = mnew + kThr;
        acc.msafe = (mnew == ninf()) ? 0.f : mnew;
        acc.lsum *= al;
        acc.alpha = al;
        pend = 1;
    }
}
__device__ __forceinline__ void apply_pending(Acc &acc, int &pend) {
    if (pend) {
        rescale_o(acc, acc.alpha);
        acc.alpha = 1.0f;
        pend = 0;
    }
}

// First tile of a q-tile: its scores s were just computed outside the pipeline.  Sets the reference max outright
// and brings s into the entry state of step(): elements 0..kLead scaled, 0..kLead-1 exponentiated, 0..kLead-2 summed.
template <class Tr>
__device__ __forceinline__ void lead_in(Acc &acc, f32x16 &s, float c2, bool mask, int h2, int lim) {
    settle(s);
    if (mask) mask_keys(s, 0, h2, lim);
    const float mx = half_max(lane_rowmax(s));
    acc.msc = mx * c2;
    acc.thr = acc.msc + kThr;
    acc.msafe = (mx == ninf()) ? 0.f : acc.msc;
    static_for<kLead + 1>([&](auto ic) { st_f<decltype(ic)::value>(s, acc.msafe, c2); });
    static_for<kLead>([&](auto ic) { st_x<decltype(ic)::value>(s); });
    static_for<kLead - 1>([&](auto ic) { st_a<Tr, decltype(ic)::value>(s, acc.lsum, acc.pk); });
}

// One pipelined step of 32 gaps (see the header):
//   sN <- scores of the tile whose K fragments are in kf (read during the previous step)        (gaps 0-15)
//   sO  = scores of the tile whose V is at vbuf, in the entry state: finished, multiplied with V^T into acc.o,
//   and sN is left in the entry state for the next step.
// mask_n: sN holds keys that must be masked (diagonal / ragged tiles); kbase_n = their first key.
// kf in: the 16 K fragments of this step; out: those of the next one (the tile at kbuf_pref).
template <class Tr, class L, class Hook>
__device__ __forceinline__ void step(const lds_char *lds, unsigned k_e, unsigned v_e, int vbuf, int kbuf_pref,
                                     const typename Tr::mfma_vec (&qf)[NKS], f32x16 &sN, f32x16 &sO, Acc &acc, int &pend,
                                     float c2, bool mask_n, int kbase_n, int h2, int lim, typename Tr::mfma_vec (&kf)[NKS],
                                     const Hook &hook) {
    using Vec = typename Tr::mfma_vec;
    constexpr int RG = L::RG;
    const lds_char *const vb_0 = lds + (v_e + vbuf), *const vb_1 = lds + ((v_e ^ 32) + vbuf);
    const lds_char *const kp_e = lds + (k_e + kbuf_pref), *const kp_o = lds + ((k_e ^ 32) + kbuf_pref);
    auto ld_kp = [&](int ks) -> Vec { return bitcast<Vec>(lds_read16(((ks & 1) ? kp_o : kp_e) + 512 * (ks >> 1))); };
    // transposed read e (0 / 1) of V fragment j (A operand of the PV MFMA of d block j % 8, k-step j / 8)
    auto ld_vt = [&](int j, int e) -> u32x2 {
        const int d = j % NDB, s = j / NDB;
        return bitcast<u32x2>(__builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_i16x4 *)((e ? vb_1 : vb_0) + RG * (2 * s + e) + 512 * d)));
    };

    apply_pending(acc, pend);

    u32x2 vlo[2 * NDB], vhi[2 * NDB];
    asm volatile("" :: "v"(kf[NKS / 2 - 1]));       // one wait for the first eight K fragments (read >= 8 gaps ago) ...
    float mx;                               // lane max of the new scores
    static_for<32>([&](auto ic) {
        constexpr int n = decltype(ic)::value;
        // ---- the MFMA of this gap ----
        if constexpr (n < 16) {
            if constexpr (n == 8) asm volatile("" :: "v"(kf[NKS - 1]));        // ... and one for the other eight
            if constexpr (n == 0) mfma_qk_first<Tr>(sN, kf[0], qf[0]);
            else mfma_qk<Tr>(sN, kf[n], qf[n]);
        } else {
            constexpr int j = n - 16, ks = j >> 3;
            u32x4 av, pv;
            av[0] = vlo[j][0]; av[1] = vlo[j][1]; av[2] = vhi[j][0]; av[3] = vhi[j][1];
            pv[0] = acc.pk[4 * ks + 0]; pv[1] = acc.pk[4 * ks + 1];
            pv[2] = acc.pk[4 * ks + 2]; pv[3] = acc.pk[4 * ks + 3];
            mfma_pv<Tr, false>(acc.o[j & 7], bitcast<Vec>(av), bitcast<Vec>(pv));
        }
        // ---- LDS reads: both halves of V fragment n (used 16 gaps later), then the K fragments of the next step ----
        if constexpr (n < 16) {
            vlo[n] = ld_vt(n, 0);
            vhi[n] = ld_vt(n, 1);
        } else {
            kf[n - 16] = ld_kp(n - 16);          // (in place: fragment i was last used in gap i)
        }
        // one s_waitcnt per batch of V fragments instead of one per MFMA
        if constexpr (n == 15) asm volatile("" :: "v"(vhi[NDB - 1]));
        if constexpr (n == 23) asm volatile("" :: "v"(vhi[2 * NDB - 1]));
        // ---- stages of the tile being consumed: F one gap ahead of X, A one behind ----
        st_f<n + kLead + 1>(sO, acc.msafe, c2);
        st_x<n + kLead>(sO);
        st_a<Tr, n + kLead - 1>(sO, acc.lsum, acc.pk);
        // ---- row max of the new scores (complete behind gap 15), the decision, the first stages ----
        if constexpr (n == 17) {
            if (mask_n) mask_keys(sN, kbase_n, h2, lim);        // wave-uniform, diagonal / ragged tiles only
            st_max2(mx, sN[0], sN[1]);
        }
        if constexpr (n > 17 && n <= 24) st_max3(mx, sN[2 * (n - 17)], sN[2 * (n - 17) + 1]);
        if constexpr (n == 25) decide(acc, mx, c2, pend);
        if constexpr (n >= 26) st_f<n - 26>(sN, acc.msafe, c2);             // elements 0..kLead     (gaps 26-31)
        if constexpr (n >= 27) st_x<n - 27>(sN);                            // elements 0..kLead - 1 (gaps 27-31)
        if constexpr (n >= 28) st_a<Tr, n - 28>(sN, acc.lsum, acc.pk);      // elements 0..kLead - 2 (gaps 28-31)
        static_assert(kLead == 5, "the lead stages above fill gaps 26-31");
        hook(n);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Last step of a q-tile for this wave: no new scores.  Finishes sO (entry state) and adds its P.V.
template <class Tr, class L>
__device__ __forceinline__ void step_last(const lds_char *lds, unsigned v_e, int vbuf, f32x16 &sO, Acc &acc, int &pend, float c2) {
    using Vec = typename Tr::mfma_vec;
    constexpr int RG = L::RG;
    const lds_char *const vb_0 = lds + (v_e + vbuf), *const vb_1 = lds + ((v_e ^ 32) + vbuf);
    apply_pending(acc, pend);
    Vec vf[2 * NDB];
#pragma unroll
    for (int j = 0; j < 2 * NDB; ++j) {
        const int d = j % NDB, s = j / NDB;
        const u32x2 lo = bitcast<u32x2>(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(vb_0 + RG * (2 * s) + 512 * d)));
        const u32x2 hi = bitcast<u32x2>(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(vb_1 + RG * (2 * s + 1) + 512 * d)));
        u32x4 av;
        av[0] = lo[0]; av[1] = lo[1]; av[2] = hi[0]; av[3] = hi[1];
        vf[j] = bitcast<Vec>(av);
    }
    static_for<16>([&](auto ic) { if constexpr (decltype(ic)::value > kLead) st_f<decltype(ic)::value>(sO, acc.msafe, c2); });
    static_for<16>([&](auto ic) { if constexpr (decltype(ic)::value >= kLead) st_x<decltype(ic)::value>(sO); });
    static_for<16>([&](auto ic) { if constexpr (decltype(ic)::value >= kLead - 1) st_a<Tr, decltype(ic)::value>(sO, acc.lsum, acc.pk); });
    static_for<16>([&](auto ic) {
        constexpr int j = decltype(ic)::value, ks = j >> 3;
        u32x4 pv;
        pv[0] = acc.pk[4 * ks + 0]; pv[1] = acc.pk[4 * ks + 1];
        pv[2] = acc.pk[4 * ks + 2]; pv[3] = acc.pk[4 * ks + 3];
        mfma_pv<Tr, j == 0>(acc.o[j & 7], vf[j], bitcast<Vec>(pv));
    });
}

}  // namespace w4d

// Which items (q-tiles) a workgroup walks (the scheme of prefill_w4_kernel.hip): blockIdx & 7 labels the XCD, which
// owns heads [xcd * bh_per_xcd, +bh_per_xcd); its list is head-major, U units per head -- causal: unit i = the q-tile
// pair (nq-1-i, i); full: unit i = q-tile i -- and the XCD's workgroup `slot` takes units slot, slot + nslots, ...
struct W4dCursor {
    int hl, i, sub;     // head index inside the XCD's range, unit inside the head, 0 = the heavy q-tile of a causal pair
    int t, nt;          // tile inside the item, tiles of the item
    int b, h, qt;       // batch, head, q-tile
    int live;
};

template <class Tr, bool CAUSAL, int RING>
__global__ void __launch_bounds__(w4c::kThreadsW4, 1)
prefill_w4d_kernel(const PrefillKernelParams p) {
    using namespace w4d;
    using Vec = typename Tr::mfma_vec;
    using L = Img<RING>;
    constexpr int D = kD;
    static_assert(RING == 3, "ring depth");
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
    auto seek = [&](W4dCursor &c, bool skip_empty) {
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
    auto next_item = [&](W4dCursor &c, bool skip_empty) {
        if (CAUSAL && c.sub == 0) { c.sub = 1; }
        else {
            c.sub = 0;
            c.i += nslots;
            while (c.i >= U) { c.i -= U; ++c.hl; }
        }
        seek(c, skip_empty);
    };
    auto first_item = [&](W4dCursor &c, bool skip_empty) {
        c.hl = slot / U; c.i = slot % U; c.sub = 0; c.t = 0; c.nt = 0; c.b = 0; c.h = 0; c.qt = 0; c.live = false;
        seek(c, skip_empty);
    };

    // ---- LDS-DMA producers ----
    // Wave w stages rows [8w, 8w+8) of every tile = row group w: NJ (4) pieces of 8 rows x 128 B.  Lane -> (sub-tile
    // lane>>5, row (lane>>2)&7, slot lane&3) of its piece; the source chunk is slot ^ ((row>>2)&3) with row = 8w + r8,
    // so that LDS, written linearly, holds the swizzled image.
    const int r8 = (lane >> 2) & 7, dslot = lane & 3, dsub = lane >> 5;
    const unsigned k_rowb = (unsigned)(2 * p.ks[2]), v_rowb = (unsigned)(2 * p.vs[2]);
    const unsigned swz_w = 2u * (wave & 1) + (r8 >> 2);
    const unsigned kvoff = (unsigned)r8 * k_rowb + 64u * dsub + 16u * (dslot ^ swz_w);
    const unsigned vvoff = (unsigned)r8 * v_rowb + 64u * dsub + 16u * (dslot ^ swz_w);
    const lds_char *const lds = (const lds_char *)smem;
    const unsigned lds0 = (unsigned)(uintptr_t)lds;         // LDS byte address of the dynamic segment
    // One head's K (or V) rows form a buffer (launch_prefill_w4d keeps it below 2 GiB); the descriptor a wave uses
    // for a tile starts at ITS 8 rows of that tile and ends with the head, so rows past the sequence end read as zeros.
    struct Desc { unsigned lo, hi; int left; };
    const int k_extent = (p.Sk - 1) * (int)k_rowb + 2 * D, v_extent = (p.Sk - 1) * (int)v_rowb + 2 * D;
    const int k_tileb = kKeys * (int)k_rowb, v_tileb = kKeys * (int)v_rowb;
    const int G = p.Hq / p.Hkv;
    auto desc_at_head = [&](bool is_k, int b, int h) -> Desc {
        const int hk = h / G;
        const uint16_t *head = is_k ? p.k + b * p.ks[0] + hk * p.ks[1] : p.v + b * p.vs[0] + hk * p.vs[1];
        const unsigned skip = 8u * wave * (is_k ? k_rowb : v_rowb);
        const unsigned long long base = (unsigned long long)(uintptr_t)head + skip;
        return Desc{(unsigned)base, (unsigned)(base >> 32), (is_k ? k_extent : v_extent) - (int)skip};
    };
    auto desc_advance = [&](Desc &d, int tileb) {
        const unsigned lo = d.lo + (unsigned)tileb;
        d.hi += lo < d.lo ? 1u : 0u;
        d.lo = lo;
        d.left -= tileb;
    };
    // `live` false (the producer has run out of tiles): zero bytes, so the pieces still issue -- no branch in the
    // MFMA gaps -- and simply zero-fill a ring slot nobody will read
    auto make_srd = [&](const Desc &d, bool live) -> u32x4s {
        u32x4s srd;
        srd[0] = d.lo;
        srd[1] = d.hi & 0xffffu;
        srd[2] = live ? (unsigned)max(d.left, 0) : 0u;
        srd[3] = 0x00020000u;
        return srd;
    };
    auto issue_piece = [&](const u32x4s &srd, bool is_k, int ring_off, int j) {
        const unsigned dst = lds0 + (is_k ? L::K_BASE : L::V_BASE) + ring_off + wave * L::RG;
        dma_piece(dst + 1024 * j, is_k ? kvoff : vvoff, srd, 128u * j);
    };

    // ---- this lane's LDS read bases (the odd twins are ^32) ----
    const int kx = (l31 >> 2) & 3;
    const unsigned k_e = L::K_BASE + L::RG * (l31 >> 3) + 64 * (l31 & 7) + 16 * (h2 ^ kx);
    const int vy = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    const unsigned v_e = L::V_BASE + 64 * (4 * h2 + ((lane & 15) >> 2)) + 16 * (vy ^ h2) + 8 * (lane & 1);

    const float c2 = p.scale_log2;

    // ---- cursors: the compute, and the DMA producer running ahead of it.  A step is only 32 MFMAs long (~1.5k
    // cycles, less than an HBM round trip), so a piece gets TWO steps to land: behind barrier(t) the producer issues
    // V(t+2) and K(t+4), and barrier(t+1) waits only for what was issued behind barrier(t-1) (s_waitcnt vmcnt(8): the
    // eight pieces of the step in between may still be in flight -- every step issues exactly eight, an exhausted
    // producer issues them with empty descriptors).  V re-issues the tile K issued two calls earlier (vq). ----
    W4dCursor cc, pc;
    first_item(cc, false);
    first_item(pc, true);
    Desc kd = {0, 0, 0}, vd = {0, 0, 0}, vq0 = {0, 0, 0}, vq1 = {0, 0, 0};     // vq0: the older pending V tile
    int vq0_live = 0, vq1_live = 0;             // (ints: bools captured by the nested producer lambdas ended up in scratch)
    if (pc.live) { kd = desc_at_head(true, pc.b, pc.h); vd = desc_at_head(false, pc.b, pc.h); }
    int kring_p = 0, vring_p = 0;               // ring byte offsets the producers write next
    auto ring_next = [](int x) -> int { return x == (RING - 1) * L::TILE ? 0 : x + L::TILE; };
    u32x4s piece_srd = {0, 0, 0, 0};            // descriptor of the tile whose pieces are being dealt out
    auto produce_v_piece = [&](int j) {
        if (j == 0) piece_srd = make_srd(vq0, vq0_live != 0);
        issue_piece(piece_srd, false, vring_p, j);
        if (j == NJ - 1) {
            vq0.lo = vq1.lo; vq0.hi = vq1.hi; vq0.left = vq1.left;
            vq0_live = vq1_live;
            vq1_live = 0;
            vring_p = ring_next(vring_p);
        }
    };
    auto produce_k_piece = [&](int j) {
        if (j == 0) piece_srd = make_srd(kd, pc.live != 0);
        issue_piece(piece_srd, true, kring_p, j);
        if (j == NJ - 1) {
            // the V tile of this stream position joins the queue (the slot behind the one produce_v just popped)
            if (vq0_live) { vq1.lo = vd.lo; vq1.hi = vd.hi; vq1.left = vd.left; vq1_live = pc.live; }
            else { vq0.lo = vd.lo; vq0.hi = vd.hi; vq0.left = vd.left; vq0_live = pc.live; }
            if (pc.live) {
                if (++pc.t < pc.nt) {
                    desc_advance(kd, k_tileb);
                    desc_advance(vd, v_tileb);
                } else {
                    next_item(pc, true);
                    if (pc.live) { kd = desc_at_head(true, pc.b, pc.h); vd = desc_at_head(false, pc.b, pc.h); }
                }
            }
            kring_p = ring_next(kring_p);
        }
    };
    auto produce_v = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) produce_v_piece(j);
    };
    auto produce_k = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) produce_k_piece(j);
    };
    auto wait_and_sync = [&]() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(2 * NJ) : "memory"); };
    // The K fragments travel through the kernel as ONE stream: at the start of stream step u (a tile of some q-tile)
    // kf holds the fragments of K(u+1) -- read from LDS during step u-1 (by every wave, whether it computes on that tile
    // or idles behind its causal diagonal), which is what frees K(u+1)'s ring slot for K(u+4) at step u.  The first
    // tile of a q-tile is scored outside the pipeline from the fragments the previous q-tile's steps left in kf.
    Vec kf[NKS];
    auto read_kf = [&](int ring_off) {
        const lds_char *const ke = lds + (k_e + ring_off), *const ko = lds + ((k_e ^ 32) + ring_off);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) kf[ks] = bitcast<Vec>(lds_read16(((ks & 1) ? ko : ke) + 512 * (ks >> 1)));
    };
    // stream prologue: K(0) .. K(2), V(0), V(1) visible, K(0)'s fragments in kf, then K(3) into K(0)'s slot
    produce_k(); produce_k(); produce_v(); produce_k(); produce_v();
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    read_kf(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    produce_k();

    int kcur = 0, vcur = 0;                     // ring byte offsets of the compute's current tile
#define SFA_W4D_SYNC_AND_STAGE()                                                                    \
    do {                                                                                            \
        wait_and_sync();                                                                            \
        produce_v();                                                                                \
        produce_k();                                                                                \
    } while (0)

    // Q rows reach the accumulator file through LDS (see prefill_w4_kernel.hip): 16 LDS-DMA pieces into a wave-private
    // 32-row image in the K layout, requested behind the last barrier of the previous q-tile, read back like K fragments.
    Vec qf[NKS];
    const unsigned q_rowb = (unsigned)(2 * p.qs[2]);
    const unsigned qvoff0 = (unsigned)r8 * q_rowb + 64u * dsub + 16u * (dslot ^ (r8 >> 2));
    const unsigned qvoff1 = (unsigned)(r8 + 8) * q_rowb + 64u * dsub + 16u * (dslot ^ (2 + (r8 >> 2)));
    auto load_q = [&](int b, int h, int qt) {
        const int row0 = qt * kRows + 32 * wave;
        const unsigned long long base = (unsigned long long)(uintptr_t)(p.q + b * p.qs[0] + h * p.qs[1]) + (unsigned long long)row0 * q_rowb;
        u32x4s srd;
        srd[0] = (unsigned)base;
        srd[1] = (unsigned)(base >> 32) & 0xffffu;
        srd[2] = row0 < p.Sq ? (unsigned)(p.Sq - 1 - row0) * q_rowb + 2u * D : 0u;
        srd[3] = 0x00020000u;
        const unsigned dst = lds0 + L::Q_BASE + wave * L::TILE;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                dma_piece(dst + rg * L::RG + 1024 * j, (rg & 1) ? qvoff1 : qvoff0, srd, 128u * j + 16u * (rg >> 1) * q_rowb);
    };
    const unsigned q_e = L::Q_BASE + L::RG * (l31 >> 3) + 64 * (l31 & 7) + 16 * (h2 ^ kx);
    auto fetch_q = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the pieces (and whatever this wave stored since)
        const lds_char *const qe = lds + (q_e + wave * L::TILE), *const qo = lds + ((q_e ^ 32) + wave * L::TILE);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = bitcast<Vec>(lds_read16(((ks & 1) ? qo : qe) + 512 * (ks >> 1)));
    };

    if (cc.live) load_q(cc.b, cc.h, cc.qt);
    while (cc.live) {
        const int qt = cc.qt, nt = cc.nt;
        const int b = cc.b, h = cc.h;
        fetch_q();
        W4dCursor nx;                           // the item after this one (set where its Q rows are requested)
        // Q^T sits in the accumulator file (written from the LDS image just now); two wait states before the first MFMA
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) asm volatile("s_nop 1" : "+a"(qf[ks]));
        const int wq0 = qt * kRows + 32 * wave;                 // this wave's first query row
        int ntw = nt;                                           // tiles this wave computes on (wave-uniform)
        if (CAUSAL) ntw = (wq0 + 31 + coff >= 0) ? min(nt, (wq0 + 31 + coff) / kKeys + 1) : 0;
        const int qrow = wq0 + l31;
        const int lim = CAUSAL ? min(p.Sk - 1, qrow + coff) : p.Sk - 1;        // last visible key of this lane's row
        // the 32 keys starting at kbase need masking (wave-uniform) when kbase lies beyond the last tile the block sees whole
        const int whole = CAUSAL ? min(wq0 + coff - 31, p.Sk - 32) : p.Sk - 32;

        Acc acc;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc.o[d][r] = 0.f;
        acc.msc = ninf();
        acc.msafe = 0.f;
        acc.thr = ninf();
        acc.lsum = 0.f;
        acc.alpha = 1.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc.pk[i] = 0u;
        int pend = 0;                                           // a rescale of O is parked in acc.alpha (wave-uniform)

        // ---- scores of the first tile (outside the pipeline, from the fragments already in kf), the fragments of the second ----
        f32x16 sA, sB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sA[r] = 0.f; sB[r] = 0.f; }
        if (ntw > 0) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks == 0) mfma_qk_first<Tr>(sA, kf[0], qf[0]);
                else mfma_qk<Tr>(sA, kf[ks], qf[ks]);
            }
        }
        if (nt > 0) read_kf(ring_next(kcur));                   // the stream invariant for step 0: kf = K(1)
        if (ntw > 0) lead_in<Tr>(acc, sA, c2, 0 > whole, h2, lim);

        // ---- FULL steps: this wave needs the next tile as well.  sA / sB alternate as "being consumed" / "being scored":
        //   barrier(t)  -- K(t+2), V(t) visible; the slots of K(t+1), V(t-1) free: V(t+2), K(t+4) are issued in gaps 16-23
        //   step(t):    S(t+1) = K(t+1) Q^T || softmax(S(t)), O += P(t) V(t) || K(t+2) fragments -> registers
        auto dma_hook = [&](int n) {                            // one piece per gap of the (lighter) PV half: V pieces, then K pieces
            if (n >= 16 && n < 16 + NJ) produce_v_piece(n - 16);
            else if (n >= 16 + NJ && n < 16 + 2 * NJ) produce_k_piece(n - 16 - NJ);
        };
        int t = 0;
        // sA holds S(t) at every loop boundary; inside a pair of steps the roles alternate STATICALLY (a run-time
        // choice between (sA, sB) and (sB, sA) made hipcc spill 150 registers), and an odd step copies sB back.
#define SFA_W4D_STEP(SNEW, SCUR)                                                                                     \
        do {                                                                                                         \
            const int kb1 = (t + 1) * kKeys;                                                                         \
            const int kpref = ring_next(ring_next(kcur));                                                            \
            wait_and_sync();                                                                                         \
            step<Tr, L>(lds, k_e, v_e, vcur, kpref, qf, SNEW, SCUR, acc, pend, c2, kb1 > whole, kb1, h2, lim, kf, dma_hook);  \
            kcur = ring_next(kcur);                                                                                  \
            vcur = ring_next(vcur);                                                                                  \
            ++t;                                                                                                     \
        } while (0)
        while (t + 2 < ntw) {                                   // two full steps
            SFA_W4D_STEP(sB, sA);
            SFA_W4D_STEP(sA, sB);
        }
        if (t + 1 < ntw) {
            SFA_W4D_STEP(sB, sA);
            sA = sB;
        }
        // ---- LAST tile of this wave: no new scores ----
        if (t < ntw) {
            SFA_W4D_SYNC_AND_STAGE();
            if (t + 1 < nt) read_kf(ring_next(ring_next(kcur)));        // (not in the q-tile's last step: kf = the next q-tile's first tile)
            step_last<Tr, L>(lds, v_e, vcur, sA, acc, pend, c2);
            kcur = ring_next(kcur);
            vcur = ring_next(vcur);
            ++t;
        }
#undef SFA_W4D_STEP
        // ---- idle steps (causal: tiles beyond this wave's diagonal): keep staging for the others ----
        for (; t < nt; ++t) {
            SFA_W4D_SYNC_AND_STAGE();
            if (t + 1 < nt) read_kf(ring_next(ring_next(kcur)));
            kcur = ring_next(kcur);
            vcur = ring_next(vcur);
        }

        // the next q-tile's Q rows are requested here, behind this wave's last barrier of the q-tile
        nx = cc;
        next_item(nx, false);
        if (nx.live) load_q(nx.b, nx.h, nx.qt);
        // ---- epilogue: normalise, convert, store O[row][:] ----
#pragma unroll
        for (int d = 0; d < NDB; ++d) settle_acc(acc.o[d]);
        const float ltot = half_sum(acc.lsum);
        const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
        if (qrow < p.Sq) {
            uint16_t *orow = p.o + b * p.os[0] + h * p.os[1] + (long long)qrow * p.os[2];
            store_o_row<Tr, D>(orow, acc.o, inv, h2);
            if (p.lse && h2 == 0)
                p.lse[((long long)b * p.Hq + h) * p.Sq + qrow] = ltot > 0.f ? (acc.msc + __log2f(ltot)) * kLn2 : ninf();
        }
        cc = nx;
    }
    // the last steps' pieces must have landed before the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef SFA_W4D_SYNC_AND_STAGE
}

template <class Tr>
int launch_w4d_t(const PrefillKernelParams &p, bool causal, hipStream_t stream) {
    using namespace w4d;
    constexpr int RING = 3;
    const int lds = Img<RING>::TOTAL;
    // one workgroup per CU, fewer when the XCD lists are shorter than 32 units
    const int nq = (p.Sq + kRows - 1) / kRows;
    const long long units_xcd = (long long)p.bh_per_xcd * (causal ? (nq + 1) / 2 : nq);
    const int nslots = (int)(units_xcd < 32 ? units_xcd : 32);
    dim3 grid(8u * nslots), block(w4c::kThreadsW4);
    static DynLdsAttr attr_c, attr_f;
    if (const int rc = causal ? attr_c.ensure(reinterpret_cast<const void *>(&prefill_w4d_kernel<Tr, true, RING>), lds, "prefill_w4d_kernel")
                              : attr_f.ensure(reinterpret_cast<const void *>(&prefill_w4d_kernel<Tr, false, RING>), lds, "prefill_w4d_kernel"))
        return rc;
    if (causal) hipLaunchKernelGGL((prefill_w4d_kernel<Tr, true, RING>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((prefill_w4d_kernel<Tr, false, RING>), grid, block, lds, stream, p);
    return check_launch("prefill_w4d_kernel");
}

}  // namespace

// SFA_OK after a launch, SFA_ERR_UNSUPPORTED_HEAD_DIM-style refusals are the caller's (launch_prefill_d256 falls back
// to the compiler-scheduled kernel for shapes this one does not take: rows spanning more than 2 GiB per head).
int launch_prefill_w4d(const PrefillKernelParams &p, int dtype, bool causal, hipStream_t stream) {
    if (dtype != SFA_DTYPE_FP16 && dtype != SFA_DTYPE_BF16)
        return fail(SFA_ERR_BAD_DTYPE, "sfa_prefill_fwd: dtype %d is not fp16(0)/bf16(1)", dtype);
    return dtype == SFA_DTYPE_FP16 ? launch_w4d_t<Fp16>(p, causal, stream) : launch_w4d_t<Bf16>(p, causal, stream);
}

// the descriptors address one head's rows through 32-bit offsets
bool prefill_w4d_serves(const PrefillKernelParams &p) {
    const long long k_ext = (long long)(p.Sk - 1) * 2 * p.ks[2] + 512, v_ext = (long long)(p.Sk - 1) * 2 * p.vs[2] + 512;
    const long long q_ext = (long long)(p.Sq - 1) * 2 * p.qs[2] + 512;
    return k_ext < (1ll << 31) && v_ext < (1ll << 31) && q_ext < (1ll << 31) && p.ks[2] * 2 < (1ll << 24) &&
           p.vs[2] * 2 < (1ll << 24) && p.qs[2] * 2 < (1ll << 24);
}

}  // namespace sfa
