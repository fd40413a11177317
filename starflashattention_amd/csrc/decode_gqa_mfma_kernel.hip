// Grouped-query decode on the matrix cores, for groups too large for the VALU kernel
// (decode_gqa_kernel.hip is VALU-bound from 8 query heads per kv head on: 4.1 of 6.5 TB/s).
// One workgroup per (batch, kv head, split), four waves, each wave walks its share of the cached rows
// in 32-key tiles with the fragment maps of prefill_core16.h (v_mfma_f32_16x16x32, 16 query columns of
// which G are real):
//   S^T[key][q] = K . Q^T     A = K rows: head-major caches are read from HBM DIRECTLY in operand layout
//                                 (lane (c = l & 15, g = l >> 4) loads K[t + 16kt + c][32ks + 8g .. +8]);
//                                 the reference layout is read row-major and re-laid out through a
//                                 wave-private LDS tile (64-B pieces of strided rows cost 8 %);
//                             B = Q^T held in registers for the whole kernel (re-laid out once through LDS);
//   O^T[d][q] += V^T . P^T    P^T = the exponentiated S^T accumulators, packed (no data movement);
//                             V^T through a wave-private LDS tile: row-major ds_write_b128 in, ds_read_b64_tr_b16
//                             out (no barrier: a wave's LDS operations execute in order).
// The query sits on the lane in both accumulators: one online-softmax state per lane, row max across the
// four 16-lane groups by v_permlane32_swap + v_permlane16_swap.  Scores are scaled in fp32 (exact).
// The new token is one more tile of one valid key whose K/V come from the prologue's registers.
// Per 32 keys a wave issues 16 MFMAs and ~60 VALU instructions for 16 KB of cache: HBM-bound for any G <= 16.
#include <cstdlib>

#include "decode_common.h"
#include "prefill_core16.h"

namespace sfa {

namespace {

using namespace decode;
using prefill::Mfma16;
using prefill::quad_max;
using prefill::quad_sum;
using prefill::lds_i16x4;

constexpr int kTile = 32;                       // keys per tile

template <class Tr, int D, int G, bool NT, bool KLDS, bool PAGED = false>
__global__ void __launch_bounds__(kDecodeWaves * 64)
decode_gqa_mfma_kernel(const DecodeKernelParams p) {
    constexpr int W = kDecodeWaves;
    constexpr int LPR = D / 8;                  // lanes (16-byte chunks) per cache row
    constexpr int RPL = 64 / LPR;               // rows one load instruction of a wave covers
    constexpr int NLD = kTile / RPL;            // row-major loads per 32-row tile (= 2 NKS)
    constexpr int NKS = D / 32;                 // k-steps of a QK^T accumulator
    constexpr int NDT = D / 16;                 // 16-wide d tiles of O^T
    constexpr int VS = 2 * D + 32;              // LDS row stride of the V tile (conflict-free transposed reads)
    constexpr int VTILE = kTile * VS;
    constexpr int KS = 2 * D + 16;              // LDS row stride of the K tile (KLDS)
    constexpr int KTILE = kTile * KS;
    constexpr int WAVE_LDS = VTILE + KTILE;     // one tile of each: the next tile waits in registers
    using Vec = typename Tr::mfma_vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int hk = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;     // MFMA lane coordinates
    const int sub = lane % LPR, grp = lane / LPR;   // row-major coordinates: which 8 dims, which row of a load (prologue: which head)
    const int S = p.num_splits;
    const int Hq = p.H, Hkv = p.Hkv;

    const int pos = p.seq_len[b];
    const int reject = reject_code<PAGED>(p, b, pos);      // same contract as decode_kernel: poison, flag, touch nothing
    if (reject) {
        if (split == 0) {
            for (int i = tid; i < G * D; i += W * 64)
                p.o[((long long)b * Hq + (long long)hk * G) * D + i] = Tr::id == 0 ? 0x7e00 : 0x7fc0;
            if (tid == 0 && hk == 0) atomicOr(p.status, reject);
        }
        return;
    }

    // wave-private LDS: a V tile (also the Q / k_new re-layout area) and a K tile
    char *const vbuf = smem + wave * WAVE_LDS;
    char *const kbuf = vbuf + VTILE;

    // ---- prologue (every wave, simple layout: lane `sub` owns dims 8 sub .. +8; lane group g handles
    // query heads g, g + 4, ...): bias, RoPE (fp32), round to storage, park in LDS in [head][d] order ----
    const long long row0 = (long long)b * p.qkv_stride + sub * 8;
    float cs[4], sn[4];
    const int rot = p.rot_dim;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pj = sub * 4 + i;
        cs[i] = 1.f; sn[i] = 0.f;
        if (2 * pj < rot) {
            if (p.cos_tab) {
                const long long ti = (long long)pos * (rot >> 1) + pj;
                cs[i] = Tr::to_f32(p.cos_tab[ti]);
                sn[i] = Tr::to_f32(p.sin_tab[ti]);
            } else {                            // same fp32 recipe as decode_kernel.hip
                const float inv_freq = 1.0f / powf(10000.0f, (float)(2 * pj) / (float)rot);
                sincosf((float)pos * inv_freq, &sn[i], &cs[i]);
            }
        }
    }
    auto rope = [&](float (&x)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = x[2 * i], bb = x[2 * i + 1];
            x[2 * i] = a * cs[i] - bb * sn[i];
            x[2 * i + 1] = bb * cs[i] + a * sn[i];
        }
    };
    uint16_t *const qs = reinterpret_cast<uint16_t *>(vbuf);            // [16][D] query rows (rows >= G zero)
    uint16_t *const kn = qs + 16 * D;                                   // [D] the new token's key
    for (int q = grp; q < 16; q += RPL) {
        uint4 pk = make_uint4(0, 0, 0, 0);
        if (q < G) {
            float x[8];
            unpack8<Tr>(*reinterpret_cast<const uint4 *>(p.qkv + row0 + (long long)(hk * G + q) * D), x);
            if (p.q_bias) {
                float t[8];
                unpack8<Tr>(*reinterpret_cast<const uint4 *>(p.q_bias + (long long)(hk * G + q) * D + sub * 8), t);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] += t[j];
            }
            rope(x);
            pk = pack8<Tr>(x);
        }
        *reinterpret_cast<uint4 *>(qs + q * D + sub * 8) = pk;
    }
    uint4 kpk = make_uint4(0, 0, 0, 0), vpk = make_uint4(0, 0, 0, 0);
    {
        float xk[8], xv[8];
        unpack8<Tr>(*reinterpret_cast<const uint4 *>(p.qkv + row0 + (long long)(Hq + hk) * D), xk);
        const uint4 v_raw = *reinterpret_cast<const uint4 *>(p.qkv + row0 + (long long)(Hq + Hkv + hk) * D);
        vpk = v_raw;
        if (p.k_bias) {
            float t[8]; unpack8<Tr>(*reinterpret_cast<const uint4 *>(p.k_bias + (long long)hk * D + sub * 8), t);
#pragma unroll
            for (int j = 0; j < 8; ++j) xk[j] += t[j];
        }
        if (p.v_bias) {
            float t[8]; unpack8<Tr>(*reinterpret_cast<const uint4 *>(p.v_bias + (long long)hk * D + sub * 8), t);
            unpack8<Tr>(v_raw, xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] += t[j];
            vpk = pack8<Tr>(xv);
        }
        rope(xk);
        kpk = pack8<Tr>(xk);
        if (grp == 0) *reinterpret_cast<uint4 *>(kn + sub * 8) = kpk;
    }
    // Q^T fragments (B operand): lane holds Q[q = c][32 ks + 8 g .. +8]
    Vec qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = bitcast<Vec>(*reinterpret_cast<const uint4 *>(qs + c * D + 32 * ks + 8 * g));
    // K fragments of the new-token tile: key 0 of the tile = k_new (lanes c == 0), everything else masked
    uint4 knf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) knf[ks] = *reinterpret_cast<const uint4 *>(kn + 32 * ks + 8 * g);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the region is reused as a V tile below

    // ---- this wave's slice of the cached rows [0, pos) ----
    int rows_per_split = (pos + S - 1) / S;
    rows_per_split = (rows_per_split + kTile - 1) / kTile * kTile;     // paged: tiles never straddle 16-row halves
    const int r0 = min(pos, split * rows_per_split);
    const int r1 = min(pos, r0 + rows_per_split);
    int per_wave = (r1 - r0 + W - 1) / W;
    per_wave = (per_wave + kTile - 1) / kTile * kTile;
    const int w0 = __builtin_amdgcn_readfirstlane(min(r1, r0 + wave * per_wave));   // wave-uniform
    const int w1 = __builtin_amdgcn_readfirstlane(min(r1, w0 + per_wave));

    const long long rs = p.kv_row_stride;
    // paged (always with KLDS): split and wave boundaries are multiples of 32 rows, so the rows 0-15 and
    // 16-31 of a tile each lie in ONE page (page_size >= 16): two scalar table look-ups per tile and a
    // compile-time choice per load, no per-lane select.  Rows past the wave's end are clamped to its last
    // row; clamping the page INDEX the same way keeps their address on that row.
    const long long head_base = PAGED ? (long long)p.layer * (rs << p.page_shift) + (long long)hk * p.kv_head_stride
                                      : ((long long)b * p.L + p.layer) * p.M * Hkv * D + hk * p.kv_head_stride;
    const int32_t *tbl = PAGED ? p.block_table + (long long)b * p.table_stride : nullptr;
    const int pmask = PAGED ? (1 << p.page_shift) - 1 : 0;
    int bad_page = 0;
    auto page_of = [&](int idx) -> long long {
        int pg = tbl[idx];
        if ((unsigned)pg >= (unsigned)p.num_pages) {
            if (tid == 0) atomicOr(p.status, 2);
            bad_page = 1;       // a read page outside the pool: page 0 is read instead, the output becomes NaN
            pg = 0;
        }
        return pg * p.page_stride;
    };
    long long po[2] = {0, 0};                   // element offsets of the pages of the tile being loaded
    auto set_pages = [&](int t) {
        if (!PAGED) return;
        const int last = (w1 - 1) >> p.page_shift;
        po[0] = page_of(min(t >> p.page_shift, last));
        po[1] = page_of(min((t + 16) >> p.page_shift, last));
    };
    auto row_off = [&](int row, int half) -> long long {
        if (!PAGED) return (long long)row * rs;
        return po[half] + (long long)(row & pmask) * rs;
    };
    const uint16_t *const kb = p.k_cache + head_base + 8 * g;          // + row * rs + 32 ks: operand layout
    const uint16_t *const vb = p.v_cache + head_base + 8 * sub;            // + row * rs: row-major chunks

    f32x4 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[dt][r] = 0.f;
    float m = neg_inf(), l = 0.f;               // per lane: query c (replicated over the 4 lane groups)

    auto load_k = [&](uint4 (&kk)[2][NKS], int t) {
        set_pages(t);                           // load_v(.., t) follows and uses the same pages
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int row = min(t + 16 * kt + c, w1 - 1);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (KLDS) {     // row-major like V: load j = kt NKS + ks covers rows RPL j + grp, chunk sub
                    const int r2 = min(t + RPL * (kt * NKS + ks) + grp, w1 - 1);
                    kk[kt][ks] = ld16<NT>(p.k_cache + head_base + row_off(r2, kt) + 8 * sub);
                } else {        // directly in operand layout: 64-B pieces of 16 rows
                    kk[kt][ks] = ld16<NT>(kb + (long long)row * rs + 32 * ks);
                }
            }
        }
    };
    auto load_v = [&](uint4 (&vv)[NLD], int t) {        // lane: rows grp + RPL i, chunk sub
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int row = min(t + grp + RPL * i, w1 - 1);
            vv[i] = ld16<NT>(vb + row_off(row, (RPL * i) >> 4));
        }
    };
    auto store_v = [&](const uint4 (&vv)[NLD], char *buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            *reinterpret_cast<uint4 *>(buf + VS * (grp + RPL * i) + 16 * sub) = vv[i];
    };
    // KLDS: the K tile came in row-major; lay it out as MFMA operands through the wave's LDS K tile
    auto to_operand = [&](uint4 (&kk)[2][NKS]) {
        if (!KLDS) return;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                *reinterpret_cast<uint4 *>(kbuf + KS * (RPL * (kt * NKS + ks) + grp) + 16 * sub) = kk[kt][ks];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                kk[kt][ks] = *reinterpret_cast<const uint4 *>(kbuf + KS * (16 * kt + c) + 64 * ks + 16 * g);
    };
    // one 32-key tile: kk = K fragments, V tile at buf, keys [t, t + nvalid) are real
    auto tile = [&](const uint4 (&kk)[2][NKS], const char *buf, int nvalid) {
        f32x4 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) s[kt] = Mfma16<Tr>::run(bitcast<Vec>(kk[kt][ks]), qf[ks], s[kt]);
        }
        float mx = neg_inf();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {       // element r of tile kt = key 16 kt + 4 g + r
                s[kt][r] = (16 * kt + 4 * g + r < nvalid) ? s[kt][r] * p.scale_log2 : neg_inf();
                mx = fmaxf(mx, s[kt][r]);
            }
        mx = fmaxf(m, quad_max(mx));
        const float ms = (mx == neg_inf()) ? 0.f : mx;
        const float alpha = fast_exp2(m - ms);
        m = mx;
        l *= alpha;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
        }
        uint32_t pb[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const float p0 = fast_exp2(s[kt][0] - ms), p1 = fast_exp2(s[kt][1] - ms);
            const float p2 = fast_exp2(s[kt][2] - ms), p3 = fast_exp2(s[kt][3] - ms);
            l += (p0 + p1) + (p2 + p3);
            pb[2 * kt] = Tr::pack2(p0, p1);
            pb[2 * kt + 1] = Tr::pack2(p2, p3);
        }
        const Vec pv = bitcast<Vec>(make_uint4(pb[0], pb[1], pb[2], pb[3]));
        // V^T fragments: lane (c, g) reads rows 4 g + (c >> 2) and 16 + ..., 8 bytes at column 16 dt + 4 (c & 3)
        const char *vr = buf + VS * (4 * g + (c >> 2)) + 8 * (c & 3);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const auto t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(vr + 32 * dt));
            const auto t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(vr + VS * 16 + 32 * dt));
            u32x4 av;
            const u32x2 a_lo = bitcast<u32x2>(t0), a_hi = bitcast<u32x2>(t1);
            av[0] = a_lo[0]; av[1] = a_lo[1]; av[2] = a_hi[0]; av[3] = a_hi[1];
            o[dt] = Mfma16<Tr>::run(bitcast<Vec>(av), pv, o[dt]);
        }
    };

    if (w0 < w1) {
        // tile t+1 is in flight into registers while tile t is computed; the LDS tiles are single: a wave's
        // LDS operations execute in order, so storing tile t+1 cannot overtake the reads of tile t
        uint4 ka[2][NKS], kb2[2][NKS], vr[NLD];
        load_k(ka, w0);
        load_v(vr, w0);
        for (int t = w0; t < w1; t += 2 * kTile) {
            store_v(vr, vbuf);
            to_operand(ka);
            const bool more1 = t + kTile < w1;
            if (more1) { load_k(kb2, t + kTile); load_v(vr, t + kTile); }
            tile(ka, vbuf, w1 - t);
            if (more1) {
                store_v(vr, vbuf);
                to_operand(kb2);
                if (t + 2 * kTile < w1) { load_k(ka, t + 2 * kTile); load_v(vr, t + 2 * kTile); }
                tile(kb2, vbuf, w1 - t - kTile);
            }
        }
    }

    // ---- the new token (position `pos`): last split, wave 0 -- a tile with one real key ----
    if (split == S - 1 && wave == 0) {
        {   // every row of the V tile = v_new (rows 1.. get weight 0, but 0 * stale LDS bits could be NaN)
            uint4 vv[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i) vv[i] = vpk;
            store_v(vv, vbuf);
        }
        uint4 kk[2][NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            kk[0][ks] = knf[ks];                // only the lanes with c == 0 matter (key 0); the rest is masked
            kk[1][ks] = make_uint4(0, 0, 0, 0);
        }
        tile(kk, vbuf, 1);
        if (grp == 0) {                         // append: LPR lanes x 16 B = one row each
            long long roff = head_base + (long long)pos * rs + sub * 8;
            if (PAGED) roff = head_base + page_of(pos >> p.page_shift) + (long long)(pos & pmask) * rs + sub * 8;
            *reinterpret_cast<uint4 *>(p.k_cache + roff) = kpk;
            *reinterpret_cast<uint4 *>(p.v_cache + roff) = vpk;
        }
    }

    // ---- merge the workgroup's waves through LDS (after every wave is done with its V tiles) ----
    if (PAGED && bad_page) l = __builtin_nanf("");
    const float ltot = quad_sum(l);             // the four lane groups hold disjoint keys of query c
    __syncthreads();
    float *const red = reinterpret_cast<float *>(smem);                 // [W][G][D + 2]
    if (c < G) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * G + c) * (D + 2) + 16 * dt + 4 * g + r] = o[dt][r];
        if (g == 0) { red[(wave * G + c) * (D + 2) + D] = m; red[(wave * G + c) * (D + 2) + D + 1] = ltot; }
    }
    __syncthreads();
    for (int idx = tid; idx < LPR * G; idx += W * 64) {
        const int q = idx / LPR, sb = idx % LPR;
        Stream tot;
        tot.init();
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const float *rw = red + (w * G + q) * (D + 2);
            float a2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a2[j] = rw[sb * 8 + j];
            tot.merge(rw[D], rw[D + 1], a2);
        }
        const long long bh = (long long)b * Hq + hk * G + q;
        if (S == 1) {
            const float inv = 1.0f / tot.l;          // l >= 1: the new token is always present
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = tot.acc[j] * inv;
            *reinterpret_cast<uint4 *>(p.o + bh * D + sb * 8) = pack8<Tr>(y);
        } else {
            float *po = p.part_o + (bh * S + split) * D + sb * 8;
            *reinterpret_cast<float4 *>(po) = make_float4(tot.acc[0], tot.acc[1], tot.acc[2], tot.acc[3]);
            *reinterpret_cast<float4 *>(po + 4) = make_float4(tot.acc[4], tot.acc[5], tot.acc[6], tot.acc[7]);
            if (sb == 0) p.part_ml[bh * S + split] = make_float2(tot.m, tot.l);
        }
    }
}

template <class Tr, int D, int G, bool NT, bool KLDS, bool PAGED = false>
int launch_k(const DecodeKernelParams &p, hipStream_t stream) {
    dim3 grid(p.Hkv, p.num_splits, p.B), block(kDecodeWaves * 64);
    constexpr int lds = kDecodeWaves * kTile * ((2 * D + 32) + (2 * D + 16));       // 71,680 B at head_dim 128
    static_assert(lds >= kDecodeWaves * G * (D + 2) * 4, "merge area fits");
    static_assert(kTile * (2 * D + 32) >= 17 * D * 2, "the query rows and the new key fit the V tile they are re-laid out in");
    static DynLdsAttr attr;
    if (const int rc = attr.ensure(reinterpret_cast<const void *>(&decode_gqa_mfma_kernel<Tr, D, G, NT, KLDS, PAGED>), lds,
                                   "decode_gqa_mfma_kernel"))
        return rc;
    hipLaunchKernelGGL((decode_gqa_mfma_kernel<Tr, D, G, NT, KLDS, PAGED>), grid, block, lds, stream, p);
    return check_launch("decode_gqa_mfma_kernel");
}

template <class Tr, int D, int G>
int launch_g(const DecodeKernelParams &p, hipStream_t stream) {
    bool nt = 4ll * p.B * p.L * p.M * p.Hkv * D > (256ll << 20);        // see decode_kernel.hip
    if (const int k = g_knobs.decode_nt.load(std::memory_order_relaxed); k >= 0) nt = k != 0;      // tests, A/B
    // Reference layout: a K row of this head is a 256-B segment H*D*2 bytes from the next, and fetching it
    // as 64-B operand pieces costs 8 % (5.96 vs 6.44 TB/s): load row-major, re-lay out through LDS.
    // Head-major caches are contiguous, the operand-layout loads go straight to registers (6.7 TB/s).
    if (p.block_table)
        return nt ? launch_k<Tr, D, G, true, true, true>(p, stream) : launch_k<Tr, D, G, false, true, true>(p, stream);
    const bool klds = p.kv_row_stride != D;
    if (klds) return nt ? launch_k<Tr, D, G, true, true>(p, stream) : launch_k<Tr, D, G, false, true>(p, stream);
    return nt ? launch_k<Tr, D, G, true, false>(p, stream) : launch_k<Tr, D, G, false, false>(p, stream);
}

template <class Tr, int D>
int launch_d(const DecodeKernelParams &p, hipStream_t stream) {
    if (p.H == 16 * p.Hkv) return launch_g<Tr, D, 16>(p, stream);
    if (p.H == 4 * p.Hkv) return launch_g<Tr, D, 4>(p, stream);
    return launch_g<Tr, D, 8>(p, stream);
}

}  // namespace

// head_dim 64 / 128 / 256, any cache layout, 4, 8 or 16 query heads per kv head
int launch_decode_gqa_mfma(const DecodeKernelParams &p, int dtype, int head_dim, hipStream_t stream) {
    const bool h = dtype == SFA_DTYPE_FP16;
    if (head_dim == 64) return h ? launch_d<Fp16, 64>(p, stream) : launch_d<Bf16, 64>(p, stream);
    if (head_dim == 256) return h ? launch_d<Fp16, 256>(p, stream) : launch_d<Bf16, 256>(p, stream);
    return h ? launch_d<Fp16, 128>(p, stream) : launch_d<Bf16, 128>(p, stream);
}

}  // namespace sfa
