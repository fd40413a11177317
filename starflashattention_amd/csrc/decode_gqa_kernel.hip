// Single-token decode attention with grouped queries (num_heads = G * num_heads_kv) for gfx950:
// the structure of decode_kernel.hip -- D/8 lanes x 16 B per cache row, K/V straight to VGPRs,
// register double-buffered, one online-softmax stream per lane group -- with ONE workgroup per
// (batch, kv head, split) serving the G query heads of the group from the same K/V registers, so the
// cache is read once, not G times (the grouped-query point: G x less HBM traffic per output head).
//
// The reference has no GQA (its structs carry no kv-head count): this is an additive
// C-ABI extension (sfa_decode_args.num_heads_kv), packed input qkv [B, Hq + 2*Hkv, D] (q heads, then
// k heads, then v heads -- with Hkv == Hq exactly the reference's [B, 3, H, D]), caches with Hkv heads.
//
// Per cached row and lane the work is G x (4 v_dot2 + 4 DPP adds + softmax update + 8 FMA): at G = 4
// the VALU is about as busy as HBM, at G = 8 the kernel is VALU-bound -- still far cheaper than
// streaming the cache 8 times.  K/V loads use the non-temporal policy under the same rule as
// decode_kernel.hip.
#include <cstdlib>

#include "decode_common.h"

namespace sfa {

namespace {

using namespace decode;

template <class Tr, int D, int G, bool NT, bool PAGED = false>
__global__ void __launch_bounds__(kDecodeWaves * 64)
decode_gqa_kernel(const DecodeKernelParams p) {
    constexpr int W = kDecodeWaves;
    constexpr int LPR = D / 8;          // lanes per cache row
    constexpr int GR = 64 / LPR;        // cache rows per wave-instruction
    constexpr int U0 = G >= 4 ? 2 : 4;  // row groups per step (registers: G streams of 10 floats each)
    constexpr int U = (PAGED && GR * U0 > 16) ? 16 / GR : U0;   // paged: a step (<= 16 rows) spans <= 2 pages
    constexpr int STEP = GR * U;
    const int hk = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sub = lane % LPR, grp = lane / LPR;
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

    // ---- G query heads, k_new, v_new for this lane's 8 dims: bias, RoPE (fp32), round to storage ----
    const long long row0 = (long long)b * p.qkv_stride + sub * 8;
    float xq[G][8], xk[8], xv[8];
#pragma unroll
    for (int g = 0; g < G; ++g)
        unpack8<Tr>(*reinterpret_cast<const uint4 *>(p.qkv + row0 + (long long)(hk * G + g) * D), xq[g]);
    unpack8<Tr>(*reinterpret_cast<const uint4 *>(p.qkv + row0 + (long long)(Hq + hk) * D), xk);
    const uint4 v_raw = *reinterpret_cast<const uint4 *>(p.qkv + row0 + (long long)(Hq + Hkv + hk) * D);
    uint4 vpk = v_raw;
    if (p.q_bias) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float t[8];
            unpack8<Tr>(*reinterpret_cast<const uint4 *>(p.q_bias + (long long)(hk * G + g) * D + sub * 8), t);
#pragma unroll
            for (int j = 0; j < 8; ++j) xq[g][j] += t[j];
        }
    }
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
    const int rot = p.rot_dim;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pj = sub * 4 + i;                 // pair index: dims (2pj, 2pj+1)
        if (2 * pj < rot) {
            float c, s;
            if (p.cos_tab) {
                const long long ti = (long long)pos * (rot >> 1) + pj;
                c = Tr::to_f32(p.cos_tab[ti]);
                s = Tr::to_f32(p.sin_tab[ti]);
            } else {                                // same fp32 recipe as decode_kernel.hip
                const float inv_freq = 1.0f / powf(10000.0f, (float)(2 * pj) / (float)rot);
                const float ang = (float)pos * inv_freq;
                sincosf(ang, &s, &c);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float q0 = xq[g][2 * i], q1 = xq[g][2 * i + 1];
                xq[g][2 * i] = q0 * c - q1 * s;
                xq[g][2 * i + 1] = q1 * c + q0 * s;
            }
            const float k0 = xk[2 * i], k1 = xk[2 * i + 1];
            xk[2 * i] = k0 * c - k1 * s;
            xk[2 * i + 1] = k1 * c + k0 * s;
        }
    }
    uint4 qpk[G];
#pragma unroll
    for (int g = 0; g < G; ++g) qpk[g] = pack8<Tr>(xq[g]);
    const uint4 kpk = pack8<Tr>(xk);

    // ---- this wave's slice of the cached rows [0, pos) ----
    const int rows_per_split = (pos + S - 1) / S;
    const int r0 = min(pos, split * rows_per_split);
    const int r1 = min(pos, r0 + rows_per_split);
    int per_wave = (r1 - r0 + W - 1) / W;
    per_wave = (per_wave + STEP - 1) / STEP * STEP;
    const int w0 = __builtin_amdgcn_readfirstlane(min(r1, r0 + wave * per_wave));   // wave-uniform
    const int w1 = __builtin_amdgcn_readfirstlane(min(r1, w0 + per_wave));

    const long long rs = p.kv_row_stride;
    // paged: see decode_kernel.hip (two scalar table look-ups per step, selected per lane)
    const long long head_base = PAGED ? (long long)p.layer * (rs << p.page_shift) + (long long)hk * p.kv_head_stride + sub * 8
                                      : ((long long)b * p.L + p.layer) * p.M * Hkv * D + hk * p.kv_head_stride + sub * 8;
    const uint16_t *kb = p.k_cache + head_base;
    const uint16_t *vb = p.v_cache + head_base;
    const int32_t *tbl = PAGED ? p.block_table + (long long)b * p.table_stride : nullptr;
    const int pmask = PAGED ? (1 << p.page_shift) - 1 : 0;
    int bad_page = 0;
    auto page_of = [&](int idx) -> int {
        int pg = tbl[idx];
        if ((unsigned)pg >= (unsigned)p.num_pages) {
            if (tid == 0) atomicOr(p.status, 2);
            bad_page = 1;       // a read page outside the pool: page 0 is read instead, the output becomes NaN
            pg = 0;
        }
        return pg;
    };
    auto row_off = [&](int row, int i0, long long o0, long long o1) -> long long {
        if (!PAGED) return (long long)row * rs;
        return ((row >> p.page_shift) == i0 ? o0 : o1) + (long long)(row & pmask) * rs;
    };

    Stream st[G];
#pragma unroll
    for (int g = 0; g < G; ++g) st[g].init();

    auto load = [&](uint4 (&kk)[U], uint4 (&vv)[U], int t) {
        int i0 = 0;
        long long o0 = 0, o1 = 0;
        if (PAGED) {
            i0 = t >> p.page_shift;
            const int i1 = min(i0 + 1, (w1 - 1) >> p.page_shift);
            o0 = page_of(i0) * p.page_stride;
            o1 = page_of(i1) * p.page_stride;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = min(t + u * GR + grp, w1 - 1);      // clamp: loads stay in range
            kk[u] = ld16<NT>(kb + row_off(row, i0, o0, o1));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = min(t + u * GR + grp, w1 - 1);
            vv[u] = ld16<NT>(vb + row_off(row, i0, o0, o1));
        }
    };
    auto consume = [&](const uint4 (&kk)[U], const uint4 (&vv)[U], int t) {
        float x[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) unpack8<Tr>(vv[u], x[u]);   // V unpacked once, shared by the G heads
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float s[U];
            float mx = st[g].m;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float d = group_sum<LPR>(dot8<Tr>(kk[u], qpk[g]));
                s[u] = (t + u * GR + grp < w1) ? d * p.scale_log2 : neg_inf();
                mx = fmaxf(mx, s[u]);
            }
            const float ms = (mx == neg_inf()) ? 0.f : mx;
            const float alpha = fast_exp2(st[g].m - ms);
            st[g].l *= alpha;
#pragma unroll
            for (int j = 0; j < 8; ++j) st[g].acc[j] *= alpha;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float pu = fast_exp2(s[u] - ms);
                st[g].l += pu;
#pragma unroll
                for (int j = 0; j < 8; ++j) st[g].acc[j] = fmaf(pu, x[u][j], st[g].acc[j]);
            }
            st[g].m = mx;
        }
    };

    if (w0 < w1) {
        uint4 ka[U], va[U], kb2[U], vb2[U];
        load(ka, va, w0);
        for (int t = w0; t < w1; t += 2 * STEP) {
            const bool more1 = t + STEP < w1;
            if (more1) load(kb2, vb2, t + STEP);
            consume(ka, va, t);
            if (more1) {
                if (t + 2 * STEP < w1) load(ka, va, t + 2 * STEP);
                consume(kb2, vb2, t + STEP);
            }
        }
    }

    // ---- the new token (position `pos`): registers only; last split, wave 0, lane group 0 ----
    if (split == S - 1 && wave == 0) {
        float dn[G];
#pragma unroll
        for (int g = 0; g < G; ++g) dn[g] = group_sum<LPR>(dot8<Tr>(kpk, qpk[g]));
        if (grp == 0) {
            float x[8];
            unpack8<Tr>(vpk, x);
#pragma unroll
            for (int g = 0; g < G; ++g) st[g].merge(dn[g] * p.scale_log2, 1.0f, x);
            long long roff = (long long)pos * rs;               // append: LPR lanes x 16 B = one row each
            if (PAGED) roff = page_of(pos >> p.page_shift) * p.page_stride + (long long)(pos & pmask) * rs;
            *reinterpret_cast<uint4 *>(p.k_cache + head_base + roff) = kpk;
            *reinterpret_cast<uint4 *>(p.v_cache + head_base + roff) = vpk;
        }
    }

    if (PAGED && bad_page) {
#pragma unroll
        for (int g = 0; g < G; ++g) st[g].l = __builtin_nanf("");
    }

    // ---- merge lane groups (same dims, different rows), then the waves through LDS ----
    __shared__ float red[W][G][D + 2];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            const float m2 = __shfl_xor(st[g].m, off), l2 = __shfl_xor(st[g].l, off);
            float a2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a2[j] = __shfl_xor(st[g].acc[j], off);
            st[g].merge(m2, l2, a2);
        }
        if (grp == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wave][g][sub * 8 + j] = st[g].acc[j];
            if (sub == 0) { red[wave][g][D] = st[g].m; red[wave][g][D + 1] = st[g].l; }
        }
    }
    __syncthreads();
    if (tid < LPR * G) {
        const int g = tid / LPR, sb = tid % LPR;
        Stream tot;
        tot.init();
#pragma unroll
        for (int w = 0; w < W; ++w) {
            float a2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a2[j] = red[w][g][sb * 8 + j];
            tot.merge(red[w][g][D], red[w][g][D + 1], a2);
        }
        const long long bh = (long long)b * Hq + hk * G + g;
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

template <class Tr, int D, int G>
int launch_g(const DecodeKernelParams &p, hipStream_t stream) {
    dim3 grid(p.Hkv, p.num_splits, p.B), block(kDecodeWaves * 64);
    bool nt = 4ll * p.B * p.L * p.M * p.Hkv * D > (256ll << 20);       // see decode_kernel.hip
    if (const int k = g_knobs.decode_nt.load(std::memory_order_relaxed); k >= 0) nt = k != 0;      // tests, A/B
    if (p.block_table) {
        if (nt) hipLaunchKernelGGL((decode_gqa_kernel<Tr, D, G, true, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((decode_gqa_kernel<Tr, D, G, false, true>), grid, block, 0, stream, p);
    } else {
        if (nt) hipLaunchKernelGGL((decode_gqa_kernel<Tr, D, G, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((decode_gqa_kernel<Tr, D, G, false>), grid, block, 0, stream, p);
    }
    return check_launch("decode_gqa_kernel");
}

template <class Tr, int D>
int launch_t(const DecodeKernelParams &p, hipStream_t stream) {
    switch (p.H / p.Hkv) {
        case 2: return launch_g<Tr, D, 2>(p, stream);
        case 4: return launch_g<Tr, D, 4>(p, stream);
        case 8: return launch_g<Tr, D, 8>(p, stream);
        default: return fail(SFA_ERR_BAD_SHAPE, "sfa_decode: num_heads / num_heads_kv = %d not in {1, 2, 4, 8}", p.H / p.Hkv);
    }
}

}  // namespace

// the attention kernel only; launch_decode (decode_kernel.hip) adds the split combine
int launch_decode_gqa(const DecodeKernelParams &p, int dtype, int head_dim, hipStream_t stream) {
    // 8 query heads per kv head make this kernel VALU-bound (4.1 TB/s at head_dim 128): the matrix-core form takes
    // over -- for groups of 16 always, for 8 at any head_dim, for 4 at head_dim 128 (measured there); 
    // sfa_debug_set("decode_gqa_mfma", 0) keeps the VALU kernel for A/B, 1 forces the matrix-core kernel for groups of 4
    if ((head_dim == 128 || head_dim == 64 || head_dim == 256) && (p.H == 16 * p.Hkv || p.H == 8 * p.Hkv || p.H == 4 * p.Hkv) &&
        (dtype == SFA_DTYPE_FP16 || dtype == SFA_DTYPE_BF16)) {
        const int knob = g_knobs.decode_gqa_mfma.load(std::memory_order_relaxed);
        const bool by_default = head_dim == 128 || p.H >= 8 * p.Hkv;
        if (p.H == 16 * p.Hkv || (knob < 0 ? by_default : knob != 0)) return launch_decode_gqa_mfma(p, dtype, head_dim, stream);
    }
    if (dtype == SFA_DTYPE_FP16) {
        if (head_dim == 128) return launch_t<Fp16, 128>(p, stream);
        if (head_dim == 256) return launch_t<Fp16, 256>(p, stream);
        if (head_dim == 64) return launch_t<Fp16, 64>(p, stream);
    } else if (dtype == SFA_DTYPE_BF16) {
        if (head_dim == 128) return launch_t<Bf16, 128>(p, stream);
        if (head_dim == 256) return launch_t<Bf16, 256>(p, stream);
        if (head_dim == 64) return launch_t<Bf16, 64>(p, stream);
    } else {
        return fail(SFA_ERR_BAD_DTYPE, "sfa_decode: dtype %d is not fp16(0)/bf16(1)", dtype);
    }
    return fail(SFA_ERR_UNSUPPORTED_HEAD_DIM, "sfa_decode: head_dim %d not in {64, 128, 256}", head_dim);
}

}  // namespace sfa
