// Single-token decode attention for gfx950 (MI355X): fused QKV unpack + bias + interleaved
// RoPE + KV-cache append + split-KV online softmax, and the split combine.
//
// Replaces (semantics, not code) the reference's flash_decoder_kernel / flash_combine_kernel
// and their device helpers (src/flash_attn.cu:554-875, 877-935, 161-447) -- see SURVEY.md
// section 8(a) rows A3-A9 for what the CUDA version intends and where it goes wrong.
//
// MI355X design (HBM-bound: AI = 1 FLOP/B, so the only goal is to keep HBM busy):
//   * cache layout is the API's [B, L, M, H, D] (one (b,h) reads 2*D-byte segments at a
//     stride of H*D*2 bytes) or, opt-in, head-major [B, L, H, M, D] (kv_row_stride / kv_head_stride).  D/8 lanes x 16 B cover one cache row, so one wave64
//     global_load_dwordx4 fetches 64/(D/8) whole rows (4 rows for D=128): every 128-B line
//     that is fetched is fully used, and neighbouring heads (blockIdx.x fastest) touch the
//     same DRAM pages at about the same time.
//   * K/V go straight to VGPRs (no LDS round trip: nothing is reused; non-temporal loads when the
//     caches exceed the Infinity Cache: +3 %; 8 or 2 waves per workgroup and U = 8 measured no better), U row-groups per
//     step, register double-buffered so 2*U K-loads + 2*U V-loads (16 B/lane each) are in
//     flight per wave; 4 waves/workgroup split the workgroup's key range.
//   * q.k uses v_dot2c_f32_{f16,bf16}; the 16-lane (8 for D=64) row sum is 4 (3) DPP adds.
//     Each of the 64/(D/8) lane groups runs its OWN online softmax over the rows it sees,
//     so nothing crosses lane groups inside the loop; groups merge once per wave
//     (__shfl_xor), waves merge once per workgroup through 2 KB of LDS.
//   * 64-bit addressing throughout (K alone is 8.6 G elements at B=256, M=8192, H=32).
//   * The new token never round-trips through memory: the last split's wave 0 takes
//     k_rot / v_new from registers, adds them to its softmax stream and writes the cache row.
#include <cstdlib>

#include "decode_common.h"

namespace sfa {

namespace {

using namespace decode;

template <class Tr, int D, int U, bool NT, bool PAGED = false, int W = kDecodeWaves>
__global__ void __launch_bounds__(W * 64)
decode_kernel(const DecodeKernelParams p) {
    constexpr int LPR = D / 8;          // lanes per cache row
    constexpr int G = 64 / LPR;         // cache rows per wave-instruction
    constexpr int STEP = G * U;         // rows per wave per step
    const int h = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int S = p.num_splits;

    const int pos = p.seq_len[b];
    // out-of-range sequence, or (paged) an append page outside the pool: touch no cache row, poison the
    // output, raise the sticky flag.  Every workgroup of batch b takes the same decision.
    const int reject = reject_code<PAGED>(p, b, pos);
    if (reject) {
        if (split == 0) {
            if (tid < D) p.o[((long long)b * p.H + h) * D + tid] = Tr::id == 0 ? 0x7e00 : 0x7fc0;
            if (tid == 0 && h == 0) atomicOr(p.status, reject);
        }
        return;
    }

    // ---- q, k_new, v_new for this lane's 8 dims: bias, RoPE (fp32), round to storage ----
    const long long qoff = (long long)b * p.qkv_stride + (long long)h * D + sub * 8;
    const long long hd = (long long)p.H * D;
    float xq[8], xk[8], xv[8];
    unpack8<Tr>(*reinterpret_cast<const uint4 *>(p.qkv + qoff), xq);
    unpack8<Tr>(*reinterpret_cast<const uint4 *>(p.qkv + qoff + hd), xk);
    const uint4 v_raw = *reinterpret_cast<const uint4 *>(p.qkv + qoff + 2 * hd);
    uint4 vpk = v_raw;
    if (p.q_bias) {
        float t[8]; unpack8<Tr>(*reinterpret_cast<const uint4 *>(p.q_bias + (long long)h * D + sub * 8), t);
#pragma unroll
        for (int j = 0; j < 8; ++j) xq[j] += t[j];
    }
    if (p.k_bias) {
        float t[8]; unpack8<Tr>(*reinterpret_cast<const uint4 *>(p.k_bias + (long long)h * D + sub * 8), t);
#pragma unroll
        for (int j = 0; j < 8; ++j) xk[j] += t[j];
    }
    if (p.v_bias) {
        float t[8]; unpack8<Tr>(*reinterpret_cast<const uint4 *>(p.v_bias + (long long)h * D + sub * 8), t);
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
            } else {
                // same fp32 recipe as the reference oracle (testFlashDecoder.py:11,20-25)
                const float inv_freq = 1.0f / powf(10000.0f, (float)(2 * pj) / (float)rot);
                const float ang = (float)pos * inv_freq;
                sincosf(ang, &s, &c);
            }
            const float q0 = xq[2 * i], q1 = xq[2 * i + 1];
            xq[2 * i] = q0 * c - q1 * s;
            xq[2 * i + 1] = q1 * c + q0 * s;
            const float k0 = xk[2 * i], k1 = xk[2 * i + 1];
            xk[2 * i] = k0 * c - k1 * s;
            xk[2 * i + 1] = k1 * c + k0 * s;
        }
    }
    const uint4 qpk = pack8<Tr>(xq);
    const uint4 kpk = pack8<Tr>(xk);

    // ---- this wave's slice of the cached rows [0, pos) ----
    const int rows_per_split = (pos + S - 1) / S;
    const int r0 = min(pos, split * rows_per_split);
    const int r1 = min(pos, r0 + rows_per_split);
    int per_wave = (r1 - r0 + W - 1) / W;
    per_wave = (per_wave + STEP - 1) / STEP * STEP;
    // (readfirstlane: wave-uniform by construction, and the paged path wants scalar table loads)
    const int w0 = __builtin_amdgcn_readfirstlane(min(r1, r0 + wave * per_wave));
    const int w1 = __builtin_amdgcn_readfirstlane(min(r1, w0 + per_wave));

    const long long rs = p.kv_row_stride;       // elements between consecutive cache rows of this head
    // contiguous layouts: rows of this (b, layer, h) start at head_base, rs apart.
    // paged: row r lives in page table[r >> page_shift] of the pool, at row r & (page_size-1).
    const long long head_base = PAGED ? (long long)p.layer * (rs << p.page_shift) + (long long)h * p.kv_head_stride + sub * 8
                                      : ((long long)b * p.L + p.layer) * p.M * hd + h * p.kv_head_stride + sub * 8;
    const uint16_t *kb = p.k_cache + head_base;
    const uint16_t *vb = p.v_cache + head_base;
    const int32_t *tbl = PAGED ? p.block_table + (long long)b * p.table_stride : nullptr;   // uniform
    const int pmask = PAGED ? (1 << p.page_shift) - 1 : 0;
    // A READ page outside the pool is not dereferenced: page 0 is read in its place, the sticky flag is
    // raised and the workgroup's row sum becomes NaN, so o[b, h] comes out NaN instead of plausible.
    // (The append page was checked above: nothing is ever stored through a substituted page.)
    int bad_page = 0;
    auto page_of = [&](int idx) -> int {        // scalar: table entry, clamped into the pool
        int pg = tbl[idx];
        if ((unsigned)pg >= (unsigned)p.num_pages) {
            if (tid == 0) atomicOr(p.status, 2);        // sticky: a block_table entry outside the pool
            bad_page = 1;
            pg = 0;
        }
        return pg;
    };

    Stream st;
    st.init();

    // A step covers STEP <= 16 consecutive rows starting at the wave-uniform t: with page_size >= 16 they
    // touch at most two pages, looked up with two SCALAR loads (no vmcnt ordering against the
    // in-flight K/V loads) and selected per lane.
    auto row_off = [&](int row, int i0, long long o0, long long o1) -> long long {
        if (!PAGED) return (long long)row * rs;
        return ((row >> p.page_shift) == i0 ? o0 : o1) + (long long)(row & pmask) * rs;
    };
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
            const int row = min(t + u * G + grp, w1 - 1);       // clamp: loads stay in range
            kk[u] = ld16<NT>(kb + row_off(row, i0, o0, o1));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = min(t + u * G + grp, w1 - 1);
            vv[u] = ld16<NT>(vb + row_off(row, i0, o0, o1));
        }
    };
    auto consume = [&](const uint4 (&kk)[U], const uint4 (&vv)[U], int t) {
        float s[U];
        float mx = st.m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float d = group_sum<LPR>(dot8<Tr>(kk[u], qpk));
            s[u] = (t + u * G + grp < w1) ? d * p.scale_log2 : neg_inf();
            mx = fmaxf(mx, s[u]);
        }
        const float ms = (mx == neg_inf()) ? 0.f : mx;
        const float alpha = fast_exp2(st.m - ms);
        st.l *= alpha;
#pragma unroll
        for (int j = 0; j < 8; ++j) st.acc[j] *= alpha;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float pu = fast_exp2(s[u] - ms);
            st.l += pu;
            float x[8];
            unpack8<Tr>(vv[u], x);
#pragma unroll
            for (int j = 0; j < 8; ++j) st.acc[j] = fmaf(pu, x[j], st.acc[j]);
        }
        st.m = mx;
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
        const float d = group_sum<LPR>(dot8<Tr>(kpk, qpk));
        if (grp == 0) {
            float x[8];
            unpack8<Tr>(vpk, x);
            const float sn = d * p.scale_log2;
            float accn[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) accn[j] = x[j];
            st.merge(sn, 1.0f, accn);
            // append to the caches: LPR lanes x 16 B = one row each
            long long roff = (long long)pos * rs;
            if (PAGED) roff = page_of(pos >> p.page_shift) * p.page_stride + (long long)(pos & pmask) * rs;
            *reinterpret_cast<uint4 *>(p.k_cache + head_base + roff) = kpk;
            *reinterpret_cast<uint4 *>(p.v_cache + head_base + roff) = vpk;
        }
    }

    if (PAGED && bad_page) st.l = __builtin_nanf("");

    // ---- merge lane groups (same dims, different rows) ----
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(st.m, off), l2 = __shfl_xor(st.l, off);
        float a2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a2[j] = __shfl_xor(st.acc[j], off);
        st.merge(m2, l2, a2);
    }

    // ---- merge the workgroup's waves through LDS ----
    __shared__ float red[W][D + 2];
    if (grp == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave][sub * 8 + j] = st.acc[j];
        if (sub == 0) { red[wave][D] = st.m; red[wave][D + 1] = st.l; }
    }
    __syncthreads();
    if (tid < LPR) {
        Stream tot;
        tot.init();
#pragma unroll
        for (int w = 0; w < W; ++w) {
            float a2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a2[j] = red[w][tid * 8 + j];
            tot.merge(red[w][D], red[w][D + 1], a2);
        }
        const long long bh = (long long)b * p.H + h;
        if (S == 1) {
            const float inv = 1.0f / tot.l;          // l >= 1: the new token is always present
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = tot.acc[j] * inv;
            *reinterpret_cast<uint4 *>(p.o + bh * D + tid * 8) = pack8<Tr>(y);
        } else {
            float *po = p.part_o + (bh * S + split) * D + tid * 8;
            *reinterpret_cast<float4 *>(po) = make_float4(tot.acc[0], tot.acc[1], tot.acc[2], tot.acc[3]);
            *reinterpret_cast<float4 *>(po + 4) = make_float4(tot.acc[4], tot.acc[5], tot.acc[6], tot.acc[7]);
            if (tid == 0) p.part_ml[bh * S + split] = make_float2(tot.m, tot.l);
        }
    }
}

// o[b,h,:] = sum_s 2^(m_s - M) o_s / sum_s 2^(m_s - M) l_s     (fp32; reference cu:877-935 does
// this in half precision with a -256 sentinel)
template <class Tr, int D>
__global__ void __launch_bounds__(256)
decode_combine_kernel(const DecodeKernelParams p) {
    constexpr int LPR = D / 8;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long bh = gid / LPR;
    const int sub = (int)(gid % LPR);
    if (bh >= (long long)p.B * p.H) return;
    const int b = (int)(bh / p.H);
    const int pos = p.seq_len[b];
    if (p.block_table ? reject_code<true>(p, b, pos) : reject_code<false>(p, b, pos)) return;   // o[b] is already poisoned
    const int S = p.num_splits;
    Stream tot;
    tot.init();
    for (int s = 0; s < S; ++s) {
        const float2 ml = p.part_ml[bh * S + s];
        const float *po = p.part_o + (bh * S + s) * D + sub * 8;
        const float4 a = *reinterpret_cast<const float4 *>(po);
        const float4 c = *reinterpret_cast<const float4 *>(po + 4);
        const float a2[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        tot.merge(ml.x, ml.y, a2);
    }
    const float inv = 1.0f / tot.l;
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = tot.acc[j] * inv;
    *reinterpret_cast<uint4 *>(p.o + bh * D + sub * 8) = pack8<Tr>(y);
}

template <class Tr, int D>
int launch_decode_t(const DecodeKernelParams &p, int dtype, hipStream_t stream) {
    int rc;
    if (p.Hkv != p.H) {
        // grouped queries: one workgroup per (batch, kv head, split) serves the whole group
        rc = launch_decode_gqa(p, dtype, D, stream);
    } else {
        dim3 grid(p.H, p.num_splits, p.B), block(kDecodeWaves * 64);
        // The cache rows are read exactly once per call.  When the two caches together do not fit the
        // 256 MB Infinity Cache nothing of them survives until the next token's call either, so they are
        // loaded non-temporally (config 4: 6.30 -> 6.51 TB/s); a small cache keeps the default policy
        // and is re-read from the Infinity Cache / L2.  sfa_debug_set("decode_nt", 0/1) overrides (tests, A/B).
        bool nt = 4ll * p.B * p.L * p.M * p.H * D > (256ll << 20);
        if (const int k = g_knobs.decode_nt.load(std::memory_order_relaxed); k >= 0) nt = k != 0;      // tests, A/B
        if (p.block_table) {
            // a step of at most 16 rows touches at most two pages (page_size >= 16)
            constexpr int UP = (16 / (64 / (D / 8))) < 4 ? (16 / (64 / (D / 8))) : 4;
            if (nt) hipLaunchKernelGGL((decode_kernel<Tr, D, UP, true, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((decode_kernel<Tr, D, UP, false, true>), grid, block, 0, stream, p);
        } else {
            if (nt) hipLaunchKernelGGL((decode_kernel<Tr, D, 4, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((decode_kernel<Tr, D, 4, false>), grid, block, 0, stream, p);
        }
        rc = check_launch("decode_kernel");
    }
    if (rc != SFA_OK) return rc;
    if (p.num_splits > 1) {
        const long long threads = (long long)p.B * p.H * (D / 8);
        dim3 g2((unsigned)((threads + 255) / 256)), b2(256);
        hipLaunchKernelGGL((decode_combine_kernel<Tr, D>), g2, b2, 0, stream, p);
        rc = check_launch("decode_combine_kernel");
    }
    return rc;
}

}  // namespace

int launch_decode(const DecodeKernelParams &p, int dtype, int head_dim, hipStream_t stream) {
    if (dtype == SFA_DTYPE_FP16) {
        if (head_dim == 128) return launch_decode_t<Fp16, 128>(p, dtype, stream);
        if (head_dim == 256) return launch_decode_t<Fp16, 256>(p, dtype, stream);
        if (head_dim == 64) return launch_decode_t<Fp16, 64>(p, dtype, stream);
    } else if (dtype == SFA_DTYPE_BF16) {
        if (head_dim == 128) return launch_decode_t<Bf16, 128>(p, dtype, stream);
        if (head_dim == 256) return launch_decode_t<Bf16, 256>(p, dtype, stream);
        if (head_dim == 64) return launch_decode_t<Bf16, 64>(p, dtype, stream);
    } else {
        return fail(SFA_ERR_BAD_DTYPE, "sfa_decode: dtype %d is not fp16(0)/bf16(1)", dtype);
    }
    return fail(SFA_ERR_UNSUPPORTED_HEAD_DIM, "sfa_decode: head_dim %d not in {64, 128, 256}", head_dim);
}

}  // namespace sfa
