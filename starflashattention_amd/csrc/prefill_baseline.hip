// BASELINE generation of the prefill kernel (one tile at a time, no software pipeline, XOR-swizzled
// LDS images): kept in the library only as the in-process A/B reference for tools/prefill_ab.py
// (SFA_PREFILL_IMPL=0); 538 TFLOPS where the product kernel (prefill_kernel.hip) reaches ~900.
// bf16 / fp16, head_dim 64 / 128, causal or full, MHA or GQA.  The reference has no prefill kernel; this is the new entry
// point BASELINE.json's headline metric is quoted on (SURVEY.md section 8(a) row A-new).
//
// MI355X design (MFMA-bound: AI = 1024 FLOP/B at S=4096, D=128):
//   * workgroup = 8 waves = 256 query rows of one (batch, head); wave w owns rows
//     [32w, 32w+32).  K/V tiles of 64 keys are staged once per workgroup into LDS
//     (register-staged, double-buffered, one barrier per tile) and shared by all 8 waves.
//   * S^T = K . Q^T with v_mfma_f32_32x32x16 (A = K rows from LDS, B = Q^T held in
//     registers for the whole kernel): the 32x32 accumulator has the QUERY on the lane and
//     keys in registers, so the online-softmax row max / row sum are in-lane loops plus one
//     v_permlane32_swap -- no LDS, no ds_bpermute.
//   * O^T += V^T . P^T: the S^T accumulator, exponentiated and converted to 16 bit in
//     place, is already the B operand of the second MFMA (it sums over the accumulator's
//     row index); A = V^T comes from the row-major V tile through ds_read_b64_tr_b16
//     (hardware transpose).  The O^T accumulator again has the query on the lane, so the
//     softmax rescale is one scalar per lane.
//   * LDS images: K rows XOR-swizzled for conflict-free ds_read_b128 of the A operand; V rows
//     XOR-swizzled for conflict-free transposed reads (cdna_hip_programming.md T2 / T10).
//   * blockIdx -> (head, q-tile) is XCD-aware: the 8 XCDs each own a contiguous range of
//     (batch, head) pairs, walk a head's q-tiles heaviest-first (causal), so all q-tiles of a
//     head stream the same K/V through one XCD's L2 at about the same time.
#include "prefill_common.h"

namespace sfa {

namespace {

using namespace prefill;

template <class Tr, int D, bool CAUSAL>
__global__ void __launch_bounds__(kThreads, 2)
prefill_kernel_baseline(const PrefillKernelParams p) {
    using Vec = typename Tr::mfma_vec;
    constexpr int NKS = D / 16;                 // k-steps of Q.K^T
    constexpr int NDB = D / 32;                 // 32-wide d blocks of O^T
    constexpr int CPR = D / 8;                  // 16-B chunks per row
    constexpr int NLD = kBN * CPR / kThreads;   // chunks staged per thread per tile
    constexpr int TILE_BYTES = kBN * D * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // K[2] then V[2]

    // ---- which (batch, head, q-tile) ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int bh = xcd * p.bh_per_xcd + slot / p.nq_tiles;
    const int qt = p.nq_tiles - 1 - (slot % p.nq_tiles);      // heaviest (last) q-tile first
    if (bh >= p.B * p.Hq) return;
    const int b = bh / p.Hq, h = bh % p.Hq;
    const int hk = h / (p.Hq / p.Hkv);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform
    const int l31 = lane & 31, h2 = lane >> 5;
    const int q0 = qt * kBM;
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

    int kv_end = p.Sk;
    if (CAUSAL) kv_end = min(p.Sk, q0 + kBM + coff);
    const int nt = kv_end > 0 ? (kv_end + kBN - 1) / kBN : 0;

    const uint16_t *kg = p.k + b * p.ks[0] + hk * p.ks[1];
    const uint16_t *vg = p.v + b * p.vs[0] + hk * p.vs[1];

    uint4 kst[NLD], vst[NLD];
    auto stage_load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + kThreads * i;
            const int row = c / CPR, ch = c % CPR;
            const long long krow = min(kt * kBN + row, p.Sk - 1);
            kst[i] = *reinterpret_cast<const uint4 *>(kg + krow * p.ks[2] + ch * 8);
            vst[i] = *reinterpret_cast<const uint4 *>(vg + krow * p.vs[2] + ch * 8);
        }
    };
    auto stage_store = [&](int buf) {
        char *kb = smem + buf * TILE_BYTES;
        char *vb = smem + (2 + buf) * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + kThreads * i;
            const int row = c / CPR, ch = c % CPR;
            *reinterpret_cast<uint4 *>(kb + k_off<D>(row, ch)) = kst[i];
            *reinterpret_cast<uint4 *>(vb + v_off<D>(row, ch)) = vst[i];
        }
    };

    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float msc = ninf();     // running row max, in log2 units (raw max * scale*log2e)
    float lsum = 0.f;       // this lane's share of the running row sum
    const float c2 = p.scale_log2;

    // lane-constant parts of the LDS read addresses
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g = (lane >> 4) & 1;

    if (nt > 0) {
        stage_load(0);
        stage_store(0);
    }
    __syncthreads();

    for (int kt = 0; kt < nt; ++kt) {
        const bool more = kt + 1 < nt;
        if (more) stage_load(kt + 1);               // HBM/L2 latency hides under the MFMAs below

        const int k0 = kt * kBN;
        const bool active = !CAUSAL || (k0 <= wq0 + 31 + coff);     // wave-uniform
        if (active) {
            const char *kb = smem + (kt & 1) * TILE_BYTES;
            const char *vb = smem + (2 + (kt & 1)) * TILE_BYTES;

            // ---- S^T = K . Q^T : two 32-key x 32-query accumulators ----
            f32x16 s0, s1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const Vec a0 = bitcast<Vec>(*reinterpret_cast<const uint4 *>(kb + k_off<D>(l31, 2 * ks + h2)));
                const Vec a1 = bitcast<Vec>(*reinterpret_cast<const uint4 *>(kb + k_off<D>(32 + l31, 2 * ks + h2)));
                s0 = Tr::mfma32(a0, qf[ks], s0);
                s1 = Tr::mfma32(a1, qf[ks], s1);
            }

            // ---- mask (diagonal tiles and the ragged last tile only) ----
            const bool need_mask = (CAUSAL && (k0 + kBN - 1 > wq0 + coff)) || (k0 + kBN > p.Sk);
            if (need_mask) {
                const int lim = CAUSAL ? min(p.Sk - 1, qrow + coff) : p.Sk - 1;   // last visible key
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                    if (key > lim) s0[r] = ninf();
                    if (key + 32 > lim) s1[r] = ninf();
                }
            }

            // ---- online softmax: the query is on the lane ----
            float mx = s0[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[r]);
            mx = half_max(mx);                                   // both halves hold the same query
            const float mnew = fmaxf(msc, mx * c2);
            const float msafe = (mnew == ninf()) ? 0.f : mnew;
            const float alpha = fast_exp2(msc - msafe);
            msc = mnew;
            lsum *= alpha;
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] = fast_exp2(fmaf(s0[r], c2, -msafe));
                s1[r] = fast_exp2(fmaf(s1[r], c2, -msafe));
                psum += s0[r] + s1[r];
            }
            lsum += psum;

            // P^T as the B operand of the PV product: registers 8s..8s+7 of accumulator i are
            // k-step 2i+s (cdna_hip_programming.md section 3, "An accumulator tile as the next
            // MFMA's operand").
            Vec pb[4];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 w0, w1;
                w0.x = Tr::pack2(s0[8 * s + 0], s0[8 * s + 1]); w0.y = Tr::pack2(s0[8 * s + 2], s0[8 * s + 3]);
                w0.z = Tr::pack2(s0[8 * s + 4], s0[8 * s + 5]); w0.w = Tr::pack2(s0[8 * s + 6], s0[8 * s + 7]);
                w1.x = Tr::pack2(s1[8 * s + 0], s1[8 * s + 1]); w1.y = Tr::pack2(s1[8 * s + 2], s1[8 * s + 3]);
                w1.z = Tr::pack2(s1[8 * s + 4], s1[8 * s + 5]); w1.w = Tr::pack2(s1[8 * s + 6], s1[8 * s + 7]);
                pb[s] = bitcast<Vec>(w0);
                pb[2 + s] = bitcast<Vec>(w1);
            }

            // ---- O^T += V^T . P^T : A operand by transposed LDS reads ----
#pragma unroll
            for (int d = 0; d < NDB; ++d) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    // element j of lane half h2 must be key 16kk + 8(j>>2) + 4*h2 + (j&3)
                    const int rA = 16 * kk + 4 * h2 + tr_q;
                    const int ch = 4 * d + 2 * tr_g + (tr_p >> 1);
                    const i16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (lds_i16x4 *)(vb + v_off<D>(rA, ch) + 8 * (tr_p & 1)));
                    const i16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (lds_i16x4 *)(vb + v_off<D>(rA + 8, ch) + 8 * (tr_p & 1)));
                    u32x4 av;
                    const u32x2 a_lo = bitcast<u32x2>(t0), a_hi = bitcast<u32x2>(t1);
                    av[0] = a_lo[0]; av[1] = a_lo[1]; av[2] = a_hi[0]; av[3] = a_hi[1];
                    o[d] = Tr::mfma32(bitcast<Vec>(av), pb[kk], o[d]);
                }
            }
        }

        if (more) stage_store((kt + 1) & 1);        // that buffer was last read in iteration kt-1
        __syncthreads();
    }

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
}

template <class Tr, int D>
int launch_prefill_t(const PrefillKernelParams &p, bool causal, hipStream_t stream) {
    const size_t lds = 4 * (size_t)kBN * D * 2;
    dim3 grid(8u * p.bh_per_xcd * p.nq_tiles), block(kThreads);
    static DynLdsAttr attr_c, attr_f;
    if (const int rc = causal ? attr_c.ensure(reinterpret_cast<const void *>(&prefill_kernel_baseline<Tr, D, true>),
                                              (int)lds, "prefill_kernel_baseline")
                              : attr_f.ensure(reinterpret_cast<const void *>(&prefill_kernel_baseline<Tr, D, false>),
                                              (int)lds, "prefill_kernel_baseline"))
        return rc;
    if (causal) {
        hipLaunchKernelGGL((prefill_kernel_baseline<Tr, D, true>), grid, block, lds, stream, p);
    } else {
        hipLaunchKernelGGL((prefill_kernel_baseline<Tr, D, false>), grid, block, lds, stream, p);
    }
    return check_launch("prefill_kernel_baseline");
}

}  // namespace

int launch_prefill_baseline(const PrefillKernelParams &p, int dtype, int head_dim, bool causal, hipStream_t stream) {
    if (dtype == SFA_DTYPE_FP16) {
        if (head_dim == 128) return launch_prefill_t<Fp16, 128>(p, causal, stream);
        if (head_dim == 64) return launch_prefill_t<Fp16, 64>(p, causal, stream);
    } else if (dtype == SFA_DTYPE_BF16) {
        if (head_dim == 128) return launch_prefill_t<Bf16, 128>(p, causal, stream);
        if (head_dim == 64) return launch_prefill_t<Bf16, 64>(p, causal, stream);
    } else {
        return fail(SFA_ERR_BAD_DTYPE, "sfa_prefill_fwd: dtype %d is not fp16(0)/bf16(1)", dtype);
    }
    return fail(SFA_ERR_UNSUPPORTED_HEAD_DIM, "sfa_prefill_fwd: head_dim %d not in {64, 128}", head_dim);
}

}  // namespace sfa
