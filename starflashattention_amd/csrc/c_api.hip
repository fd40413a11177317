// extern "C" entry points of libStarFlashAttention.so (declared in include/star_flash_attn.h).
// Validation + parameter marshalling only; the kernels live in decode_kernel.hip,
// prefill_kernel.hip and aux_kernels.hip.  Nothing here allocates, copies or synchronises
// (except sfa_decode_poll_status, which exists to do exactly that).
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "sfa_host.h"

namespace sfa {

static thread_local char g_err[512] = "";

DebugKnobs g_knobs;

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int status, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return status;
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SFA_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SFA_OK;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Split count when the caller does not choose one (the reference hard-codes 4 with a TODO,
// flash_api.cpp:38, flash_attn.cu:1024).  Each workgroup is 4 waves on one (b,h,split); aim
// for >= 4 workgroups per CU (1024 on 256 CUs) but keep >= 256 cached rows per workgroup so
// every wave still streams >= 64 rows.
// Split count for num_splits <= 0.  Measured on MI355X (tools/decode_small.py, fp16 D=128): splitting
// pays only while B*H alone leaves most CUs without a workgroup -- B=1 H=32 M=8192: 72.8 us unsplit,
// 29.1 us at 4-8 splits, 35.7 at 32; B=2: 74.2 -> 46.8 at 2 splits, 52.0 at 16; from B*H >= 256 on one
// split is best (the combine kernel and the partials cost ~3 us).  Aim for ~128 workgroups, and keep
// every split at least 2048 cache rows long.
static int auto_splits(int B, int H, int /*D*/, int M) {
    const long long bh = (long long)B * H;
    long long s = (128 + bh - 1) / bh;
    const long long cap = M / 2048 > 1 ? M / 2048 : 1;
    if (s > cap) s = cap;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    return (int)s;
}

}  // namespace sfa

using namespace sfa;

extern "C" {

int sfa_abi_version(void) { return SFA_ABI_VERSION; }

const char *sfa_status_string(int status) {
    switch (status) {
        case SFA_OK: return "ok";
        case SFA_ERR_NULL_POINTER: return "null pointer";
        case SFA_ERR_BAD_SHAPE: return "bad shape";
        case SFA_ERR_BAD_DTYPE: return "bad dtype";
        case SFA_ERR_UNSUPPORTED_HEAD_DIM: return "unsupported head_dim";
        case SFA_ERR_WORKSPACE_TOO_SMALL: return "workspace too small";
        case SFA_ERR_LAUNCH: return "HIP launch failure";
        case SFA_ERR_SEQ_LEN_RANGE: return "seq_len out of range";
        case SFA_ERR_BLOCK_TABLE_RANGE: return "block_table entry out of range";
        default: return "unknown status";
    }
}

const char *sfa_last_error(void) { return g_err; }

int sfa_debug_set(const char *knob, int value) {
    if (!knob) return fail(SFA_ERR_NULL_POINTER, "sfa_debug_set: knob is NULL");
    std::atomic<int> *k = nullptr;
    if (!strcmp(knob, "prefill_impl")) k = &g_knobs.prefill_impl;
    else if (!strcmp(knob, "prefill_pairs")) k = &g_knobs.prefill_pairs;
    else if (!strcmp(knob, "decode_nt")) k = &g_knobs.decode_nt;
    else if (!strcmp(knob, "decode_gqa_mfma")) k = &g_knobs.decode_gqa_mfma;
    else if (!strcmp(knob, "bm128_one_wg")) k = &g_knobs.bm128_one_wg;
    else return fail(SFA_ERR_BAD_SHAPE, "sfa_debug_set: unknown knob '%s'", knob);
    k->store(value, std::memory_order_relaxed);
    return SFA_OK;
}

int sfa_debug_get(const char *knob) {
    if (!knob) return INT_MIN;
    if (!strcmp(knob, "prefill_impl")) return g_knobs.prefill_impl.load(std::memory_order_relaxed);
    if (!strcmp(knob, "prefill_pairs")) return g_knobs.prefill_pairs.load(std::memory_order_relaxed);
    if (!strcmp(knob, "decode_nt")) return g_knobs.decode_nt.load(std::memory_order_relaxed);
    if (!strcmp(knob, "decode_gqa_mfma")) return g_knobs.decode_gqa_mfma.load(std::memory_order_relaxed);
    if (!strcmp(knob, "bm128_one_wg")) return g_knobs.bm128_one_wg.load(std::memory_order_relaxed);
    if (!strcmp(knob, "last_prefill_kernel")) return g_knobs.last_prefill_kernel.load(std::memory_order_relaxed);
    return INT_MIN;
}

int sfa_decode_auto_splits(int batch_size, int num_heads, int head_dim, int memory_max_len) {
    if (batch_size <= 0 || num_heads <= 0 || memory_max_len <= 0) return 1;
    return auto_splits(batch_size, num_heads, head_dim, memory_max_len);
}

size_t sfa_decode_workspace_bytes(int batch_size, int num_heads, int head_dim, int memory_max_len,
                                  int num_splits) {
    if (batch_size <= 0 || num_heads <= 0 || head_dim <= 0) return kStatusBytes;
    const int S = num_splits > 0 ? num_splits : auto_splits(batch_size, num_heads, head_dim, memory_max_len);
    size_t bytes = kStatusBytes;
    if (S > 1) {
        const size_t bhs = (size_t)batch_size * num_heads * S;
        bytes += align_up(bhs * head_dim * sizeof(float), 256);
        bytes += align_up(bhs * sizeof(float2), 256);
    }
    return bytes;
}

size_t sfa_decode_workspace_bytes_gqa(int batch_size, int num_heads, int num_heads_kv, int head_dim, int memory_max_len,
                                      int num_splits) {
    const int hkv = num_heads_kv > 0 ? num_heads_kv : num_heads;
    const int S = num_splits > 0 ? num_splits : auto_splits(batch_size, hkv, head_dim, memory_max_len);
    return sfa_decode_workspace_bytes(batch_size, num_heads, head_dim, memory_max_len, S);
}

int sfa_decode_reset_status(void *workspace, void *stream) {
    if (!workspace) return fail(SFA_ERR_NULL_POINTER, "sfa_decode_reset_status: workspace is NULL");
    const hipError_t e = hipMemsetAsync(workspace, 0, kStatusBytes, (hipStream_t)stream);
    if (e != hipSuccess) return fail(SFA_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    return SFA_OK;
}

int sfa_decode_poll_status(const void *workspace, void *stream) {
    if (!workspace) return fail(SFA_ERR_NULL_POINTER, "sfa_decode_poll_status: workspace is NULL");
    int32_t word = 0;
    hipError_t e = hipMemcpyAsync(&word, workspace, sizeof(word), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail(SFA_ERR_LAUNCH, "sfa_decode_poll_status: %s", hipGetErrorString(e));
    if (word & 2)
        return fail(SFA_ERR_BLOCK_TABLE_RANGE,
                    "sfa_decode: a block_table entry was outside [0, num_pages); nothing was stored through it and those "
                    "outputs are NaN");
    if (word != 0)
        return fail(SFA_ERR_SEQ_LEN_RANGE,
                    "sfa_decode: some seq_len[b] was outside [0, memory_max_len); those outputs are NaN "
                    "and their cache rows were not written");
    return SFA_OK;
}

int sfa_decode(const sfa_decode_args *a, void *stream) {
    if (!a) return fail(SFA_ERR_NULL_POINTER, "sfa_decode: args is NULL");
    if (!a->qkv || !a->o || !a->seq_len || !a->k_cache_table || !a->v_cache_table)
        return fail(SFA_ERR_NULL_POINTER, "sfa_decode: qkv/o/seq_len/k_cache_table/v_cache_table must be non-NULL");
    if ((a->rotary_cos_table == nullptr) != (a->rotary_sin_table == nullptr))
        return fail(SFA_ERR_NULL_POINTER, "sfa_decode: give both rotary tables or neither");
    if (a->batch_size < 0 || a->num_heads <= 0 || a->memory_max_len <= 0 || a->num_layer <= 0)
        return fail(SFA_ERR_BAD_SHAPE, "sfa_decode: batch_size=%d num_heads=%d memory_max_len=%d num_layer=%d",
                    a->batch_size, a->num_heads, a->memory_max_len, a->num_layer);
    if (a->idx_layer < 0 || a->idx_layer >= a->num_layer)
        return fail(SFA_ERR_BAD_SHAPE, "sfa_decode: idx_layer=%d outside [0, num_layer=%d)", a->idx_layer, a->num_layer);
    if (a->head_dim != 64 && a->head_dim != 128 && a->head_dim != 256)
        return fail(SFA_ERR_UNSUPPORTED_HEAD_DIM, "sfa_decode: head_dim %d not in {64, 128, 256}", a->head_dim);
    if (a->rotary_embedding_dim < 0 || a->rotary_embedding_dim > a->head_dim || (a->rotary_embedding_dim & 1))
        return fail(SFA_ERR_BAD_SHAPE, "sfa_decode: rotary_embedding_dim=%d must be even and in [0, head_dim]",
                    a->rotary_embedding_dim);
    if (a->dtype != SFA_DTYPE_FP16 && a->dtype != SFA_DTYPE_BF16)
        return fail(SFA_ERR_BAD_DTYPE, "sfa_decode: dtype %d is not fp16(0)/bf16(1)", a->dtype);
    const int hkv = a->num_heads_kv > 0 ? a->num_heads_kv : a->num_heads;
    const int group = hkv > 0 ? a->num_heads / hkv : 0;
    if (a->num_heads_kv < 0 || group * hkv != a->num_heads ||
        (group != 1 && group != 2 && group != 4 && group != 8 && group != 16))
        return fail(SFA_ERR_BAD_SHAPE, "sfa_decode: num_heads=%d / num_heads_kv=%d must be 1, 2, 4, 8 or 16",
                    a->num_heads, a->num_heads_kv);
    const long long hd = (long long)hkv * a->head_dim;          // elements per cache row
    const long long row = (long long)(a->num_heads + 2 * hkv) * a->head_dim;    // packed q,k,v of one token
    const long long stride = a->stride > 0 ? a->stride : row;
    if (stride < row || (stride % 8) != 0)
        return fail(SFA_ERR_BAD_SHAPE, "sfa_decode: qkv stride %lld must be >= (H + 2*Hkv)*D and a multiple of 8", stride);
    if (a->num_splits > 1024)
        return fail(SFA_ERR_BAD_SHAPE, "sfa_decode: num_splits=%d > 1024", a->num_splits);
    if (a->kv_layout != SFA_KV_BLMHD && a->kv_layout != SFA_KV_BLHMD && a->kv_layout != SFA_KV_PAGED)
        return fail(SFA_ERR_BAD_SHAPE, "sfa_decode: kv_layout %d is not SFA_KV_BLMHD(0)/SFA_KV_BLHMD(1)/SFA_KV_PAGED(2)",
                    a->kv_layout);
    int page_shift = 0;
    if (a->kv_layout == SFA_KV_PAGED) {
        if (!a->block_table) return fail(SFA_ERR_NULL_POINTER, "sfa_decode: kv_layout PAGED needs block_table");
        if (a->page_size < 16 || (a->page_size & (a->page_size - 1)))
            return fail(SFA_ERR_BAD_SHAPE, "sfa_decode: page_size=%d must be a power of two >= 16", a->page_size);
        while ((1 << page_shift) < a->page_size) ++page_shift;
        if (a->num_pages <= 0 ||
            (long long)a->block_table_stride * a->page_size < (long long)a->memory_max_len)
            return fail(SFA_ERR_BAD_SHAPE,
                        "sfa_decode: num_pages=%d, block_table_stride=%d * page_size=%d must cover memory_max_len=%d",
                        a->num_pages, a->block_table_stride, a->page_size, a->memory_max_len);
        if ((uintptr_t)a->block_table & 3) return fail(SFA_ERR_BAD_SHAPE, "sfa_decode: block_table must be 4-byte aligned");
    }
    const uintptr_t align_or = (uintptr_t)a->qkv | (uintptr_t)a->o | (uintptr_t)a->k_cache_table |
                               (uintptr_t)a->v_cache_table | (uintptr_t)a->q_bias | (uintptr_t)a->k_bias |
                               (uintptr_t)a->v_bias;
    if (align_or & 15) return fail(SFA_ERR_BAD_SHAPE, "sfa_decode: tensors must be 16-byte aligned");
    if (a->batch_size == 0) return SFA_OK;

    // grouped queries launch one workgroup per KV head: the split count follows the kv-head count
    int S = a->num_splits > 0
                ? a->num_splits
                : auto_splits(a->batch_size, hkv, a->head_dim, a->memory_max_len);
    // A caller that left the choice to the library sized its workspace with sfa_decode_workspace_bytes(..., 0), which
    // knows the query-head count only: with grouped queries the library's own choice can be larger than the one that
    // size was computed for.  Take the largest split count the workspace holds rather than fail.
    if (a->num_splits <= 0 && a->workspace)
        while (S > 1 && a->workspace_bytes < sfa_decode_workspace_bytes(a->batch_size, a->num_heads, a->head_dim,
                                                                         a->memory_max_len, S))
            --S;
    const size_t need = sfa_decode_workspace_bytes(a->batch_size, a->num_heads, a->head_dim,
                                                   a->memory_max_len, S);
    if (!a->workspace) return fail(SFA_ERR_NULL_POINTER, "sfa_decode: workspace is NULL (need %zu bytes)", need);
    if (a->workspace_bytes < need)
        return fail(SFA_ERR_WORKSPACE_TOO_SMALL, "sfa_decode: workspace has %zu bytes, need %zu",
                    a->workspace_bytes, need);
    if ((uintptr_t)a->workspace & 255)
        return fail(SFA_ERR_BAD_SHAPE, "sfa_decode: workspace must be 256-byte aligned");

    DecodeKernelParams p;
    memset(&p, 0, sizeof(p));
    p.qkv = (const uint16_t *)a->qkv;
    p.q_bias = (const uint16_t *)a->q_bias;
    p.k_bias = (const uint16_t *)a->k_bias;
    p.v_bias = (const uint16_t *)a->v_bias;
    p.o = (uint16_t *)a->o;
    p.seq_len = (const int32_t *)a->seq_len;
    p.k_cache = (uint16_t *)a->k_cache_table;
    p.v_cache = (uint16_t *)a->v_cache_table;
    p.cos_tab = (const uint16_t *)a->rotary_cos_table;
    p.sin_tab = (const uint16_t *)a->rotary_sin_table;
    char *ws = (char *)a->workspace;
    p.status = (int32_t *)ws;
    const size_t bhs = (size_t)a->batch_size * a->num_heads * S;
    p.part_o = (float *)(ws + kStatusBytes);
    p.part_ml = (float2 *)(ws + kStatusBytes + align_up(bhs * a->head_dim * sizeof(float), 256));
    p.B = a->batch_size;
    p.M = a->memory_max_len;
    p.H = a->num_heads;
    p.Hkv = hkv;
    p.L = a->num_layer;
    p.layer = a->idx_layer;
    p.rot_dim = a->rotary_embedding_dim;
    p.num_splits = S;
    p.qkv_stride = stride;
    if (a->kv_layout == SFA_KV_PAGED) {
        p.kv_row_stride = hd;                   // rows of a page are [page_size, H, D]
        p.kv_head_stride = a->head_dim;
        p.block_table = (const int32_t *)a->block_table;
        p.page_shift = page_shift;
        p.table_stride = a->block_table_stride;
        p.num_pages = a->num_pages;
        p.page_stride = (long long)a->num_layer * a->page_size * hd;
    } else if (a->kv_layout == SFA_KV_BLHMD) {
        p.kv_row_stride = a->head_dim;
        p.kv_head_stride = (long long)a->memory_max_len * a->head_dim;
    } else {
        p.kv_row_stride = hd;
        p.kv_head_stride = a->head_dim;
    }
    const float scale = a->head_dim_inv > 0.f ? a->head_dim_inv : 1.0f / std::sqrt((float)a->head_dim);
    p.scale_log2 = scale * 1.4426950408889634f;
    return launch_decode(p, a->dtype, a->head_dim, (hipStream_t)stream);
}

int sfa_prefill_fwd(const sfa_prefill_args *a, void *stream) {
    if (!a) return fail(SFA_ERR_NULL_POINTER, "sfa_prefill_fwd: args is NULL");
    if (!a->q || !a->k || !a->v || !a->o)
        return fail(SFA_ERR_NULL_POINTER, "sfa_prefill_fwd: q/k/v/o must be non-NULL");
    if (a->batch < 0 || a->heads_q <= 0 || a->heads_kv <= 0 || a->seqlen_q < 0 || a->seqlen_k < 0)
        return fail(SFA_ERR_BAD_SHAPE, "sfa_prefill_fwd: batch=%d heads_q=%d heads_kv=%d seqlen_q=%d seqlen_k=%d",
                    a->batch, a->heads_q, a->heads_kv, a->seqlen_q, a->seqlen_k);
    if (a->heads_q % a->heads_kv)
        return fail(SFA_ERR_BAD_SHAPE, "sfa_prefill_fwd: heads_q=%d not a multiple of heads_kv=%d", a->heads_q, a->heads_kv);
    if (a->head_dim != 64 && a->head_dim != 128 && a->head_dim != 256)
        return fail(SFA_ERR_UNSUPPORTED_HEAD_DIM, "sfa_prefill_fwd: head_dim %d not in {64, 128, 256}", a->head_dim);
    if (a->dtype != SFA_DTYPE_FP16 && a->dtype != SFA_DTYPE_BF16)
        return fail(SFA_ERR_BAD_DTYPE, "sfa_prefill_fwd: dtype %d is not fp16(0)/bf16(1)", a->dtype);
    const int64_t *st[4] = {a->q_stride, a->k_stride, a->v_stride, a->o_stride};
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 3; ++i)
            if (st[t][i] < 0 || (st[t][i] % 8) != 0)
                return fail(SFA_ERR_BAD_SHAPE, "sfa_prefill_fwd: strides must be >= 0 and multiples of 8 elements");
    if (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->o) & 15)
        return fail(SFA_ERR_BAD_SHAPE, "sfa_prefill_fwd: tensors must be 16-byte aligned");
    if ((long long)a->batch * a->heads_q > (1ll << 24))
        return fail(SFA_ERR_BAD_SHAPE, "sfa_prefill_fwd: batch*heads_q too large");
    if (a->batch == 0 || a->seqlen_q == 0) return SFA_OK;

    PrefillKernelParams p;
    memset(&p, 0, sizeof(p));
    p.q = (const uint16_t *)a->q;
    p.k = (const uint16_t *)a->k;
    p.v = (const uint16_t *)a->v;
    p.o = (uint16_t *)a->o;
    p.lse = a->lse;
    p.fast_scale = a->fast_scale != 0 && a->lse == nullptr;
    p.B = a->batch;
    p.Hq = a->heads_q;
    p.Hkv = a->heads_kv;
    p.Sq = a->seqlen_q;
    p.Sk = a->seqlen_k;
    for (int i = 0; i < 3; ++i) {
        p.qs[i] = a->q_stride[i];
        p.ks[i] = a->k_stride[i];
        p.vs[i] = a->v_stride[i];
        p.os[i] = a->o_stride[i];
    }
    const float scale = a->softmax_scale > 0.f ? a->softmax_scale : 1.0f / std::sqrt((float)a->head_dim);
    p.scale_log2 = scale * 1.4426950408889634f;
    if (a->seqlen_k == 0)       // no keys: every row is empty -> zeros, lse = -inf (the header's promise)
        return launch_prefill_no_keys(p, a->head_dim, (hipStream_t)stream);
    p.nq_tiles = (a->seqlen_q + 255) / 256;
    p.bh_per_xcd = (a->batch * a->heads_q + 7) / 8;
    if ((long long)8 * p.bh_per_xcd * p.nq_tiles > 0x7fffffffll)
        return fail(SFA_ERR_BAD_SHAPE, "sfa_prefill_fwd: grid too large");
    return launch_prefill(p, a->dtype, a->head_dim, a->causal != 0, (hipStream_t)stream);
}

int sfa_compute_rotary_table(void *cos_table, void *sin_table, int max_seq_len, int rot_dim, int dtype,
                             void *stream) {
    if (!cos_table || !sin_table) return fail(SFA_ERR_NULL_POINTER, "sfa_compute_rotary_table: NULL table");
    if (max_seq_len < 0 || rot_dim < 0 || (rot_dim & 1))
        return fail(SFA_ERR_BAD_SHAPE, "sfa_compute_rotary_table: max_seq_len=%d rot_dim=%d", max_seq_len, rot_dim);
    return launch_rotary_table(cos_table, sin_table, max_seq_len, rot_dim, dtype, (hipStream_t)stream);
}

int sfa_fill_16bit(void *array, uint16_t bits, size_t n, void *stream) {
    if (!array && n) return fail(SFA_ERR_NULL_POINTER, "sfa_fill_16bit: NULL array");
    return launch_fill16(array, bits, n, (hipStream_t)stream);
}

}  // extern "C"
