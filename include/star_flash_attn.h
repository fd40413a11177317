/*
 * star_flash_attn.h -- C ABI of libStarFlashAttention.so (MI355X / gfx950 build).
 *
 * This is the drop-in boundary for the reference's one hot path (fused decode
 * attention) plus the prefill forward the north-star adds.  Plain pointers and
 * sizes only: no torch / ATen types, no C++ in the signatures.  Every pointer
 * is a DEVICE pointer unless it says "host".  `stream` is a hipStream_t passed
 * as void* (NULL = the null stream).  All entry points are asynchronous on
 * `stream`, allocate nothing, never synchronise the device and are safe to
 * capture in a hipGraph.  They return SFA_OK or a negative sfa_status;
 * sfa_last_error() gives the message for the calling thread.
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   sfa_decode                  <- run_flash_decoder<T>      src/flash_attn.h:7-8,  src/flash_attn.cu:937-1018
 *                                  (+ the two kernels it launches, cu:554-935)
 *   sfa_decode_args             <- Flash_decoder_input       src/params.h:10-51
 *                                  Flash_decoder_params      src/params.h:53-58
 *                                  Flash_decoder_buffers     src/params.h:60-68 (now `workspace`)
 *   sfa_compute_rotary_table    <- compute_rotary_table<T>   src/flash_attn.h:9-10, cu:512-538
 *   sfa_fill_16bit              <- init_half_array           src/flash_attn.h:11,   cu:493-510
 *   sfa_prefill_fwd             <- (no reference function; BASELINE.json configs 2,3,5)
 * The Python-facing mha_fwd_cuda (src/flash_api.cpp:42-68) and the C++ template
 * surface (src/flash_attn.h) in this repo are thin layers over these symbols.
 */
#ifndef STAR_FLASH_ATTN_C_API_H_
#define STAR_FLASH_ATTN_C_API_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct below changes layout or an entry point changes meaning:
 *   2  kv_layout, fast_scale
 *   3  page_size / block_table / block_table_stride / num_pages / num_heads_kv appended to sfa_decode_args;
 *      a block_table entry outside the pool on the APPEND page now rejects the sequence; sfa_debug_set
 *   4  sfa_debug_get, sfa_decode_workspace_bytes_gqa added (nothing changed) */
#define SFA_ABI_VERSION 4

typedef enum sfa_status {
    SFA_OK = 0,
    SFA_ERR_NULL_POINTER = -1,   /* a required pointer is NULL                       */
    SFA_ERR_BAD_SHAPE = -2,      /* sizes/strides out of the supported range         */
    SFA_ERR_BAD_DTYPE = -3,
    SFA_ERR_UNSUPPORTED_HEAD_DIM = -4,
    SFA_ERR_WORKSPACE_TOO_SMALL = -5,
    SFA_ERR_LAUNCH = -6,         /* HIP reported a launch/runtime failure            */
    SFA_ERR_SEQ_LEN_RANGE = -7,  /* reported by sfa_decode_poll_status: some
                                    seq_len[b] was outside [0, memory_max_len)       */
    SFA_ERR_BLOCK_TABLE_RANGE = -8 /* reported by sfa_decode_poll_status: a block_table
                                    entry was outside [0, num_pages): nothing was stored through
                                    it and the outputs that depended on it are NaN            */
} sfa_status;

typedef enum sfa_dtype {
    SFA_DTYPE_FP16 = 0,          /* IEEE half  (the reference's only dtype)          */
    SFA_DTYPE_BF16 = 1
} sfa_dtype;

/* How the KV caches are laid out.  SFA_KV_BLMHD is the reference's documented layout
 * (src/params.h:22-25, examples/python/testFlashDecoder.py:115-116) and what mha_fwd_cuda /
 * run_flash_decoder use.  SFA_KV_BLHMD keeps every (batch, layer, head) contiguous -- the
 * head-major layout the reference's kernel indexes internally (src/flash_attn.cu:617-619) --
 * so one (b,h) streams a dense 2*M*D-byte region instead of 2*D-byte segments H*D*2 bytes apart. */
typedef enum sfa_kv_layout {
    SFA_KV_BLMHD = 0,            /* [batch, num_layer, memory_max_len, num_heads, head_dim] */
    SFA_KV_BLHMD = 1,            /* [batch, num_layer, num_heads, memory_max_len, head_dim] */
    SFA_KV_PAGED = 2             /* page pools [num_pages, num_layer, page_size, num_heads, head_dim]
                                    addressed through block_table (the reference only NAMES its
                                    cache pointers k_cache_table / v_cache_table, src/params.h:22-25):
                                    token t of sequence b lives in page block_table[b][t / page_size],
                                    row t % page_size.  memory_max_len = the capacity of one sequence
                                    (<= block_table_stride * page_size).                            */
} sfa_kv_layout;

/* ---- library ------------------------------------------------------------------ */
int sfa_abi_version(void);
const char *sfa_status_string(int status);
const char *sfa_last_error(void);            /* thread-local, never NULL            */

/* ---- decode: one new token per sequence, fused RoPE + KV append + split-KV ----- */
/*
 * Field order, names and meaning of the first block mirror Flash_decoder_input
 * (src/params.h:10-51) so a binding can fill it 1:1; the trailing fields replace
 * Flash_decoder_params / Flash_decoder_buffers.
 *
 * Semantics, for every (b, h):
 *   pos   = seq_len[b]                      tokens already cached, 0 <= pos < memory_max_len
 *   q,k,v = qkv[b, {0,1,2}, h, :] (+ bias)  16-bit, `stride` elements between batches
 *   q,k   = rope(q,k; pos)                  interleaved pairs (x[2j], x[2j+1]),
 *                                           angle pos * 10000^(-2j/rot_dim) for 2j < rot_dim,
 *                                           fp32 math, result rounded to the 16-bit dtype
 *   k_cache[b, idx_layer, pos, h, :] = k;  v_cache[b, idx_layer, pos, h, :] = v
 *   o[b, h, :] = softmax(q . K[0..pos]^T * head_dim_inv) . V[0..pos]     (fp32 accumulate)
 * Caches are [batch, num_layer, memory_max_len, num_heads, head_dim], contiguous
 * (kv_layout = SFA_KV_BLMHD), or head-major with kv_layout = SFA_KV_BLHMD.
 * seq_len is NOT incremented (caller's job, as in the reference).
 * A sequence whose seq_len is out of range -- or, with paged caches, whose
 * block_table entry for the page the new token goes to lies outside [0, num_pages) --
 * is left untouched in the caches, gets NaN in o[b] and raises the sticky status word
 * (see sfa_decode_poll_status).  A bad entry on a page that is only READ is not
 * dereferenced either (page 0 is read in its place), raises the same status and turns the
 * outputs of that sequence's affected heads into NaN.
 */
typedef struct sfa_decode_args {
    void *qkv;                      /* [batch, 3, num_heads, head_dim]                */
    const void *q_bias;             /* [num_heads, head_dim] or NULL                  */
    const void *k_bias;             /* [num_heads, head_dim] or NULL                  */
    const void *v_bias;             /* [num_heads, head_dim] or NULL                  */
    void *o;                        /* [batch, num_heads, head_dim]                   */
    const void *seq_len;            /* int32 [batch]                                  */
    void *k_cache_table;            /* see above; written at one row per sequence     */
    void *v_cache_table;
    const void *rotary_cos_table;   /* [memory_max_len, rot_dim/2] 16-bit, or NULL:   */
    const void *rotary_sin_table;   /*   NULL => cos/sin computed in-kernel in fp32   */
    int batch_size;
    int memory_max_len;
    int num_heads;
    int head_dim;                   /* 64, 128 or 256                                 */
    float head_dim_inv;             /* softmax scale; <= 0 => 1/sqrt(head_dim)        */
    int rotary_embedding_dim;       /* even, 0..head_dim                              */
    int max_input_length;           /* carried for interface parity; unused           */
    int stride;                     /* elements between qkv batches; 0 => 3*H*D       */
    int num_layer;
    int idx_layer;
    /* -- replaces Flash_decoder_params (kBlockN / kNThreads are internal now) -- */
    int num_splits;                 /* KV splits per (b,h); <= 0 => chosen by the library */
    int dtype;                      /* sfa_dtype                                      */
    /* -- replaces Flash_decoder_buffers -- */
    void *workspace;                /* >= sfa_decode_workspace_bytes(...), 256-B aligned */
    size_t workspace_bytes;
    int kv_layout;                  /* sfa_kv_layout; 0 = the reference's layout (ABI v2) */
    /* -- SFA_KV_PAGED only -- */
    int page_size;                  /* tokens per page: a power of two >= 16             */
    const void *block_table;        /* int32 [batch, block_table_stride] page numbers    */
    int block_table_stride;         /* entries per sequence, >= ceil(memory_max_len / page_size) */
    int num_pages;                  /* pages in each pool (bounds the table entries)     */
    /* -- grouped queries (no counterpart in the reference; 0 = num_heads) -- */
    int num_heads_kv;               /* kv heads; num_heads / num_heads_kv in {1, 2, 4, 8, 16}.  With
                                       num_heads_kv != num_heads: qkv is [batch, num_heads +
                                       2*num_heads_kv, head_dim] (q heads, k heads, v heads; default
                                       stride (num_heads + 2*num_heads_kv)*head_dim), k_bias / v_bias
                                       and the caches carry num_heads_kv heads, o and q_bias num_heads */
} sfa_decode_args;

/* Bytes of scratch sfa_decode needs for this shape (num_splits <= 0: the library's choice
 * for this shape, which is what sfa_decode will then use). Never 0: the first 256 bytes hold
 * the status word.  Grouped queries (num_heads_kv != num_heads): the library sizes its split
 * count by the KV-head count (one workgroup serves a whole group), so ask
 * sfa_decode_auto_splits(batch_size, num_heads_kv, ...) and pass that count here with num_heads. */
size_t sfa_decode_workspace_bytes(int batch_size, int num_heads, int head_dim,
                                  int memory_max_len, int num_splits);
/* The same for grouped queries: with num_splits <= 0 it sizes for the split count sfa_decode picks from the KV-head
 * count.  (A workspace sized with sfa_decode_workspace_bytes(..., 0) still works: sfa_decode then takes the largest
 * split count that fits it.) */
size_t sfa_decode_workspace_bytes_gqa(int batch_size, int num_heads, int num_heads_kv, int head_dim,
                                      int memory_max_len, int num_splits);
/* The split count the library picks for num_splits <= 0; num_heads = the KV-head count. */
int sfa_decode_auto_splits(int batch_size, int num_heads, int head_dim, int memory_max_len);
/* Zero the sticky status word (async). Call once after allocating a workspace. */
int sfa_decode_reset_status(void *workspace, void *stream);
/* Synchronises `stream`, reads the status word: SFA_OK, SFA_ERR_SEQ_LEN_RANGE or
 * SFA_ERR_BLOCK_TABLE_RANGE. */
int sfa_decode_poll_status(const void *workspace, void *stream);
int sfa_decode(const sfa_decode_args *args, void *stream);

/* ---- prefill: O = softmax(mask(Q K^T * scale)) V ------------------------------- */
/*
 * q [batch, heads_q, seqlen_q, head_dim], k/v [batch, heads_kv, seqlen_k, head_dim],
 * o like q; any batch/head/seq strides (in elements), head_dim contiguous, 16-byte
 * aligned rows.  heads_q % heads_kv == 0 (GQA: query head h reads kv head
 * h / (heads_q/heads_kv)).  causal != 0: key j visible to query i iff
 * j <= i + (seqlen_k - seqlen_q).  lse (optional, fp32 [batch, heads_q, seqlen_q],
 * contiguous) receives log(sum(exp(scaled scores))) per row.  A row with no visible
 * key gets zeros (lse = -inf).
 */
typedef struct sfa_prefill_args {
    const void *q;
    const void *k;
    const void *v;
    void *o;
    float *lse;                     /* may be NULL                                    */
    int batch;
    int heads_q;
    int heads_kv;
    int seqlen_q;
    int seqlen_k;
    int head_dim;                   /* 64, 128 or 256                                 */
    int64_t q_stride[3];            /* {batch, head, seq} strides in elements         */
    int64_t k_stride[3];
    int64_t v_stride[3];
    int64_t o_stride[3];
    float softmax_scale;            /* <= 0 => 1/sqrt(head_dim)                       */
    int causal;
    int dtype;                      /* sfa_dtype                                      */
    int fast_scale;                 /* 0 (default): scores are the fp32 Q.K^T times the scale in fp32.
                                       nonzero, and only when lse == NULL: allow the prescaled-Q kernels
                                       (Q * scale * log2(e) rounded to the 16-bit dtype once, ~5 % faster):
                                       every score then carries a relative error of up to 2^-9 (bf16) /
                                       2^-12 (fp16) of the |q_i k_i| it sums -- invisible for unit-variance
                                       data, several per cent of a softmax weight once logits reach
                                       hundreds (ABI v2)                                                */
} sfa_prefill_args;

int sfa_prefill_fwd(const sfa_prefill_args *args, void *stream);

/* ---- test / A-B hooks: NOT part of the drop-in surface ------------------------------ */
/* The launch paths read no environment variable; the test-suite and tools/ select kernel
 * variants through this call.  knob: "prefill_impl" (-1 auto; kernel generation, see
 * csrc/prefill_dispatch.hip), "prefill_pairs" (1/2), "decode_nt" (0/1), "decode_gqa_mfma" (0/1),
 * "bm128_one_wg" (0/1).  value -1 = the library's own choice.  Process-wide. */
int sfa_debug_set(const char *knob, int value);
/* Read back: the knobs above, and "last_prefill_kernel" = which kernel the last sfa_prefill_fwd of this process
 * launched (the dispatcher's choice included): 1 / 3 the 8-wave 256-row kernel (exact / prescaled), 20 + f the 128-row
 * geometry, 40 + f the 4-wave persistent kernel (f: 1 prescaled, 2 exact), 60 / 61 the head_dim 256 kernels, -1 none
 * yet.  Unknown knob: INT_MIN.  (bench.py names its roofline kernel from this.) */
int sfa_debug_get(const char *knob);

/* ---- small helpers the reference's C++ harness uses ----------------------------- */
/* cos/sin LUT, [max_seq_len, rot_dim/2] each, entry (pos, j) = cos/sin(pos * 10000^(-2j/rot_dim)). */
int sfa_compute_rotary_table(void *cos_table, void *sin_table, int max_seq_len,
                             int rot_dim, int dtype, void *stream);
/* array[i] = bits for i < n (16-bit elements). */
int sfa_fill_16bit(void *array, uint16_t bits, size_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* STAR_FLASH_ATTN_C_API_H_ */
