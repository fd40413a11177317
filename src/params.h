// POD parameter blocks of the flash decoder, HIP build for MI355X (gfx950).
//
// Layout contract: field names, order and types are those of the reference's src/params.h
// (Flash_decoder_input :10-51, Flash_decoder_params :53-58, Flash_decoder_buffers :60-68), so code
// written against the reference -- its pybind binding (src/flash_api.cpp:7-40) and its C++ harness
// (examples/cpp/testFlashDecoder.cc:13-50) -- fills these structs unchanged.  Only the two CUDA
// includes became their HIP twins.  The static_asserts at the bottom pin the ABI.
//
// What each field means HERE (the reference leaves several of them unused or inconsistent,
// SURVEY.md section 8a):
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <src/traits.h>

struct Flash_decoder_input
{
    // new token's packed projections, [batch_size, 3, num_heads, head_dim], 16-bit
    void * __restrict__ qkv = nullptr;
    // optional [num_heads, head_dim] biases added before RoPE (nullptr = none)
    void * __restrict__ q_bias = nullptr;
    void * __restrict__ k_bias = nullptr;
    void * __restrict__ v_bias = nullptr;
    // output, [batch_size, num_heads, head_dim]
    void * __restrict__ o = nullptr;
    // int32 [batch_size]: tokens already in the cache (= RoPE position of the new token)
    void * __restrict__ seq_len = nullptr;

    // KV caches, [batch_size, num_layer, memory_max_len, num_heads, head_dim];
    // row seq_len[b] of layer idx_layer is WRITTEN by the call
    void * __restrict__ k_cache_table = nullptr;
    void * __restrict__ v_cache_table = nullptr;

    // optional cos/sin LUT, [memory_max_len, rotary_embedding_dim/2], 16-bit;
    // nullptr = angles computed in the kernel in fp32
    void * __restrict__ rotary_cos_table = nullptr;
    void * __restrict__ rotary_sin_table = nullptr;

    int batch_size = 0;
    int memory_max_len = 0;
    int num_heads = 0;
    int head_dim = 0;
    // softmax scale, normally 1/sqrt(head_dim)
    float head_dim_inv = 0;
    // leading dims of each head that rotate (even, <= head_dim; 0 = no RoPE)
    int rotary_embedding_dim = 0;
    int max_input_length = 0;
    // elements between consecutive batches of qkv (3 * num_heads * head_dim when packed)
    int stride = 0;
    int num_layer = 0;
    int idx_layer = 0;
};

struct Flash_decoder_params
{
    int kBlockN;     // reference tile height; the HIP kernel picks its own tiling, value ignored
    int num_splits;  // KV splits per (batch, head); <= 0 lets the library choose
    int kNThreads;   // reference block size; ignored (workgroups are 4 wave64s)
};

struct Flash_decoder_buffers
{
    // [batch_size, num_heads, n_split, head_dim] partial outputs (fp32 here)
    void *o_split = nullptr;
    // [batch_size, num_heads, n_split] running sums
    void *ell = nullptr;
    // [batch_size, num_heads, n_split] running maxima
    void *m_formula = nullptr;
};

static_assert(offsetof(Flash_decoder_input, qkv) == 0 && offsetof(Flash_decoder_input, o) == 32 &&
              offsetof(Flash_decoder_input, seq_len) == 40 && offsetof(Flash_decoder_input, k_cache_table) == 48 &&
              offsetof(Flash_decoder_input, rotary_cos_table) == 64 && offsetof(Flash_decoder_input, batch_size) == 80 &&
              offsetof(Flash_decoder_input, head_dim_inv) == 96 && offsetof(Flash_decoder_input, idx_layer) == 116 &&
              sizeof(Flash_decoder_input) == 120, "Flash_decoder_input ABI changed");
static_assert(sizeof(Flash_decoder_params) == 12 && sizeof(Flash_decoder_buffers) == 24, "params ABI changed");
