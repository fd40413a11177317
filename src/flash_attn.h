// C++ surface of libStarFlashAttention.so, HIP twin of the reference's src/flash_attn.h:7-11
// (same names and argument lists; cudaStream_t -> hipStream_t, half = __half from hip_fp16.h).
// The kernels themselves are not part of the surface any more: everything routes through the C ABI
// in include/star_flash_attn.h.
#pragma once
#include <src/params.h>

// One decode step (fused RoPE + KV append + split-KV attention).  T = half or hip_bfloat16-sized
// 16-bit type; explicit instantiations exist for __half and __hip_bfloat16.
// Asynchronous on `stream`.  Scratch comes from a workspace per (device, stream) owned by the library (grown
// on first use -- do that outside graph capture).  Throws std::runtime_error on bad arguments.
template <typename T>
void run_flash_decoder(Flash_decoder_input &input, Flash_decoder_params &params, hipStream_t stream);

// cos/sin LUT [max_seq_len, rot_embed_dim/2] on the null stream.
template <typename T>
void compute_rotary_table(T *rotary_cos_table, T *rotary_sin_table, int max_seq_len, int rot_embed_dim);

// array[i] = value for i < n (numBlocks / blockSize are accepted for source compatibility and ignored).
void init_half_array(half *array, half value, int n, int numBlocks, int blockSize);

// Synchronises the device and throws if any earlier run_flash_decoder call saw a seq_len[b] outside
// [0, memory_max_len) (the reference silently writes out of bounds there).
void check_flash_decoder_status();
