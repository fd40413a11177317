// Compile-time geometry of the gfx950 flash decoder.  The reference's src/traits.h sizes CUDA
// shared memory for one 32-lane warp; none of that applies on CDNA4 (no LDS staging at all on the
// decode path), so this header only documents the geometry the HIP kernels use and keeps the
// `Traits<elem, head_dim, blockN>` name alive for code that mentions it.  params.h includes it.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

template <typename elem_type, int head_dim, int n_elem_per_blockN>
struct Traits {
    static constexpr int wave_size = 64;                                // CDNA wavefront
    static constexpr int n_elem_per_vec = 16 / sizeof(elem_type);       // one global_load_dwordx4
    static constexpr int lanes_per_row = head_dim / n_elem_per_vec;     // 16 lanes cover a 128-wide row
    static constexpr int rows_per_load = wave_size / lanes_per_row;     // cache rows per wave instruction
    static constexpr int waves_per_block = 4;
    static constexpr int smemSize = waves_per_block * (head_dim + 2) * (int)sizeof(float);   // merge scratch
    static_assert(head_dim % n_elem_per_vec == 0, "head_dim must be a multiple of 8");
};
