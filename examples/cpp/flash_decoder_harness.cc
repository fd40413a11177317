// Torch-free harness for the C++ surface (src/flash_attn.h) on MI355X.
// Exercises the same scenario as the reference's examples/cpp/testFlashDecoder.cc (B=2, H=32, D=128,
// 4 layers, all-ones tensors, six (max_seq_len, seq_len) pairs, 100 warm-up calls) but checks the
// known answer instead of printing it, times the steady state with HIP events, and shows that the
// reference's out-of-range pair (4096, 4096) is rejected instead of overrunning the cache.
//
// The timed iterations sit inside a roctx range, as the reference brackets its own with NVTX
// (examples/cpp/testFlashDecoder.cc:7,99,106): `rocprofv3 --marker-trace -- build/flash_decoder_harness`.
//
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -I. examples/cpp/flash_decoder_harness.cc \
//         -Lstarflashattention_amd/lib -lStarFlashAttention -L/opt/rocm/lib -lroctx64 -o build/flash_decoder_harness
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <roctracer/roctx.h>

#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <vector>

#include <src/flash_attn.h>

#define HIP_OK(x)                                                                          \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

static int run_case(int B, int H, int D, int M, int seq, int L, int layer, int splits, bool expect_reject) {
    Flash_decoder_input in;
    Flash_decoder_params prm;
    prm.kBlockN = 128;
    prm.num_splits = splits;
    prm.kNThreads = 32;
    in.batch_size = B; in.num_heads = H; in.head_dim = D;
    in.head_dim_inv = 1.0f / std::sqrt((float)D);
    in.memory_max_len = M; in.max_input_length = M;
    in.rotary_embedding_dim = D;
    in.stride = 3 * H * D;
    in.idx_layer = layer; in.num_layer = L;

    const size_t n_qkv = (size_t)B * 3 * H * D, n_o = (size_t)B * H * D;
    const size_t n_cache = (size_t)B * L * M * H * D;
    HIP_OK(hipMalloc((void **)&in.qkv, n_qkv * sizeof(half)));
    HIP_OK(hipMalloc((void **)&in.o, n_o * sizeof(half)));
    HIP_OK(hipMalloc((void **)&in.k_cache_table, n_cache * sizeof(half)));
    HIP_OK(hipMalloc((void **)&in.v_cache_table, n_cache * sizeof(half)));
    HIP_OK(hipMalloc((void **)&in.seq_len, B * sizeof(int)));
    HIP_OK(hipMalloc((void **)&in.rotary_cos_table, (size_t)M * (D / 2) * sizeof(half)));
    HIP_OK(hipMalloc((void **)&in.rotary_sin_table, (size_t)M * (D / 2) * sizeof(half)));
    compute_rotary_table<half>((half *)in.rotary_cos_table, (half *)in.rotary_sin_table, M, D);

    std::vector<int> lens(B, seq);
    HIP_OK(hipMemcpy(in.seq_len, lens.data(), B * sizeof(int), hipMemcpyHostToDevice));
    const half one = __float2half(1.0f);
    init_half_array((half *)in.qkv, one, (int)n_qkv, 0, 0);
    init_half_array((half *)in.k_cache_table, one, (int)n_cache, 0, 0);
    init_half_array((half *)in.v_cache_table, one, (int)n_cache, 0, 0);
    init_half_array((half *)in.o, __float2half(0.0f), (int)n_o, 0, 0);

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    HIP_OK(hipDeviceSynchronize());
    for (int i = 0; i < 100; ++i) run_flash_decoder<half>(in, prm, stream);
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    const int iters = 50;
    HIP_OK(hipEventRecord(e0, stream));
    roctxRangePushA("run_flash_decoder Range");
    for (int i = 0; i < iters; ++i) run_flash_decoder<half>(in, prm, stream);
    roctxRangePop();
    HIP_OK(hipEventRecord(e1, stream));
    HIP_OK(hipStreamSynchronize(stream));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));

    bool rejected = false;
    try {
        check_flash_decoder_status();
    } catch (const std::runtime_error &e) {
        rejected = true;
        if (!expect_reject) fprintf(stderr, "unexpected: %s\n", e.what());
    }

    std::vector<half> o(n_o);
    HIP_OK(hipMemcpy(o.data(), in.o, n_o * sizeof(half), hipMemcpyDeviceToHost));
    int bad = 0;
    if (!expect_reject)
        for (size_t i = 0; i < n_o; ++i)
            if (std::fabs(__half2float(o[i]) - 1.0f) > 1e-3f) ++bad;
    const double bytes = 2.0 * B * (double)(seq + 1) * H * D * sizeof(half);
    printf("max_seq_len=%5d seq_len=%5d splits=%d  %8.2f us/call  %7.1f GB/s  first=%.3f last=%.3f  %s\n",
           M, seq, splits, 1e3 * ms / iters, bytes / (ms / iters * 1e-3) / 1e9, __half2float(o[0]),
           __half2float(o[n_o - 1]),
           expect_reject ? (rejected ? "REJECTED (as it must be)" : "NOT REJECTED") : (bad ? "WRONG" : "ok"));

    HIP_OK(hipStreamDestroy(stream));
    for (void *ptr : {in.qkv, in.o, in.k_cache_table, in.v_cache_table, in.seq_len, in.rotary_cos_table,
                      in.rotary_sin_table})
        HIP_OK(hipFree(ptr));
    if (expect_reject) return rejected ? 0 : 1;
    return (bad || rejected) ? 1 : 0;
}

// Two streams decoding different problems of the same shape at once, num_splits > 1: each stream's split
// partials must live in its own scratch.  Problem A is all ones (answer 1.0); problem B has qkv = 2 and a
// V cache of 2 (every value row is 2, so the answer is 2.0 whatever the softmax weights are).
static int run_two_streams() {
    const int B = 2, H = 32, D = 128, M = 2048, seq = 2047, L = 1, iters = 200;
    Flash_decoder_input in[2];
    Flash_decoder_params prm;
    prm.kBlockN = 128; prm.num_splits = 4; prm.kNThreads = 32;
    hipStream_t st[2];
    const size_t n_qkv = (size_t)B * 3 * H * D, n_o = (size_t)B * H * D, n_cache = (size_t)B * L * M * H * D;
    std::vector<int> lens(B, seq);
    for (int i = 0; i < 2; ++i) {
        Flash_decoder_input &x = in[i];
        x.batch_size = B; x.num_heads = H; x.head_dim = D; x.head_dim_inv = 1.0f / std::sqrt((float)D);
        x.memory_max_len = M; x.max_input_length = M; x.rotary_embedding_dim = D; x.stride = 3 * H * D;
        x.idx_layer = 0; x.num_layer = L;
        x.rotary_cos_table = nullptr; x.rotary_sin_table = nullptr;
        HIP_OK(hipMalloc((void **)&x.qkv, n_qkv * sizeof(half)));
        HIP_OK(hipMalloc((void **)&x.o, n_o * sizeof(half)));
        HIP_OK(hipMalloc((void **)&x.k_cache_table, n_cache * sizeof(half)));
        HIP_OK(hipMalloc((void **)&x.v_cache_table, n_cache * sizeof(half)));
        HIP_OK(hipMalloc((void **)&x.seq_len, B * sizeof(int)));
        HIP_OK(hipMemcpy(x.seq_len, lens.data(), B * sizeof(int), hipMemcpyHostToDevice));
        const half val = __float2half(i == 0 ? 1.0f : 2.0f);
        init_half_array((half *)x.qkv, val, (int)n_qkv, 0, 0);
        init_half_array((half *)x.k_cache_table, __float2half(1.0f), (int)n_cache, 0, 0);
        init_half_array((half *)x.v_cache_table, val, (int)n_cache, 0, 0);
        init_half_array((half *)x.o, __float2half(0.0f), (int)n_o, 0, 0);
        HIP_OK(hipStreamCreate(&st[i]));
    }
    HIP_OK(hipDeviceSynchronize());
    for (int it = 0; it < iters; ++it)
        for (int i = 0; i < 2; ++i) run_flash_decoder<half>(in[i], prm, st[i]);
    HIP_OK(hipDeviceSynchronize());
    int bad = 0;
    for (int i = 0; i < 2; ++i) {
        std::vector<half> o(n_o);
        HIP_OK(hipMemcpy(o.data(), in[i].o, n_o * sizeof(half), hipMemcpyDeviceToHost));
        for (size_t j = 0; j < n_o; ++j)
            if (std::fabs(__half2float(o[j]) - (i == 0 ? 1.0f : 2.0f)) > 2e-3f) ++bad;
        HIP_OK(hipStreamDestroy(st[i]));
        for (void *ptr : {in[i].qkv, in[i].o, in[i].k_cache_table, in[i].v_cache_table, in[i].seq_len}) HIP_OK(hipFree(ptr));
    }
    printf("two streams, 4 splits, %d interleaved calls each: %s\n", iters, bad ? "WRONG (streams share scratch?)" : "ok");
    return bad ? 1 : 0;
}

int main() {
    const int B = 2, H = 32, D = 128, L = 4, layer = 0, splits = 4;
    const int max_seq_len[6] = {512, 1024, 2048, 4096, 8192, 8192};
    const int seq_len[6] = {511, 1023, 2047, 4096, 6143, 8191};
    int rc = 0;
    for (int i = 0; i < 6; ++i)
        rc |= run_case(B, H, D, max_seq_len[i], seq_len[i], L, layer, splits, seq_len[i] >= max_seq_len[i]);
    rc |= run_case(B, H, D, 8192, 8191, L, layer, 0, false);       // library-chosen split count
    rc |= run_two_streams();
    printf(rc ? "FAILED\n" : "all cases ok\n");
    return rc;
}
