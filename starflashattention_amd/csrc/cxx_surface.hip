// The C++ template surface declared in src/flash_attn.h, implemented over the C ABI.
// (Reference: run_flash_decoder<T> flash_attn.cu:937-1018 allocates, copies and frees on every call
// and synchronises the device; here scratch is a grow-only workspace per (device, stream) -- the split
// partials of two streams never share memory -- and nothing syncs.)
// SFA_ROCTX=1 in the environment AT LOAD TIME wraps every run_flash_decoder call in a roctx range
// (the reference brackets its timed loop with NVTX, examples/cpp/testFlashDecoder.cc:99-106); the
// roctx library is opened with dlopen, so nothing links against it.
#include <dlfcn.h>
#include <hip/hip_bf16.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>

#include <src/flash_attn.h>

#include "sfa_host.h"

namespace {

struct Workspace { void *ptr = nullptr; size_t bytes = 0; };
std::mutex g_mu;
std::map<std::pair<int, hipStream_t>, Workspace> g_ws;      // per (device, stream), as the pybind layer keys its own

// roctx ranges, decided once when the library is loaded
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char *e = std::getenv("SFA_ROCTX");
        if (!e || !std::atoi(e)) return;
        void *h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const Roctx g_roctx;
struct RoctxRange {
    explicit RoctxRange(const char *name) { if (g_roctx.push) g_roctx.push(name); }
    ~RoctxRange() { if (g_roctx.push) g_roctx.pop(); }
};

[[noreturn]] void raise(const char *what) {
    throw std::runtime_error(std::string(what) + ": " + sfa_last_error());
}

Workspace &workspace_for(size_t need, hipStream_t stream) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_mu);
    Workspace &w = g_ws[{dev, stream}];
    if (w.bytes < need) {
        // growing allocates and (when a block exists) synchronises: not legal inside a stream capture --
        // make the first call of a shape outside the capture (src/flash_attn.h)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            throw std::runtime_error("run_flash_decoder: the workspace must grow, which cannot happen during stream "
                                     "capture; call once with this shape before capturing");
        int32_t sticky = 0;
        if (w.ptr) {
            // earlier calls on this stream may still use the old block; carry its sticky status word over
            (void)hipMemcpyAsync(&sticky, w.ptr, sizeof(sticky), hipMemcpyDeviceToHost, stream);
            (void)hipStreamSynchronize(stream);
            (void)hipFree(w.ptr);
        }
        size_t bytes = need < (1u << 20) ? (1u << 20) : need;
        if (hipMalloc(&w.ptr, bytes) != hipSuccess) {
            w = Workspace();
            throw std::runtime_error("run_flash_decoder: hipMalloc of the workspace failed");
        }
        w.bytes = bytes;
        if (sfa_decode_reset_status(w.ptr, stream) != SFA_OK) raise("run_flash_decoder");
        if (sticky) (void)hipMemcpyAsync(w.ptr, &sticky, sizeof(sticky), hipMemcpyHostToDevice, stream),
                    (void)hipStreamSynchronize(stream);
    }
    return w;
}

template <typename T> constexpr int dtype_of() {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    return std::is_same<T, __hip_bfloat16>::value ? SFA_DTYPE_BF16 : SFA_DTYPE_FP16;
}

}  // namespace

template <typename T>
void run_flash_decoder(Flash_decoder_input &in, Flash_decoder_params &params, hipStream_t stream) {
    const RoctxRange range("run_flash_decoder");
    sfa_decode_args a = {};             // zero: the reference layout, no paging, num_heads_kv = num_heads
    a.qkv = in.qkv;
    a.q_bias = in.q_bias;
    a.k_bias = in.k_bias;
    a.v_bias = in.v_bias;
    a.o = in.o;
    a.seq_len = in.seq_len;
    a.k_cache_table = in.k_cache_table;
    a.v_cache_table = in.v_cache_table;
    a.rotary_cos_table = in.rotary_cos_table;
    a.rotary_sin_table = in.rotary_sin_table;
    a.batch_size = in.batch_size;
    a.memory_max_len = in.memory_max_len;
    a.num_heads = in.num_heads;
    a.head_dim = in.head_dim;
    a.head_dim_inv = in.head_dim_inv;
    a.rotary_embedding_dim = in.rotary_embedding_dim;
    a.max_input_length = in.max_input_length;
    a.stride = in.stride;
    a.num_layer = in.num_layer;
    a.idx_layer = in.idx_layer;
    a.num_splits = params.num_splits;
    a.dtype = dtype_of<T>();
    a.kv_layout = SFA_KV_BLMHD;         // the layout params.h documents
    const size_t need = sfa_decode_workspace_bytes(a.batch_size, a.num_heads, a.head_dim,
                                                   a.memory_max_len, a.num_splits);
    Workspace &w = workspace_for(need, stream);
    a.workspace = w.ptr;
    a.workspace_bytes = w.bytes;
    if (sfa_decode(&a, stream) != SFA_OK) raise("run_flash_decoder");
}

template <typename T>
void compute_rotary_table(T *cos_table, T *sin_table, int max_seq_len, int rot_embed_dim) {
    if (sfa_compute_rotary_table(cos_table, sin_table, max_seq_len, rot_embed_dim, dtype_of<T>(), nullptr) != SFA_OK)
        raise("compute_rotary_table");
}

void init_half_array(half *array, half value, int n, int /*numBlocks*/, int /*blockSize*/) {
    uint16_t bits;
    static_assert(sizeof(half) == 2, "half is 16 bit");
    __builtin_memcpy(&bits, &value, 2);
    if (sfa_fill_16bit(array, bits, n < 0 ? 0 : (size_t)n, nullptr) != SFA_OK) raise("init_half_array");
}

void check_flash_decoder_status() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceSynchronize() != hipSuccess) throw std::runtime_error("hipDeviceSynchronize failed");
    std::lock_guard<std::mutex> lock(g_mu);
    int worst = SFA_OK;
    for (auto &kv : g_ws) {                     // every stream's workspace of this device
        if (kv.first.first != dev || !kv.second.ptr) continue;
        const int st = sfa_decode_poll_status(kv.second.ptr, nullptr);
        if (st != SFA_OK) {
            (void)sfa_decode_reset_status(kv.second.ptr, nullptr);
            worst = st;
        }
    }
    if (worst != SFA_OK) {
        throw std::runtime_error("run_flash_decoder: a seq_len[b] was outside [0, memory_max_len) or a block_table "
                                 "entry outside the pool; those outputs are NaN and their cache rows were not written");
    }
}

template void run_flash_decoder<half>(Flash_decoder_input &, Flash_decoder_params &, hipStream_t);
template void run_flash_decoder<__hip_bfloat16>(Flash_decoder_input &, Flash_decoder_params &, hipStream_t);
template void compute_rotary_table<half>(half *, half *, int, int);
template void compute_rotary_table<__hip_bfloat16>(__hip_bfloat16 *, __hip_bfloat16 *, int, int);
