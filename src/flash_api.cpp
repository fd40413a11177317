// Python extension `star_flash_attn` for PyTorch-ROCm on MI355X.
//
// Same module name, function name, argument names / order and return value as the reference's
// binding (reference src/flash_api.cpp:42-80), so `examples/python/testFlashDecoder.py` drops in:
//     o = star_flash_attn.mha_fwd_cuda(qkv, q_bias, k_bias, v_bias, k_cache_table, v_cache_table,
//                                      seq_len, o, batch_size, memory_max_len, num_heads, head_dim,
//                                      rotary_embedding_dim, max_input_length, num_layer, idx_layer)
// It binds to the hand-written HIP kernels through the C ABI of libStarFlashAttention.so
// (include/star_flash_attn.h); ATen is used for tensor plumbing only (pointers, current stream,
// scratch from the caching allocator).
//
// Differences from the reference binding, all deliberate:
//   * arguments are validated (device, dtype, shape, contiguity) and errors raise RuntimeError;
//   * bf16 is accepted as well as fp16;
//   * biases are honoured (the reference accepts and drops them); an empty tensor means "none";
//   * the call is asynchronous on the current stream -- no device-wide sync, no per-call malloc;
//   * num_splits is chosen by the library (the reference hard-codes 4);
//   * additive entry points: mha_fwd (prefill forward), compute_rotary_table, check_errors,
//     release_workspaces.
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>      // PyTorch-ROCm tensors say "cuda"
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include <cmath>
#include <map>
#include <mutex>
#include <tuple>

#include <include/star_flash_attn.h>
#include <src/params.h>

namespace {

int dtype_code(const at::Tensor &t, const char *name) {
    if (t.scalar_type() == at::kHalf) return SFA_DTYPE_FP16;
    if (t.scalar_type() == at::kBFloat16) return SFA_DTYPE_BF16;
    TORCH_CHECK(false, "star_flash_attn: ", name, " must be float16 or bfloat16, got ", t.scalar_type());
}

void check_tensor(const at::Tensor &t, const char *name, at::ScalarType dt, at::IntArrayRef shape,
                  const at::Device &dev) {
    TORCH_CHECK(t.defined(), "star_flash_attn: ", name, " is undefined");
    TORCH_CHECK(t.is_cuda(), "star_flash_attn: ", name, " must live on a HIP device, got ", t.device());
    TORCH_CHECK(t.device() == dev, "star_flash_attn: ", name, " is on ", t.device(), ", expected ", dev);
    TORCH_CHECK(t.scalar_type() == dt, "star_flash_attn: ", name, " has dtype ", t.scalar_type(), ", expected ", dt);
    TORCH_CHECK(t.sizes() == shape, "star_flash_attn: ", name, " has shape ", t.sizes(), ", expected ", shape);
    TORCH_CHECK(t.is_contiguous(), "star_flash_attn: ", name, " must be contiguous");
}

void check_status(int st, const char *what) {
    TORCH_CHECK(st == SFA_OK, "star_flash_attn: ", what, ": ", sfa_last_error(), " (status ", st, ")");
}

// grow-only scratch per (device, stream), taken from torch's caching allocator
std::mutex g_mu;
std::map<std::pair<int, void *>, at::Tensor> g_workspaces;

at::Tensor &workspace(const at::Device &dev, hipStream_t stream, size_t need) {
    std::lock_guard<std::mutex> lock(g_mu);
    at::Tensor &ws = g_workspaces[{dev.index(), (void *)stream}];
    if (!ws.defined() || (size_t)ws.numel() < need) {
        const int64_t bytes = std::max<int64_t>((int64_t)need, 1 << 20);
        at::Tensor grown = at::empty({bytes}, at::TensorOptions().dtype(at::kByte).device(dev));
        if (ws.defined()) {
            // carry the sticky status block over, in stream order (no sync): a raised flag survives growth
            TORCH_CHECK(hipMemcpyAsync(grown.data_ptr(), ws.data_ptr(), 256, hipMemcpyDeviceToDevice, stream) == hipSuccess,
                        "star_flash_attn: workspace: hipMemcpyAsync failed");
        } else {
            check_status(sfa_decode_reset_status(grown.data_ptr(), stream), "workspace");
        }
        ws = grown;     // the old block goes back to the caching allocator, which keeps it stream-ordered
    }
    return ws;
}

// Drop every cached workspace (they are otherwise kept for the life of the process, one per
// (device, stream) that ever decoded).  Pending sticky flags are lost: call check_errors() first.
void release_workspaces() {
    std::lock_guard<std::mutex> lock(g_mu);
    g_workspaces.clear();
}

const void *opt_ptr(const at::Tensor &t) { return (t.defined() && t.numel() > 0) ? t.data_ptr() : nullptr; }

}  // namespace

// Fill the reference's POD (src/params.h) exactly as its set_input does (reference flash_api.cpp:7-33),
// biases included.
void set_input(Flash_decoder_input &input, at::Tensor qkv, at::Tensor q_bias, at::Tensor k_bias,
               at::Tensor v_bias, at::Tensor o, at::Tensor k_cache_table, at::Tensor v_cache_table,
               at::Tensor seq_len, int batch_size, int memory_max_len, int num_heads, int head_dim,
               int rotary_embedding_dim, int max_input_length, int num_layer, int idx_layer) {
    input.qkv = qkv.data_ptr();
    input.q_bias = const_cast<void *>(opt_ptr(q_bias));
    input.k_bias = const_cast<void *>(opt_ptr(k_bias));
    input.v_bias = const_cast<void *>(opt_ptr(v_bias));
    input.o = o.data_ptr();
    input.seq_len = seq_len.data_ptr();
    input.k_cache_table = k_cache_table.data_ptr();
    input.v_cache_table = v_cache_table.data_ptr();
    input.batch_size = batch_size;
    input.memory_max_len = memory_max_len;
    input.num_heads = num_heads;
    input.head_dim = head_dim;
    input.head_dim_inv = 1.0f / std::sqrt((float)head_dim);
    input.rotary_embedding_dim = rotary_embedding_dim;
    input.max_input_length = max_input_length;
    input.stride = 3 * num_heads * head_dim;
    input.num_layer = num_layer;
    input.idx_layer = idx_layer;
}

void set_default_params(Flash_decoder_params &params) {
    params.kBlockN = 0;         // tile shape is internal to the HIP kernel
    params.num_splits = 0;      // library's choice (sfa_decode_auto_splits)
    params.kNThreads = 256;     // informational: 4 wave64s per workgroup
}

at::Tensor mha_fwd_cuda(at::Tensor &qkv, at::Tensor &q_bias, at::Tensor &k_bias, at::Tensor &v_bias,
                        at::Tensor &k_cache_table, at::Tensor &v_cache_table, at::Tensor &seq_len,
                        at::Tensor &o, int batch_size, int memory_max_len, int num_heads, int head_dim,
                        int rotary_embedding_dim, int max_input_length, int num_layer, int idx_layer) {
    TORCH_CHECK(qkv.defined() && qkv.is_cuda(), "star_flash_attn: qkv must live on a HIP device");
    const int dt = dtype_code(qkv, "qkv");
    const at::ScalarType st = qkv.scalar_type();
    const at::Device dev = qkv.device();
    const int64_t B = batch_size, H = num_heads, D = head_dim, M = memory_max_len, L = num_layer;
    check_tensor(qkv, "qkv", st, {B, 3, H, D}, dev);
    check_tensor(o, "o", st, {B, H, D}, dev);
    check_tensor(seq_len, "seq_len", at::kInt, {B}, dev);
    check_tensor(k_cache_table, "k_cache_table", st, {B, L, M, H, D}, dev);
    check_tensor(v_cache_table, "v_cache_table", st, {B, L, M, H, D}, dev);
    if (opt_ptr(q_bias)) check_tensor(q_bias, "q_bias", st, {H, D}, dev);
    if (opt_ptr(k_bias)) check_tensor(k_bias, "k_bias", st, {H, D}, dev);
    if (opt_ptr(v_bias)) check_tensor(v_bias, "v_bias", st, {H, D}, dev);

    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    Flash_decoder_input input;
    set_input(input, qkv, q_bias, k_bias, v_bias, o, k_cache_table, v_cache_table, seq_len, batch_size,
              memory_max_len, num_heads, head_dim, rotary_embedding_dim, max_input_length, num_layer, idx_layer);
    Flash_decoder_params params;
    set_default_params(params);
    hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();

    sfa_decode_args a = {};             // zero: the reference layout, no paging, num_heads_kv = num_heads
    a.qkv = input.qkv;
    a.q_bias = input.q_bias;
    a.k_bias = input.k_bias;
    a.v_bias = input.v_bias;
    a.o = input.o;
    a.seq_len = input.seq_len;
    a.k_cache_table = input.k_cache_table;
    a.v_cache_table = input.v_cache_table;
    a.rotary_cos_table = input.rotary_cos_table;      // nullptr: angles computed in-kernel
    a.rotary_sin_table = input.rotary_sin_table;
    a.batch_size = input.batch_size;
    a.memory_max_len = input.memory_max_len;
    a.num_heads = input.num_heads;
    a.head_dim = input.head_dim;
    a.head_dim_inv = input.head_dim_inv;
    a.rotary_embedding_dim = input.rotary_embedding_dim;
    a.max_input_length = input.max_input_length;
    a.stride = input.stride;
    a.num_layer = input.num_layer;
    a.idx_layer = input.idx_layer;
    a.num_splits = params.num_splits;
    a.dtype = dt;
    a.kv_layout = SFA_KV_BLMHD;         // the reference's cache layout (src/params.h:22-25)
    const size_t need = sfa_decode_workspace_bytes(batch_size, num_heads, head_dim, memory_max_len, a.num_splits);
    at::Tensor &ws = workspace(dev, stream, need);
    a.workspace = ws.data_ptr();
    a.workspace_bytes = (size_t)ws.numel();
    check_status(sfa_decode(&a, stream), "mha_fwd_cuda");
    return o;
}

// Raise if any earlier mha_fwd_cuda call on the current stream saw seq_len[b] outside
// [0, memory_max_len).  Synchronises that stream.
void check_errors() {
    const int devi = c10::hip::current_device();
    hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(devi).stream();
    at::Tensor ws;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_workspaces.find({devi, (void *)stream});
        if (it == g_workspaces.end()) return;
        ws = it->second;
    }
    const int st = sfa_decode_poll_status(ws.data_ptr(), stream);
    if (st != SFA_OK) (void)sfa_decode_reset_status(ws.data_ptr(), stream);
    check_status(st, "mha_fwd_cuda");
}

// Prefill forward: q [B,Hq,Sq,D], k/v [B,Hkv,Sk,D] (head_dim contiguous, any other strides).
std::vector<at::Tensor> mha_fwd(const at::Tensor &q, const at::Tensor &k, const at::Tensor &v,
                                c10::optional<at::Tensor> out_, bool causal, double softmax_scale,
                                bool return_lse, bool fast_scale) {
    TORCH_CHECK(q.defined() && q.is_cuda(), "star_flash_attn: q must live on a HIP device");
    const int dt = dtype_code(q, "q");
    for (const at::Tensor *t : {&q, &k, &v}) {
        TORCH_CHECK(t->is_cuda() && t->device() == q.device(), "star_flash_attn: q, k, v must share a device");
        TORCH_CHECK(t->scalar_type() == q.scalar_type(), "star_flash_attn: q, k, v must share a dtype");
        TORCH_CHECK(t->dim() == 4 && t->stride(3) == 1, "star_flash_attn: expected [batch, heads, seq, head_dim] with contiguous head_dim");
    }
    TORCH_CHECK(k.sizes() == v.sizes() && k.size(0) == q.size(0) && k.size(3) == q.size(3),
                "star_flash_attn: k/v shapes ", k.sizes(), " / ", v.sizes(), " do not match q ", q.sizes());
    at::Tensor out = out_.has_value() ? *out_ : at::empty_like(q, q.options(), at::MemoryFormat::Contiguous);
    TORCH_CHECK(out.sizes() == q.sizes() && out.scalar_type() == q.scalar_type() && out.device() == q.device() &&
                out.stride(3) == 1, "star_flash_attn: out must match q");
    at::Tensor lse;
    if (return_lse) lse = at::empty({q.size(0), q.size(1), q.size(2)}, q.options().dtype(at::kFloat));

    c10::hip::HIPGuardMasqueradingAsCUDA guard(q.device());
    sfa_prefill_args a = {};
    a.q = q.data_ptr(); a.k = k.data_ptr(); a.v = v.data_ptr(); a.o = out.data_ptr();
    a.lse = return_lse ? lse.data_ptr<float>() : nullptr;
    a.batch = (int)q.size(0); a.heads_q = (int)q.size(1); a.heads_kv = (int)k.size(1);
    a.seqlen_q = (int)q.size(2); a.seqlen_k = (int)k.size(2); a.head_dim = (int)q.size(3);
    for (int i = 0; i < 3; ++i) {
        a.q_stride[i] = q.stride(i); a.k_stride[i] = k.stride(i);
        a.v_stride[i] = v.stride(i); a.o_stride[i] = out.stride(i);
    }
    a.softmax_scale = (float)softmax_scale;
    a.causal = causal ? 1 : 0;
    a.fast_scale = fast_scale ? 1 : 0;      // opt-in: prescaled-Q kernels (include/star_flash_attn.h)
    a.dtype = dt;
    check_status(sfa_prefill_fwd(&a, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(q.device().index()).stream()), "mha_fwd");
    if (return_lse) return {out, lse};
    return {out};
}

std::vector<at::Tensor> compute_rotary_table(int max_seq_len, int rot_dim, at::ScalarType dtype, at::Device device) {
    at::Tensor c = at::empty({max_seq_len, rot_dim / 2}, at::TensorOptions().dtype(dtype).device(device));
    at::Tensor s = at::empty_like(c);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(device);
    check_status(sfa_compute_rotary_table(c.data_ptr(), s.data_ptr(), max_seq_len, rot_dim, dtype_code(c, "table"),
                                          c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(device.index()).stream()),
                 "compute_rotary_table");
    return {c, s};
}

PYBIND11_MODULE(star_flash_attn, m) {
    m.doc() = "Fused flash-decoder / flash-attention forward for AMD MI355X (HIP, gfx950)";

    m.def("mha_fwd_cuda", &mha_fwd_cuda,
          "One decode step: fused RoPE + KV-cache append + split-KV attention (mutates o and the caches, returns o)",
          py::arg("qkv"), py::arg("q_bias"), py::arg("k_bias"), py::arg("v_bias"),
          py::arg("k_cache_table"), py::arg("v_cache_table"), py::arg("seq_len"),
          py::arg("o"), py::arg("batch_size"), py::arg("memory_max_len"),
          py::arg("num_heads"), py::arg("head_dim"), py::arg("rotary_embedding_dim"),
          py::arg("max_input_length"), py::arg("num_layer"), py::arg("idx_layer"));
    m.def("check_errors", &check_errors,
          "Synchronise the current stream and raise if a decode call saw an out-of-range seq_len");
    m.def("release_workspaces", &release_workspaces,
          "Free the cached decode scratch of every (device, stream); call check_errors() first if flags matter");
    m.def("mha_fwd", &mha_fwd, "Attention forward (prefill): returns [out] or [out, lse]",
          py::arg("q"), py::arg("k"), py::arg("v"), py::arg("out") = py::none(), py::arg("causal") = false,
          py::arg("softmax_scale") = 0.0, py::arg("return_lse") = false, py::arg("fast_scale") = false);
    m.def("compute_rotary_table", &compute_rotary_table, "cos/sin LUT [max_seq_len, rot_dim/2]",
          py::arg("max_seq_len"), py::arg("rot_dim"), py::arg("dtype"), py::arg("device"));
}
