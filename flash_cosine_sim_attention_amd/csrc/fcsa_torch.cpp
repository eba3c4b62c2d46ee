// fcsa_torch.cpp -- compiled PyTorch binding of the C ABI (include/fcsa.h): the counterpart of the reference's pybind module
// (flash_cosine_sim_attention_cuda.cu:1928-1933) and of the tensor handling of its host launchers (cu:1630-1698, cu:1752-1827).
//
// Host-only C++ (no device code): shape canonicalisation, argument checks (TORCH_CHECK -> Python exceptions instead of the
// reference's compiled-out asserts), output / saved-state allocation with ATen, then ONE call into libfcsa_hip.so on the current
// HIP stream of q's device.  Registered as dispatcher ops (TORCH_LIBRARY) so that the Python wrapper is a thin
// autograd.Function over `torch.ops.fcsa.*` -- cheap per call, and traceable by torch.compile (fake kernels are registered in
// Python, flash_cosine_sim_attention_amd/_torch_ops.py).  No torch type crosses into libfcsa_hip.so: the boundary stays the C ABI.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <tuple>

#include "../../include/fcsa.h"

namespace {

// The C ABI is called through pointers so that measurement tools can swap in another build of the same ABI at run time
// (fcsa_torch_use_library, tools/ab_libs.py: interleaved A/B of kernel variants in ONE process).  Default: the library this
// module is linked against.
struct Abi {
  int (*forward)(const fcsa_forward_args*) = &fcsa_forward;
  int (*backward)(const fcsa_backward_args*) = &fcsa_backward;
  size_t (*forward_ws)(const fcsa_problem*) = &fcsa_forward_workspace_bytes;
  size_t (*backward_ws)(const fcsa_problem*) = &fcsa_backward_workspace_bytes;
  int (*needs_qn)(const fcsa_problem*, int32_t) = &fcsa_forward_needs_qn;
  const char* (*last_error)(void) = &fcsa_last_error;
} g_abi;

using at::Tensor;
using c10::optional;

// Host-time accounting of the two ops (tools/host_overhead.py; small problems are bound by host time per call, not by the
// kernels): nanoseconds spent in [0] forward checks / canonicalisation, [1] forward allocations, [2] fcsa_forward (validation +
// launches), [3..5] the same for backward, [6] forward calls, [7] backward calls.  Two clock reads per section, always on.
std::atomic<uint64_t> g_host_ns[8];
struct Lap {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void mark(int slot) {
    const auto n = std::chrono::steady_clock::now();
    g_host_ns[slot].fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count(), std::memory_order_relaxed);
    t = n;
  }
};

int dtype_code(at::ScalarType t) {
  switch (t) {
    case at::kFloat: return FCSA_F32;
    case at::kHalf: return FCSA_F16;
    case at::kBFloat16: return FCSA_BF16;
    default: TORCH_CHECK_TYPE(false, "unsupported dtype ", t, "; expected float32, float16 or bfloat16");
  }
  return -1;
}

// rows 16-byte aligned and the feature dim contiguous: consumed in place (e.g. the `b n (h d) -> b h n d` views of
// transformer.py:100); anything else is made contiguous
bool rows_ok(const Tensor& t) {
  const int64_t m = 16 / t.element_size();
  if (t.stride(-1) != 1 || (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) != 0) return false;
  for (int64_t d = 0; d + 1 < t.dim(); ++d)
    if (t.stride(d) % m != 0) return false;
  return true;
}
Tensor prep(const Tensor& t) { return rows_ok(t) ? t : t.contiguous(); }

fcsa_tensor view4(const Tensor& t) {
  fcsa_tensor v;
  v.ptr = t.data_ptr();
  v.stride0 = t.stride(0);
  v.stride1 = t.stride(1);
  v.stride2 = t.stride(2);
  return v;
}

struct Canon {
  Tensor q, k, v;                 // 4-D, rows ok
  optional<Tensor> mask, bias;    // contiguous
  bool bias_batch, merged;
  int64_t B, H, Hk, N, M, D;
};

Canon canonicalise(const Tensor& q, const Tensor& k, const Tensor& v, const optional<Tensor>& mask, const optional<Tensor>& bias,
                   bool bias_batch, bool causal) {
  TORCH_CHECK(q.is_cuda(), "flash_cosine_sim_attention_amd: q, k, v must be GPU tensors (HIP kernels only, no CPU fallback)");
  auto same_dev = [&](const char* name, const Tensor& t) {
    TORCH_CHECK_VALUE(t.device() == q.device(), name, " is on ", t.device(), " but q is on ", q.device(), ": all tensors must live on q's GPU");
  };
  same_dev("k", k);
  same_dev("v", v);
  if (mask.has_value()) same_dev("mask", *mask);
  if (bias.has_value()) same_dev("attn_bias", *bias);
  TORCH_CHECK_TYPE(q.scalar_type() == k.scalar_type() && q.scalar_type() == v.scalar_type(), "q, k, v must share a dtype, got ",
                   q.scalar_type(), ", ", k.scalar_type(), ", ", v.scalar_type());
  dtype_code(q.scalar_type());
  TORCH_CHECK_VALUE(!(causal && mask.has_value()), "mask should not be supplied if causality is needed");       // fcsa.py:88, cu:1675
  Canon c;
  c.merged = q.dim() == 3;
  c.bias_batch = bias_batch;
  if (c.merged) {
    TORCH_CHECK_VALUE(k.dim() == 3 && v.dim() == 3, "if batch and heads are merged for queries, keys and values must also have 3 dimensions");
    c.bias_batch = true;                                                                                          // cu:1652
    c.q = q.unsqueeze(1);
  } else {
    TORCH_CHECK_VALUE(q.dim() == 4, "q must have 3 or 4 dimensions, got ", q.dim());
    c.q = q;
  }
  c.k = k.dim() == 3 ? k.unsqueeze(1) : k;
  c.v = v.dim() == 3 ? v.unsqueeze(1) : v;
  TORCH_CHECK_VALUE(c.k.dim() == 4 && c.v.dim() == 4, "k and v must have 3 or 4 dimensions");
  c.B = c.q.size(0); c.H = c.q.size(1); c.N = c.q.size(2); c.D = c.q.size(3);
  c.Hk = c.k.size(1); c.M = c.k.size(2);
  TORCH_CHECK_VALUE(c.v.sizes() == c.k.sizes(), "k and v must have the same shape, got ", k.sizes(), " and ", v.sizes());
  TORCH_CHECK_VALUE(c.k.size(3) == c.D, "query, key, value dimensions must be the same");                          // cu:1673
  TORCH_CHECK_VALUE(c.D == 16 || c.D == 32 || c.D == 64 || c.D == 96 || c.D == 128,
                    "only dimensions (16, 32, 64, 96, 128) allowed for now, got ", c.D);                            // cu:1674
  TORCH_CHECK_VALUE(c.k.size(0) == c.B, "batch mismatch between q (", c.B, ") and k/v (", c.k.size(0), ")");
  TORCH_CHECK_VALUE(c.Hk == c.H || c.Hk == 1, "k/v heads must equal q heads (", c.H, ") or be 1 (single-headed key/values), got ", c.Hk);
  if (mask.has_value()) {
    TORCH_CHECK_VALUE(mask->scalar_type() == at::kBool && mask->dim() == 2 && mask->size(0) == c.B && mask->size(1) == c.M,
                      "mask must be a bool tensor of shape (", c.B, ", ", c.M, "), got ", mask->scalar_type(), " ", mask->sizes());
    c.mask = mask->contiguous();
  }
  if (bias.has_value()) {
    const int64_t lead = c.bias_batch ? c.B : c.H;
    TORCH_CHECK_VALUE(bias->dim() == 3 && bias->size(0) == lead && bias->size(1) == c.N && bias->size(2) == c.M,
                      "attn_bias must have shape (", lead, ", ", c.N, ", ", c.M, "), got ", bias->sizes());
    TORCH_CHECK_TYPE(bias->scalar_type() == q.scalar_type(), "attn_bias must have the dtype of q");
    c.bias = bias->contiguous();
  }
  c.q = prep(c.q); c.k = prep(c.k); c.v = prep(c.v);
  return c;
}

fcsa_problem problem(const Canon& c, at::ScalarType dt, bool causal, bool l2norm_qk, int64_t groups, double scale) {
  fcsa_problem p;
  p.dtype = dtype_code(dt);
  p.batch = (int32_t)c.B; p.heads = (int32_t)c.H; p.kv_heads = (int32_t)c.Hk;
  p.q_len = (int32_t)c.N; p.k_len = (int32_t)c.M; p.dim_head = (int32_t)c.D;
  p.causal = causal; p.bias_batch_dim = c.bias_batch; p.l2norm_qk = l2norm_qk;
  p.groups = l2norm_qk ? (int32_t)groups : 1;
  p.scale = (float)scale;
  return p;
}

void check(int rc, const char* what) {
  TORCH_CHECK(rc == FCSA_OK, what, " failed (status ", rc, "): ", g_abi.last_error());
}

void* stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

// (o, inv_l, qn, kn, rq, rk); the saved-state tensors are empty (numel 0) where they are not produced
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> forward(const Tensor& q, const Tensor& k, const Tensor& v, const optional<Tensor>& mask,
                                                                    const optional<Tensor>& attn_bias, bool attn_bias_batch_dim, double scale,
                                                                    bool causal, bool l2norm_qk, int64_t groups, bool need_backward) {
  Lap lap;
  const Canon c = canonicalise(q, k, v, mask, attn_bias, attn_bias_batch_dim, causal);
  TORCH_CHECK_VALUE(!l2norm_qk || (groups >= 1 && c.D % groups == 0), "groups (", groups, ") must divide the head dimension (", c.D, ")");
  c10::DeviceGuard guard(q.device());
  lap.mark(0);
  const auto opt = q.options();
  const auto f32 = opt.dtype(at::kFloat);
  fcsa_forward_args a;
  a.p = problem(c, q.scalar_type(), causal, l2norm_qk, groups, scale);
  Tensor o = at::empty({c.B, c.H, c.N, c.D}, opt);
  Tensor none = at::empty({0}, f32);
  Tensor inv_l = need_backward ? at::empty({c.B, c.H, c.N}, f32) : none;
  Tensor qn = at::empty({0}, opt), kn = qn, rq = none, rk = none;
  if (l2norm_qk) {
    // An inference call (need_backward false) gets kn only -- and qn where q takes the row kernel (fcsa_forward_needs_qn): the
    // 16-bit forward kernels then write nothing but `o` (the reference's need_store_rowsum == false path, cu:1086, cu:1241).
    if (g_abi.needs_qn(&a.p, need_backward ? 1 : 0) != 0) qn = at::empty({c.B, c.H, c.N, c.D}, opt);
    kn = at::empty({c.B, c.Hk, c.M, c.D}, opt);
    if (need_backward) {
      rq = at::empty({c.B, c.H, c.N, groups}, f32);
      rk = at::empty({c.B, c.Hk, c.M, groups}, f32);
    }
  }
  a.q = view4(c.q); a.k = view4(c.k); a.v = view4(c.v); a.o = view4(o);
  a.inv_l = need_backward ? inv_l.data_ptr<float>() : nullptr;
  a.mask = c.mask.has_value() ? static_cast<const uint8_t*>(c.mask->data_ptr()) : nullptr;
  a.attn_bias = c.bias.has_value() ? c.bias->data_ptr() : nullptr;
  a.norm.qn = qn.numel() > 0 ? qn.data_ptr() : nullptr;
  a.norm.kn = l2norm_qk ? kn.data_ptr() : nullptr;
  a.norm.rq = (l2norm_qk && need_backward) ? rq.data_ptr<float>() : nullptr;
  a.norm.rk = (l2norm_qk && need_backward) ? rk.data_ptr<float>() : nullptr;
  Tensor ws;
  a.workspace = nullptr; a.workspace_bytes = 0;
  if (c.B * c.H * c.N < (causal ? 65536 : 32768)) {   // only grids that cannot fill the chip ever split the key range (< 256 row tiles of 128; causal: < 256 PAIRS of them)
    const size_t need = g_abi.forward_ws(&a.p);
    if (need > 0) {
      ws = at::empty({(int64_t)need}, opt.dtype(at::kByte));
      a.workspace = ws.data_ptr(); a.workspace_bytes = need;
    }
  }
  a.stream = stream_of(q);
  lap.mark(1);
  check(g_abi.forward(&a), "fcsa_forward");
  lap.mark(2);
  g_host_ns[6].fetch_add(1, std::memory_order_relaxed);
  if (c.merged) o = o.squeeze(1);                                                                                  // cu:1740-1741
  return std::make_tuple(o, inv_l, qn, kn, rq, rk);
}

// (dq, dk, dv, d_bias) in the shapes / dtype of the inputs; d_bias is empty when not requested
std::tuple<Tensor, Tensor, Tensor, Tensor> backward(const Tensor& d_out, const Tensor& o, const Tensor& inv_l, const Tensor& q, const Tensor& k,
                                                    const Tensor& v, const optional<Tensor>& mask, const optional<Tensor>& attn_bias,
                                                    const Tensor& qn, const Tensor& kn, const Tensor& rq, const Tensor& rk,
                                                    bool attn_bias_batch_dim, double scale, bool causal, bool l2norm_qk, int64_t groups,
                                                    bool need_bias_grad) {
  Lap lap;
  const Canon c = canonicalise(q, k, v, mask, attn_bias, attn_bias_batch_dim, causal);
  c10::DeviceGuard guard(q.device());
  const auto opt = q.options();
  Tensor o4 = prep(o.dim() == 3 ? o.unsqueeze(1) : o);
  Tensor do4 = d_out.dim() == 3 ? d_out.unsqueeze(1) : d_out;
  // A broadcast gradient (`out.sum().backward()`, the reference's own timing protocol, benchmark.py:46-48: one scalar expanded with
  // all strides 0) is not materialised at [B,H,N,D]: one contiguous feature row is, and the kernels read it with a row pitch of 0.
  bool broadcast = do4.numel() > 0;
  for (int64_t d = 0; d < do4.dim(); ++d) broadcast = broadcast && (do4.stride(d) == 0 || do4.size(d) == 1);
  if (broadcast && do4.dim() == 4) {
    Tensor row = do4.as_strided({do4.size(3)}, {0}).to(q.scalar_type()).contiguous();                  // [D], a D-element copy kernel
    do4 = row.as_strided(do4.sizes(), {0, 0, 0, 1});
  }
  if (do4.scalar_type() != q.scalar_type()) do4 = do4.to(q.scalar_type());
  do4 = prep(do4);
  TORCH_CHECK_VALUE(do4.sizes() == o4.sizes(), "d_out must have the shape of the output");
  TORCH_CHECK_VALUE(o4.size(0) == c.B && o4.size(1) == c.H && o4.size(2) == c.N && o4.size(3) == c.D, "o does not belong to these inputs");
  // this op is public (torch.ops.fcsa.backward, ext.backward): everything a kernel dereferences is checked, not only its size
  auto saved_ok = [&](const char* name, const Tensor& t, at::ScalarType st, int64_t numel) {
    TORCH_CHECK_VALUE(t.defined() && t.device() == q.device() && t.scalar_type() == st && t.numel() == numel && t.is_contiguous(),
                      name, " does not belong to these inputs (expected a contiguous ", st, " tensor of ", numel, " elements on ", q.device(), ")");
  };
  TORCH_CHECK_TYPE(o4.scalar_type() == q.scalar_type() && o4.device() == q.device(), "o must have the dtype and device of q");
  TORCH_CHECK_VALUE(do4.device() == q.device(), "d_out is on ", do4.device(), " but q is on ", q.device());
  saved_ok("inv_l", inv_l, at::kFloat, c.B * c.H * c.N);
  if (l2norm_qk) {
    saved_ok("qn", qn, q.scalar_type(), c.B * c.H * c.N * c.D);
    saved_ok("kn", kn, q.scalar_type(), c.B * c.Hk * c.M * c.D);
    saved_ok("rq", rq, at::kFloat, c.B * c.H * c.N * groups);
    saved_ok("rk", rk, at::kFloat, c.B * c.Hk * c.M * groups);
  }
  lap.mark(3);
  Tensor dq = at::empty({c.B, c.H, c.N, c.D}, opt);
  Tensor dk = at::empty({c.B, c.Hk, c.M, c.D}, opt);
  Tensor dv = at::empty({c.B, c.Hk, c.M, c.D}, opt);
  // d_bias in the bias dtype, every element written once by the library: no zero-fill, no f32 tensor, no cast pass (cf. cu:1827, cu:1912)
  Tensor db = (c.bias.has_value() && need_bias_grad) ? at::empty(c.bias->sizes(), opt) : at::empty({0}, opt);
  fcsa_backward_args a;
  a.p = problem(c, q.scalar_type(), causal, l2norm_qk, groups, scale);
  size_t wsb = g_abi.backward_ws(&a.p);
  if (wsb < 256) wsb = 256;
  Tensor ws = at::empty({(int64_t)wsb}, opt.dtype(at::kByte));
  a.d_out = view4(do4); a.o = view4(o4);
  a.inv_l = inv_l.data_ptr<float>();
  a.q = view4(c.q); a.k = view4(c.k); a.v = view4(c.v);
  a.mask = c.mask.has_value() ? static_cast<const uint8_t*>(c.mask->data_ptr()) : nullptr;
  a.attn_bias = c.bias.has_value() ? c.bias->data_ptr() : nullptr;
  a.norm.qn = l2norm_qk ? qn.data_ptr() : nullptr;
  a.norm.kn = l2norm_qk ? kn.data_ptr() : nullptr;
  a.norm.rq = l2norm_qk ? rq.data_ptr<float>() : nullptr;
  a.norm.rk = l2norm_qk ? rk.data_ptr<float>() : nullptr;
  a.dq = view4(dq); a.dk = view4(dk); a.dv = view4(dv);
  a.d_bias = db.numel() > 0 ? db.data_ptr() : nullptr;
  a.workspace = ws.data_ptr(); a.workspace_bytes = wsb;
  a.stream = stream_of(q);
  lap.mark(4);
  check(g_abi.backward(&a), "fcsa_backward");
  lap.mark(5);
  g_host_ns[7].fetch_add(1, std::memory_order_relaxed);
  return std::make_tuple(dq.reshape(q.sizes()), dk.reshape(k.sizes()), dv.reshape(v.sizes()), db);
}

// ---- autograd in C++ (reference: the Python autograd.Function FlashCosineSimAttention, flash_cosine_sim_attention.py:245-302).
// A Python Function costs ~60 us of interpreter / engine hand-over per forward+backward; this node costs a few.  forward and
// backward go through the dispatcher (fcsa::forward / fcsa::backward), so torch.compile traces them with the fake kernels.
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct AttentionFn : public torch::autograd::Function<AttentionFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& q, const Tensor& k, const Tensor& v, const optional<Tensor>& mask,
                        const optional<Tensor>& attn_bias, bool attn_bias_batch_dim, double scale, bool causal, bool l2norm_qk,
                        int64_t groups) {
    // (Function::apply runs this with grad mode OFF: the no_grad case is decided by the caller, attention_autograd)
    const bool bias_grad = attn_bias.has_value() && attn_bias->requires_grad();
    const bool need = q.requires_grad() || k.requires_grad() || v.requires_grad() || bias_grad;                    // cu:1689
    at::AutoDispatchBelowADInplaceOrView guard;
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("fcsa::forward", "")
        .typed<std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const optional<Tensor>&,
                                                                           const optional<Tensor>&, bool, double, bool, bool, int64_t, bool)>();
    auto r = op.call(q, k, v, mask, attn_bias, attn_bias_batch_dim, scale, causal, l2norm_qk, groups, need);
    if (need) {
      ctx->save_for_backward({std::get<0>(r), std::get<1>(r), q, k, v, mask.has_value() ? *mask : Tensor(),
                              attn_bias.has_value() ? *attn_bias : Tensor(), std::get<2>(r), std::get<3>(r), std::get<4>(r), std::get<5>(r)});
      ctx->saved_data["bias_batch"] = attn_bias_batch_dim;
      ctx->saved_data["scale"] = scale;
      ctx->saved_data["causal"] = causal;
      ctx->saved_data["l2norm_qk"] = l2norm_qk;
      ctx->saved_data["groups"] = groups;
      ctx->saved_data["bias_grad"] = bias_grad;
    }
    return std::get<0>(r);
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto s = ctx->get_saved_variables();
    const optional<Tensor> mask = s[5].defined() ? optional<Tensor>(s[5]) : c10::nullopt;
    const optional<Tensor> bias = s[6].defined() ? optional<Tensor>(s[6]) : c10::nullopt;
    const bool bias_grad = ctx->saved_data["bias_grad"].toBool();
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("fcsa::backward", "")
        .typed<std::tuple<Tensor, Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                                                          const optional<Tensor>&, const optional<Tensor>&, const Tensor&, const Tensor&,
                                                          const Tensor&, const Tensor&, bool, double, bool, bool, int64_t, bool)>();
    auto g = op.call(grads[0], s[0], s[1], s[2], s[3], s[4], mask, bias, s[7], s[8], s[9], s[10], ctx->saved_data["bias_batch"].toBool(),
                     ctx->saved_data["scale"].toDouble(), ctx->saved_data["causal"].toBool(), ctx->saved_data["l2norm_qk"].toBool(),
                     ctx->saved_data["groups"].toInt(), bias_grad);
    return {std::get<0>(g), std::get<1>(g), std::get<2>(g), Tensor(), bias_grad ? std::get<3>(g) : Tensor(), Tensor(), Tensor(), Tensor(),
            Tensor(), Tensor()};
  }
};

Tensor attention_autograd(const Tensor& q, const Tensor& k, const Tensor& v, const optional<Tensor>& mask, const optional<Tensor>& attn_bias,
                          bool attn_bias_batch_dim, double scale, bool causal, bool l2norm_qk, int64_t groups) {
  // under torch.no_grad() nothing will ever call backward, whatever the inputs' requires_grad says: take the inference path
  // (no saved state, no inv_l / inverse-norm / normalised-q writes), like a Python Function's ctx.needs_input_grad would
  const bool tracked = at::GradMode::is_enabled() &&
                       (q.requires_grad() || k.requires_grad() || v.requires_grad() || (attn_bias.has_value() && attn_bias->requires_grad()));
  if (!tracked) {
    at::AutoDispatchBelowADInplaceOrView guard;
    return std::get<0>(forward(q, k, v, mask, attn_bias, attn_bias_batch_dim, scale, causal, l2norm_qk, groups, false));
  }
  return AttentionFn::apply(q, k, v, mask, attn_bias, attn_bias_batch_dim, scale, causal, l2norm_qk, groups);
}

// no autograd (inference / inputs that do not require grad): forward without saved state
Tensor attention_plain(const Tensor& q, const Tensor& k, const Tensor& v, const optional<Tensor>& mask, const optional<Tensor>& attn_bias,
                       bool attn_bias_batch_dim, double scale, bool causal, bool l2norm_qk, int64_t groups) {
  return std::get<0>(forward(q, k, v, mask, attn_bias, attn_bias_batch_dim, scale, causal, l2norm_qk, groups, false));
}

}  // namespace

// Measurement hook: read (and reset) the host-time counters above.
extern "C" void fcsa_torch_host_ns(uint64_t* out8) {
  for (int i = 0; i < 8; ++i) out8[i] = g_host_ns[i].exchange(0, std::memory_order_relaxed);
}

// Measurement hook (not part of the operator surface): route the ops to another build of libfcsa_hip.so.  Returns 0 on success.
extern "C" int fcsa_torch_use_library(const char* path) {
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (h == nullptr) return -1;
  Abi a;
  a.forward = reinterpret_cast<decltype(a.forward)>(dlsym(h, "fcsa_forward"));
  a.backward = reinterpret_cast<decltype(a.backward)>(dlsym(h, "fcsa_backward"));
  a.forward_ws = reinterpret_cast<decltype(a.forward_ws)>(dlsym(h, "fcsa_forward_workspace_bytes"));
  a.backward_ws = reinterpret_cast<decltype(a.backward_ws)>(dlsym(h, "fcsa_backward_workspace_bytes"));
  a.needs_qn = reinterpret_cast<decltype(a.needs_qn)>(dlsym(h, "fcsa_forward_needs_qn"));
  a.last_error = reinterpret_cast<decltype(a.last_error)>(dlsym(h, "fcsa_last_error"));
  if (!a.forward || !a.backward || !a.forward_ws || !a.backward_ws || !a.needs_qn || !a.last_error) { dlclose(h); return -2; }
  // Only libraries of THIS ABI: the binding allocates for the struct layouts and buffer contracts of include/fcsa.h as compiled in
  // (e.g. ABI 3 writes d_bias once in the bias dtype into an uninitialised buffer; an ABI-2 library would accumulate float32 into
  // it -- twice the buffer's size for the 16-bit types).  -3: the library reports another version.
  auto dbg = reinterpret_cast<int (*)(char*, size_t)>(dlsym(h, "fcsa_debug"));
  if (!dbg || dbg(nullptr, 0) != FCSA_ABI_VERSION) { dlclose(h); return -3; }
  g_abi = a;
  return 0;
}

TORCH_LIBRARY(fcsa, m) {
  m.def("forward(Tensor q, Tensor k, Tensor v, Tensor? mask, Tensor? attn_bias, bool attn_bias_batch_dim, float scale, bool causal, "
        "bool l2norm_qk, int groups, bool need_backward) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("backward(Tensor d_out, Tensor o, Tensor inv_l, Tensor q, Tensor k, Tensor v, Tensor? mask, Tensor? attn_bias, Tensor qn, Tensor kn, "
        "Tensor rq, Tensor rk, bool attn_bias_batch_dim, float scale, bool causal, bool l2norm_qk, int groups, bool need_bias_grad) "
        "-> (Tensor, Tensor, Tensor, Tensor)");
  // the operator itself: differentiable w.r.t. q, k, v, attn_bias (Autograd kernel below)
  m.def("attention(Tensor q, Tensor k, Tensor v, Tensor? mask, Tensor? attn_bias, bool attn_bias_batch_dim, float scale, bool causal, "
        "bool l2norm_qk, int groups) -> Tensor");
}

TORCH_LIBRARY_IMPL(fcsa, CUDA, m) {       // ROCm builds of PyTorch dispatch HIP tensors under the CUDA key
  m.impl("forward", &forward);
  m.impl("backward", &backward);
  m.impl("attention", &attention_plain);
}

TORCH_LIBRARY_IMPL(fcsa, Autograd, m) {
  m.impl("attention", &attention_autograd);
}
