// fcsa_kernels.h -- host-side launch interface between the C ABI (fcsa_capi.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace fcsa {

// [B, H, L, D] view with BYTE strides; feature dim contiguous.
struct View {
  char*   p;
  int64_t sb, sh, sn;
};

struct FwdParams {
  View q, k, v, o;          // q,k: already normalised (or raw when !l2norm)
  float* inv_l;             // [B,H,N] or nullptr; dyn: log2(1 / sum_j exp(S_ij)), else 1 / max(rowsum, l_eps)
  const uint8_t* mask;      // [B,M] or nullptr
  const char* bias;         // [Hb,N,M] contiguous, element type = dtype, or nullptr
  int B, H, N, M;
  int causal, bias_batch;
  float c1;                 // scale * log2(e)
  float c2;                 // shift * log2(e)      (P~ = exp2(c1 * qk - c2))
  float bias_c;             // log2(e)              (bias enters as bias * log2e)
  float l_eps;              // clamp of the row sum: 1e-10 (cu:83) rescaled by exp(scale - shift)
  int q_scaled;             // 1: q already carries the factor c1 (fused l2norm writes c1 * q^); 0: the kernel applies it
  int dyn;                  // 1: per-row exponent reference, kept online by the kernel (online_recentre); c2 is 0 then
  int q_raw;                // 1 (16-bit types, fusable groups): q is the RAW query; the kernel prologue does its grouped l2norm,
                            //    folds c1 in, and publishes the saved state of the backward:
  char* qn_out;             //    [B,H,N,D] contiguous c1 * q^ (dtype), or nullptr when no backward follows
  float* rq_out;            //    [B,H,N,G] 1 / max(||q_group||, eps), or nullptr when no backward follows
  int G, lgm;               //    groups; log2(group size / 8)
  float norm_eps;
  int splits;               // > 1: the key range is split over gridDim.y workgroups that write un-normalised partials
  float* ws_o;              // [splits][B*H][N][D] f32 partial P~V
  float* ws_l;              // [splits][B*H][N]    f32 partial row sums
};

struct BwdParams {
  View q, k, v, o, d_out;   // q,k normalised
  View dq;                  // [B,H,N,D]  dtype, or f32 slab when dq_f32
  View dk, dv;              // [B,H,M,D]  (per q-head!) dtype or f32 slabs
  int dq_f32, dk_f32, dv_f32;   // element type of the gradient outputs above: 1 = float32
  const float* inv_l;       // [B,H,N]: 1 / rowsum, or log2 of it (invl_log2: the forward ran its per-row-shift form)
  int invl_log2;
  float* delta;             // [B,H,N] scratch: written by the dq kernel, read by the dkv kernel
  const uint8_t* mask;
  const char* bias;
  void* d_bias;             // [Hb,N,M] in the bias dtype, written once per element by bwd_dbias_kernel, or nullptr
  int dq_splits;            // > 1: the dQ kernel splits the KEY range over gridDim.y workgroups that write partial f32 slabs
  int64_t dq_split_stride;  //      byte distance between the slabs of consecutive splits (dq then views slab 0)
  int dkv_splits;           // > 1: the dK/dV kernel splits the QUERY range over gridDim.y workgroups, partial f32 slabs
  int64_t dkv_split_stride; //      byte distance between the dk (and dv) slabs of consecutive splits
  int B, H, N, M;
  int causal, bias_batch;
  float c1, c2, bias_c;
  float scale;
  int q_scaled;             // see FwdParams
  // fused l2norm backward in the epilogues (group size multiple of 8 with a power-of-two number of 8-blocks):
  const float* rq;          // [B,H,N,G] inverse norms of q, or nullptr: dq kernel writes plain dQ^ (dtype or f32 slab)
  const float* rk;          // [B,H,M,G] inverse norms of k, or nullptr (never set for single-headed K/V)
  int G, lgm;               // groups; log2(group size / 8)
  float norm_eps;           // 1e-12
};

struct NormParams {         // grouped l2norm forward:  x -> xn, inv_norm
  View x;                   // [B,H,L,D]
  char* xn;                 // contiguous [B,H,L,D]
  float* inv_norm;          // contiguous [B,H,L,G] or nullptr
  int B, H, L, D, G;
  float eps;
  float out_scale;          // xn = out_scale * x / max(||x||, eps)   (c1 for q in the fused path, else 1)
};

struct NormBwdParams {      // dx = reduce_heads(slab) then (optionally) l2norm backward
  const char* slab;         // [B,HS,L,D] contiguous; element type f32 (slab_f32) or dtype
  int slab_f32;
  int HS;                   // heads in the slab (summed down to HO when HS != HO)
  const char* xn;           // contiguous [B,HO,L,D] normalised input (dtype) or nullptr (no norm)
  const float* inv_norm;    // [B,HO,L,G]
  View dx;                  // [B,HO,L,D] out (dtype)
  int B, HO, L, D, G;
  float eps;
  float xn_scale;           // x^ = xn_scale * xn   (1/c1 when xn was written with out_scale = c1)
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel: raise it once per (instantiation, device).
// `done` is the instantiation's bit mask of devices that have it (one static per launcher); thread safe, idempotent.
template <typename K>
static inline hipError_t ensure_dynamic_lds(K kern, size_t lds, std::atomic<uint64_t>& done) {
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  const uint64_t bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;        // devices >= 64: set it on every launch
  if (bit != 0 && (done.load(std::memory_order_acquire) & bit) != 0) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess && bit != 0) done.fetch_or(bit, std::memory_order_release);
  return e;
}

// Compute units of the CURRENT device: what every "does this grid cover the chip" threshold of the launchers and of the C ABI's split
// rules is a multiple of.  Cached per device id (a process may drive unlike devices, e.g. partitioned CPX modes); a query that fails is
// answered with 256 and NOT remembered (host-only callers: workspace-size queries in the CPU tests; a process that asks before its
// device is usable gets the real number the next time).  The workspace size a caller is told and the launch that uses it agree as long
// as both run with the same current device.
inline int cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();      // a host without a device: do not leave a sticky error behind
    return 256;
  }
  const bool slot = dev >= 0 && dev < 64;
  if (slot) {
    const int n = cached[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
  }
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
    (void)hipGetLastError();
    return 256;
  }
  if (slot) cached[dev].store(cus, std::memory_order_relaxed);
  return cus;
}

// dtype: 1 = f16, 2 = bf16 (fcsa_dtype); returns hipError_t of the launch
hipError_t launch_forward(int dtype, int D, const FwdParams& p, hipStream_t s);
// fcsa_fwd3.hip: the 64-rows-per-wave, one-wave-per-SIMD forward for 16-bit D = 128 (launch_forward dispatches to it)
bool use_forward_wide128(int dtype, int D, const FwdParams& p);
int forward_wide128_mode(int set);      // debug knob behind fcsa_debug_forward_form: set < 0 queries; returns the previous value
hipError_t launch_forward_wide128(int dtype, const FwdParams& p, hipStream_t s);
hipError_t launch_backward_dq(int dtype, int D, const BwdParams& p, hipStream_t s);
hipError_t launch_backward_dbias(int dtype, int D, const BwdParams& p, hipStream_t s);   // d_bias from recomputed dS tiles (after dq: needs delta)
hipError_t launch_backward_dkv(int dtype, int D, const BwdParams& p, hipStream_t s);
hipError_t launch_l2norm(int dtype, const NormParams& p, hipStream_t s);
hipError_t launch_l2norm_pair(int dtype, const NormParams& a, const NormParams& b, hipStream_t s);   // q and k in one grid
hipError_t launch_l2norm_bwd(int dtype, const NormBwdParams& p, hipStream_t s);
hipError_t launch_l2norm_bwd_pair(int dtype, const NormBwdParams& a, const NormBwdParams& b, hipStream_t s);   // two passes, one grid
hipError_t launch_l2norm_bwd_triple(int dtype, const NormBwdParams& a, const NormBwdParams& b, const NormBwdParams& c, hipStream_t s);   // three

}  // namespace fcsa
