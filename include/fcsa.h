/*
 * fcsa.h -- C ABI of the MI355X (gfx950) fused cosine-similarity attention library
 *           (libfcsa_hip.so).
 *
 * This is the drop-in boundary for the hot path of lucidrains/flash-cosine-sim-attention:
 * it exports what the reference's native extension exports through pybind
 *   forward  (flash_cosine_sim_attention_cuda.cu:1630-1748, bound at cu:1928-1933)
 *   backward (flash_cosine_sim_attention_cuda.cu:1752-1917)
 *   debug    (cu:1928-1933)
 * as plain `extern "C"` functions over raw device pointers, element strides and sizes.
 * No torch / ATen types cross this boundary.  The caller owns every buffer (inputs,
 * outputs, saved state, workspace) and passes the HIP stream to launch on; the library
 * never allocates, never synchronises the device and never touches the default stream
 * unless `stream` is NULL (reference: default stream + cudaDeviceSynchronize after every
 * call, cu:1720, cu:1745, cu:1889).
 *
 * Layout conventions
 *   q, o, d_out, dq        : [B, H, N, D]          (reference accessor order, cu:30-35)
 *   k, v, dk, dv           : [B, Hk, M, D], Hk == H, or Hk == 1 for single-headed key/values
 *                            (reference: 3-D k/v unsqueezed at cu:1656-1660, is_single_head_kv cu:1679)
 *   mask                   : [B, M] bytes, non-zero = keep   (cu:1208-1211; torch.bool storage)
 *   attn_bias              : [Hb, N, M], Hb == H (per head) or Hb == B when bias_batch_dim (cu:1168, cu:1214)
 *   inv_l                  : [B, H, N] float32 = 1 / max(rowsum, eps): the row sums are taken with a library-chosen
 *                            constant exponent shift.  l2norm_qk == 0: shift = scale and eps = 1e-10, the
 *                            reference's values exactly (cu:1216, cu:1236-1242).  l2norm_qk == 1: opaque to the
 *                            caller (forward and backward of this library agree on it); the clamp is the
 *                            reference's rescaled to the shift, except in the wide-range regime below.
 *   Tensors are described by a base pointer and ELEMENT strides for the three leading
 *   dims; the feature dim must be contiguous (stride 1) and every row 16-byte aligned.
 *   A merged batch-heads query ([BH, N, D], cu:1647-1654) is passed as B = BH, H = 1.
 *
 * Errors: every entry point returns FCSA_OK (0) or a negative code and records a message
 * retrievable with fcsa_last_error() (thread local).  Unsupported dtypes / head dims are
 * rejected (the reference silently does nothing: dispatch.h:50-52).
 */
#ifndef FCSA_H_
#define FCSA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCSA_ABI_VERSION 4

enum fcsa_status {
  FCSA_OK = 0,
  FCSA_ERR_INVALID_ARG = -1,   /* bad shape / stride / null pointer / mask together with causal (cu:1675) */
  FCSA_ERR_UNSUPPORTED = -2,   /* dtype or head dim outside {f32,f16,bf16} x {16,32,64,96,128} (cu:1702-1703) */
  FCSA_ERR_LAUNCH = -3,        /* hipGetLastError() after a launch */
  FCSA_ERR_WORKSPACE = -4      /* workspace too small */
};

enum fcsa_dtype {
  FCSA_F32 = 0,
  FCSA_F16 = 1,
  FCSA_BF16 = 2
};

/* A [d0, d1, d2, D] tensor view: element strides of the leading dims, last dim contiguous. */
typedef struct fcsa_tensor {
  void*   ptr;
  int64_t stride0;   /* batch */
  int64_t stride1;   /* head  (0 allowed: broadcast over heads, e.g. single-head kv) */
  int64_t stride2;   /* sequence position */
} fcsa_tensor;

/* Problem description shared by forward and backward (reference: the scalar arguments of
 * forward_kernel / backward_kernel, cu:1072-1088, cu:1339-1360). */
typedef struct fcsa_problem {
  int32_t dtype;            /* fcsa_dtype of q,k,v,o,grads,bias */
  int32_t batch;            /* B */
  int32_t heads;            /* H */
  int32_t kv_heads;         /* H or 1 */
  int32_t q_len;            /* N */
  int32_t k_len;            /* M */
  int32_t dim_head;         /* D in {16,32,64,96,128} (cu:84) */
  int32_t causal;           /* cu:1210: key j valid for query i iff j - (M - N) <= i */
  int32_t bias_batch_dim;   /* attn_bias leading dim is batch (1) or heads (0) (cu:1474) */
  int32_t l2norm_qk;        /* 1: q,k are RAW and the library normalises them (fused form of
                               flash_cosine_sim_attention.py:320-321); 0: q,k used as given
                               (exactly the reference extension's contract) */
  int32_t groups;           /* l2norm groups (flash_cosine_sim_attention.py:50-55); 1 if !l2norm_qk */
  float   scale;            /* logits = scale * qh.kh ; reference exponent shift = -scale (cu:1216).
                               With l2norm_qk the logit range is +-|scale|*groups.  Where no constant shift fits that
                               range into the exponent of the type P is rounded to (float16: |scale|*groups > 11, or any
                               attn_bias -- a bias is unbounded and float16 has no room for it; bfloat16 / float32: > 75,
                               or > 40 with an attn_bias) the forward
                               kernel keeps a per-row exponent reference (online max), normalises the row exactly (no
                               1e-10 clamp: in exp(S - scale) units that clamp would attenuate or zero rows there; the
                               reference kernel itself overflows / zeroes) and saves log2 of the normaliser instead of
                               the normaliser, so any finite scale the public signature admits runs
                               (flash_cosine_sim_attention.py:308-319 has no limit).  Only float16 with
                               |scale| * log2(e) > 60000 is refused: the folded c1 * q^ would leave the type.
                               Supported attn_bias magnitude: float16 and every per-row-reference problem: any finite bias
                               (the online reference includes it).  bfloat16 / float32 with |scale|*groups <= 40 use the
                               constant shift: bias values up to +45 above the largest logit are exact (exp stays below
                               float32's e^88 with ln(M) to spare for the row sum); larger positive values can overflow a
                               row to inf/NaN -- the reference's float32 exp(S - scale + bias) has the same limit at +88 --
                               and strongly negative values underflow to an exact 0 weight, as in the reference. */
} fcsa_problem;

/* State the fused-l2norm forward saves for backward (all caller-allocated, contiguous):
 *   qn [B,H,N,D], kn [B,Hk,M,D] in `dtype`  : normalised q, k (what the reference's autograd
 *        would have saved as the outputs of F.normalize)
 *   rq [B,H,N,G], rk [B,Hk,M,G] float32     : 1 / max(||x_group||, 1e-12)
 * Unused (may be NULL) when l2norm_qk == 0. */
typedef struct fcsa_norm_state {
  void*  qn;                /* may be NULL in a forward call when fcsa_forward_needs_qn() says 0 (inference path) */
  void*  kn;
  float* rq;
  float* rk;
} fcsa_norm_state;

typedef struct fcsa_forward_args {
  fcsa_problem    p;
  fcsa_tensor     q, k, v;       /* inputs (borrowed, never written) */
  fcsa_tensor     o;             /* output [B,H,N,D] */
  float*          inv_l;         /* [B,H,N] contiguous, or NULL when no backward will follow
                                    (reference: need_store_rowsum, cu:1086, cu:1241).  Opaque to the caller: 1 / rowsum,
                                    or log2 of it in the per-row-shift regime (see `scale`) */
  const uint8_t*  mask;          /* [B,M] contiguous or NULL */
  const void*     attn_bias;     /* [Hb,N,M] contiguous or NULL */
  fcsa_norm_state norm;
  void*           workspace;     /* optional: >= fcsa_forward_workspace_bytes(&p) bytes, 256-byte aligned, or NULL.
                                    With it, launches whose row tiles cannot fill the chip split the KEY range over
                                    several workgroups (un-normalised partial (P~V, l) add up exactly -- there is no
                                    running max to reconcile -- and a combine kernel normalises).  Without it the
                                    result is the same, from fewer workgroups.  No reference counterpart (its grid is
                                    row tiles only, cu:1714-1718). */
  size_t          workspace_bytes;
  void*           stream;        /* hipStream_t */
} fcsa_forward_args;

typedef struct fcsa_backward_args {
  fcsa_problem    p;
  fcsa_tensor     d_out, o;      /* [B,H,N,D] */
  const float*    inv_l;         /* [B,H,N] from forward.  Its encoding (1 / rowsum, or log2 of it) is a function of `p`
                                    AND of whether an attn_bias is present: fcsa_backward must be given the same problem
                                    and the same bias (NULL or not) as the fcsa_forward call that wrote it -- which the
                                    mathematics requires anyway (dS depends on the bias) -- and an inv_l is only valid for
                                    the library version that produced it */
  fcsa_tensor     q, k, v;       /* the same tensors forward saw (q,k ignored when l2norm_qk: qn/kn are used) */
  const uint8_t*  mask;
  const void*     attn_bias;
  fcsa_norm_state norm;          /* from forward (l2norm_qk only) */
  fcsa_tensor     dq;            /* [B,H,N,D]  out */
  fcsa_tensor     dk, dv;        /* [B,Hk,M,D] out */
  void*           d_bias;        /* [Hb,N,M] in `dtype` (the dtype the reference returns it in, cu:1912) or NULL.  Every
                                    element is WRITTEN exactly once, deterministically: the d_bias kernel recomputes the dS
                                    tiles of a bias slice, sums the broadcast index (batch or heads) in float32 registers
                                    and rounds once (reference: f32 atomicAdd per element into a zeroed f32 tensor,
                                    cu:1574-1576, then a cast pass, cu:1912).  No zero-fill, no cast needed. */
  void*           workspace;     /* >= fcsa_backward_workspace_bytes(&p) bytes, 256-byte aligned: delta [B,H,N] f32, plus f32
                                    slabs where an epilogue cannot finish the job -- partial dq of the split-key dQ kernel, partial
                                    dk / dv of the split-query dK/dV kernel and of single-headed K/V, l2norm groups that are not
                                    8 * 2^k features wide (one group over the whole head counts as fused at any D: D = 96).  The split forms also need dq (dk, dv) with stride0 == heads * stride1;
                                    other layouts run the unsplit kernels. */
  size_t          workspace_bytes;
  void*           stream;
} fcsa_backward_args;

/* Replaces flash_cosine_sim_attention_forward (cu:1630-1748).
 * Zero-size problems launch nothing: batch, heads or q_len == 0 return FCSA_OK untouched (pointers of empty tensors may be NULL);
 * k_len == 0 makes every row a row without a valid key: o is zero-filled on the stream (inv_l = 1).  fcsa_backward likewise:
 * q_len == 0 or k_len == 0 zero-fills whichever of dq / dk / dv has elements.  (The reference launches an empty grid there.) */
int fcsa_forward(const fcsa_forward_args* args);

/* Replaces flash_cosine_sim_attention_backward (cu:1752-1917): delta pre-pass (cu:1256-1335)
 * + gradient kernels (cu:1339-1626) + the casts at cu:1893-1916.  With l2norm_qk it also
 * applies the l2norm backward that torch.autograd performs in the reference. */
int fcsa_backward(const fcsa_backward_args* args);

/* Scratch needed by fcsa_backward for this problem (delta, f32 gradient slabs). */
size_t fcsa_backward_workspace_bytes(const fcsa_problem* p);

/* Bytes of optional forward scratch that enable the split-key forward for this problem (0: never split). */
size_t fcsa_forward_workspace_bytes(const fcsa_problem* p);

/* 1 if fcsa_forward needs the norm.qn buffer for this problem, else 0.  It always does when a backward follows
 * (`need_backward`: qn is saved state) or l2norm_qk is off (unused then); an inference call needs it only where q is
 * normalised by the row kernel (float32, or several groups that are not 8 * 2^k features wide) -- the 16-bit forward kernels
 * normalise q in registers and then write NOTHING but `o` (the reference's need_store_rowsum == false path, cu:1086). */
int fcsa_forward_needs_qn(const fcsa_problem* p, int32_t need_backward);

/* Standalone grouped l2norm on device: the public l2norm_tensors (flash_cosine_sim_attention.py:57-65).
 * x [B,H,N,D] (strided) -> xn [B,H,N,D] contiguous, inv_norm [B,H,N,G] float32 (may be NULL). */
int fcsa_l2norm(int32_t dtype, int32_t batch, int32_t heads, int32_t len, int32_t dim_head, int32_t groups,
                const fcsa_tensor* x, void* xn, float* inv_norm, void* stream);

/* Replaces the extension's debug() hook (cu:1928-1933): returns the ABI version and, when
 * `buf` is non-NULL, writes a NUL-terminated description of the compiled kernels into it. */
int fcsa_debug(char* buf, size_t buf_bytes);

/* Optional per-kernel timing for bench.py's roofline line (not part of the reference's surface).
 * While enabled, the library records a HIP event pair on the launch stream around every kernel it
 * launches; fcsa_profile_collect waits for those events, aggregates them per kernel name
 * ("l2norm", "fwd", "bwd_dq", "bwd_dkv", "finalize"), writes up to `capacity` entries and returns
 * the number of distinct kernels seen (negative on error).  Collecting resets the record. */
typedef struct fcsa_kernel_stat {
  char    name[32];
  int32_t calls;
  float   total_ms;
  float   min_ms;
  float   max_ms;
} fcsa_kernel_stat;
int fcsa_profile_enable(int32_t enable);
int fcsa_profile_collect(fcsa_kernel_stat* stats, int32_t capacity);

/* Debug knob (same-process A/B runs and triage; no reference counterpart): which forward form 16-bit D = 128 problems may take.
 * form = 1: automatic (the 64-rows-per-wave kernel of csrc/fcsa_fwd3.hip where its dispatch rule applies -- the default);
 * form = 0: never that kernel (the 32-rows-per-wave forms instead); form < 0: query only.  Returns the previous setting.
 * The initial value comes from the environment variable FCSA_FWD_WIDE128 ("0" = form 0), read ONCE when the library is loaded;
 * no entry point reads the environment afterwards.  The two forms agree to one ulp of the 16-bit output (row sums of the
 * un-rounded vs the rounded P~: DESIGN.md section 5), so results are form-dependent at that level; callers that need
 * form-independent bits pin the form with this call. */
int fcsa_debug_forward_form(int32_t form);

/* Message for the last non-OK status returned on this thread ("" if none). */
const char* fcsa_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* FCSA_H_ */
