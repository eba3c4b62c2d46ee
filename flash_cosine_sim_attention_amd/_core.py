"""Python face of the host launchers.

What the reference's C++ host launchers do around the kernels (flash_cosine_sim_attention_cuda.cu:1630-1748 forward,
cu:1752-1917 backward) -- shape canonicalisation (3-D q = merged batch-heads, 3-D k/v = single-headed key/values,
cu:1647-1660), output / saved-state allocation (cu:1697-1698, cu:1820-1827), argument checks (raised as Python exceptions
instead of the reference's compiled-out C asserts, cu:1650, cu:1673-1675) and the final casts (cu:1893-1916) -- lives in the
compiled binding csrc/fcsa_torch.cpp (`torch.ops.fcsa.forward / backward`); this module only names the saved state and
forwards the calls.  torch is used for device memory and the current stream only.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib, _torch_ops

_DTYPES = {torch.float32: _lib.FCSA_F32, torch.float16: _lib.FCSA_F16, torch.bfloat16: _lib.FCSA_BF16}
ALLOWED_DIM_HEADS = (16, 32, 64, 96, 128)        # cu:84


@dataclass
class Saved:
    """What forward keeps for backward (reference: ctx.save_for_backward, flash_cosine_sim_attention.py:270).
    q, k, v are the caller's tensors (any accepted shape); qn / kn / rq / rk are empty tensors when l2norm_qk is off."""
    o: torch.Tensor
    inv_l: torch.Tensor
    q: torch.Tensor
    k: torch.Tensor
    v: torch.Tensor
    mask: Optional[torch.Tensor]
    attn_bias: Optional[torch.Tensor]
    qn: Optional[torch.Tensor]
    kn: Optional[torch.Tensor]
    rq: Optional[torch.Tensor]
    rk: Optional[torch.Tensor]
    scale: float
    groups: int
    causal: bool
    l2norm_qk: bool
    attn_bias_batch_dim: bool


def _rows_ok(t: torch.Tensor) -> bool:
    es = t.element_size()
    m = 16 // es
    return (t.stride(-1) == 1 and t.data_ptr() % 16 == 0 and all(s % m == 0 for s in t.stride()[:-1]))


def _prep(t: torch.Tensor) -> torch.Tensor:
    """Strided [.., L, D] views with a contiguous, 16-byte aligned feature dim are consumed in place
    (e.g. the `b n (h d) -> b h n d` views of transformer.py:100); anything else is made contiguous."""
    return t if _rows_ok(t) else t.contiguous()


# ---- ctypes helpers for code that drives the C ABI directly (tests, tools): struct builders over torch tensors ----

def _tensor4(t: torch.Tensor) -> _lib.Tensor:
    assert t.dim() == 4
    return _lib.Tensor(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


def _problem(dtype, dims, causal, bias_batch, l2norm_qk, groups, scale) -> _lib.Problem:
    B, H, Hk, N, M, D = dims
    return _lib.Problem(_DTYPES[dtype], B, H, Hk, N, M, D, int(bool(causal)), int(bool(bias_batch)),
                        int(bool(l2norm_qk)), int(groups if l2norm_qk else 1), float(scale))


def attention_forward(q, k, v, mask=None, attn_bias=None, attn_bias_batch_dim=False, scale=8.0, causal=False,
                      l2norm_qk=False, groups=1, need_backward=False) -> Tuple[torch.Tensor, Optional[Saved]]:
    """o = fused cosine-sim attention.  With l2norm_qk the (grouped) l2norm of q, k is done by the library.
    One call into the compiled binding (csrc/fcsa_torch.cpp), which canonicalises, checks, allocates and launches."""
    if not q.is_cuda:
        raise RuntimeError("flash_cosine_sim_attention_amd: q, k, v must be GPU tensors (HIP kernels only, no CPU fallback)")
    ops = _torch_ops.load()
    o, inv_l, qn, kn, rq, rk = ops.forward(q, k, v, mask, attn_bias, bool(attn_bias_batch_dim), float(scale), bool(causal),
                                           bool(l2norm_qk), int(groups), bool(need_backward))
    saved = None
    if need_backward:
        saved = Saved(o, inv_l, q, k, v, mask, attn_bias, qn, kn, rq, rk, float(scale), int(groups), bool(causal),
                      bool(l2norm_qk), bool(attn_bias_batch_dim))
    return o, saved


def attention_backward(d_out: torch.Tensor, s: Saved, q_shape=None, k_shape=None, v_shape=None, need_bias_grad: bool = False):
    """(dq, dk, dv, d_bias) in the shapes / dtype of the original inputs (d_bias None unless requested)."""
    ops = _torch_ops.load()
    want_db = bool(need_bias_grad and s.attn_bias is not None)
    dq, dk, dv, db = ops.backward(d_out, s.o, s.inv_l, s.q, s.k, s.v, s.mask, s.attn_bias, s.qn, s.kn, s.rq, s.rk,
                                  s.attn_bias_batch_dim, s.scale, s.causal, s.l2norm_qk, s.groups, want_db)
    return dq, dk, dv, (db if want_db else None)


def l2norm_device(t: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """Grouped l2norm of a GPU tensor with the library's row kernel (the C entry point fcsa_l2norm, include/fcsa.h)."""
    lib = _lib.load()
    shape = t.shape
    D = shape[-1]
    t4 = _prep(t.reshape(1, 1, -1, D) if t.dim() < 4 else t.reshape(-1, shape[-3], shape[-2], D))
    out = torch.empty(t4.shape, device=t.device, dtype=t.dtype)
    with torch.cuda.device(t.device):
        x = _lib.Tensor(t4.data_ptr(), t4.stride(0), t4.stride(1), t4.stride(2))
        _lib.check(lib.fcsa_l2norm(_DTYPES[t.dtype], t4.shape[0], t4.shape[1], t4.shape[2], D, groups, C.byref(x), out.data_ptr(),
                                   None, torch.cuda.current_stream(t.device).cuda_stream), "fcsa_l2norm")
    return out.reshape(shape)
