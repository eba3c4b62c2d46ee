"""Host glue between torch tensors and the C ABI (include/fcsa.h).

Mirrors what the reference's C++ host launchers do around the kernels
(flash_cosine_sim_attention_cuda.cu:1630-1748 forward, cu:1752-1917 backward):
shape canonicalisation (3-D q = merged batch-heads, 3-D k/v = single-headed
key/values, cu:1647-1660), output / saved-state allocation (cu:1697-1698,
cu:1820-1827), argument checks (as Python exceptions instead of the reference's
compiled-out C asserts, cu:1650, cu:1673-1675), and the final casts
(cu:1893-1916).  torch is used for device memory and the current stream only.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib

_DTYPES = {torch.float32: _lib.FCSA_F32, torch.float16: _lib.FCSA_F16, torch.bfloat16: _lib.FCSA_BF16}
ALLOWED_DIM_HEADS = (16, 32, 64, 96, 128)        # cu:84


@dataclass
class Saved:
    """What forward keeps for backward (reference: ctx.save_for_backward, flash_cosine_sim_attention.py:270)."""
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


def _tensor4(t: torch.Tensor) -> _lib.Tensor:
    assert t.dim() == 4
    return _lib.Tensor(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already current (the context manager costs ~10 us per call,
    a third of a small forward)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if torch.cuda.current_device() == (dev.index if dev.index is not None else 0) else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _canonicalise(q, k, v, mask, attn_bias, attn_bias_batch_dim, causal):
    if not q.is_cuda:
        raise RuntimeError("flash_cosine_sim_attention_amd: q, k, v must be GPU tensors (HIP kernels only, no CPU fallback)")
    for name, t in (("k", k), ("v", v), ("mask", mask), ("attn_bias", attn_bias)):
        if t is not None and t.device != q.device:
            raise ValueError(f"{name} is on {t.device} but q is on {q.device}: all tensors must live on q's GPU")
    if not (q.dtype == k.dtype == v.dtype):
        raise TypeError(f"q, k, v must share a dtype, got {q.dtype}, {k.dtype}, {v.dtype}")
    if q.dtype not in _DTYPES:
        raise TypeError(f"unsupported dtype {q.dtype}; expected float32, float16 or bfloat16")
    if causal and mask is not None:
        raise ValueError("mask should not be supplied if causality is needed")          # fcsa.py:88, cu:1675
    merged = q.dim() == 3
    if merged:
        if not (k.dim() == 3 and v.dim() == 3):
            raise ValueError("if batch and heads are merged for queries, keys and values must also have 3 dimensions")
        attn_bias_batch_dim = True                                                       # cu:1652
        q4 = q.unsqueeze(1)
    else:
        if q.dim() != 4:
            raise ValueError(f"q must have 3 or 4 dimensions, got {q.dim()}")
        q4 = q
    k4 = k.unsqueeze(1) if k.dim() == 3 else k
    v4 = v.unsqueeze(1) if v.dim() == 3 else v
    if k4.dim() != 4 or v4.dim() != 4:
        raise ValueError("k and v must have 3 or 4 dimensions")
    B, H, N, D = q4.shape
    Bk, Hk, M, Dk = k4.shape
    if tuple(v4.shape) != tuple(k4.shape):
        raise ValueError(f"k and v must have the same shape, got {tuple(k.shape)} and {tuple(v.shape)}")
    if Dk != D:
        raise ValueError("query, key, value dimensions must be the same")                 # cu:1673
    if D not in ALLOWED_DIM_HEADS:
        raise ValueError(f"only dimensions {ALLOWED_DIM_HEADS} allowed for now, got {D}")  # cu:1674
    if Bk != B:
        raise ValueError(f"batch mismatch between q ({B}) and k/v ({Bk})")
    if Hk != H and Hk != 1:
        raise ValueError(f"k/v heads must equal q heads ({H}) or be 1 (single-headed key/values), got {Hk}")
    if mask is not None:
        if mask.dtype != torch.bool or tuple(mask.shape) != (B, M):
            raise ValueError(f"mask must be a bool tensor of shape {(B, M)}, got {mask.dtype} {tuple(mask.shape)}")
        mask = mask.contiguous()
    if attn_bias is not None:
        lead = B if attn_bias_batch_dim else H
        if tuple(attn_bias.shape) != (lead, N, M):
            raise ValueError(f"attn_bias must have shape {(lead, N, M)}, got {tuple(attn_bias.shape)}")
        if attn_bias.dtype != q.dtype:
            raise TypeError("attn_bias must have the dtype of q")
        attn_bias = attn_bias.contiguous()
    return q4, k4, v4, mask, attn_bias, attn_bias_batch_dim, merged, (B, H, Hk, N, M, D)


def _problem(dtype, dims, causal, bias_batch, l2norm_qk, groups, scale) -> _lib.Problem:
    B, H, Hk, N, M, D = dims
    return _lib.Problem(_DTYPES[dtype], B, H, Hk, N, M, D, int(bool(causal)), int(bool(bias_batch)),
                        int(bool(l2norm_qk)), int(groups if l2norm_qk else 1), float(scale))


def attention_forward(q, k, v, mask=None, attn_bias=None, attn_bias_batch_dim=False, scale=8.0, causal=False,
                      l2norm_qk=False, groups=1, need_backward=False) -> Tuple[torch.Tensor, Optional[Saved]]:
    """o = fused cosine-sim attention.  With l2norm_qk the (grouped) l2norm of q, k is done by the library."""
    lib = _lib.load()
    q4, k4, v4, mask, attn_bias, bias_batch, merged, dims = _canonicalise(q, k, v, mask, attn_bias, attn_bias_batch_dim, causal)
    B, H, Hk, N, M, D = dims
    if l2norm_qk and (groups < 1 or D % groups != 0):
        raise ValueError(f"groups ({groups}) must divide the head dimension ({D})")
    q4, k4, v4 = _prep(q4), _prep(k4), _prep(v4)
    dev, dt = q.device, q.dtype
    with _on_device(dev):
        o = torch.empty((B, H, N, D), device=dev, dtype=dt)
        inv_l = torch.empty((B, H, N), device=dev, dtype=torch.float32) if need_backward else None
        qn = kn = rq = rk = None
        if l2norm_qk:
            qn = torch.empty((B, H, N, D), device=dev, dtype=dt)
            kn = torch.empty((B, Hk, M, D), device=dev, dtype=dt)
            if need_backward:
                rq = torch.empty((B, H, N, groups), device=dev, dtype=torch.float32)
                rk = torch.empty((B, Hk, M, groups), device=dev, dtype=torch.float32)
        prob = _problem(dt, dims, causal, bias_batch, l2norm_qk, groups, scale)
        ws = None
        if not causal and B * H * N <= 16384:          # only grids that cannot fill the chip ever split (saves the call otherwise)
            ws_bytes = int(lib.fcsa_forward_workspace_bytes(C.byref(prob)))
            if ws_bytes:
                ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
        args = _lib.ForwardArgs(
            prob,
            _tensor4(q4), _tensor4(k4), _tensor4(v4), _tensor4(o),
            _ptr(inv_l), _ptr(mask), _ptr(attn_bias),
            _lib.NormState(_ptr(qn), _ptr(kn), _ptr(rq), _ptr(rk)),
            _ptr(ws), 0 if ws is None else ws.numel(),
            _stream_ptr(dev))
        _lib.check(lib.fcsa_forward(C.byref(args)), "fcsa_forward")
    out = o.squeeze(1) if merged else o                                                  # cu:1740-1741
    saved = None
    if need_backward:
        saved = Saved(o, inv_l, q4, k4, v4, mask, attn_bias, qn, kn, rq, rk, float(scale), int(groups),
                      bool(causal), bool(l2norm_qk), bool(bias_batch))
    return out, saved


def attention_backward(d_out: torch.Tensor, s: Saved, q_shape, k_shape, v_shape, need_bias_grad: bool):
    """(dq, dk, dv, d_bias) in the shapes / dtype of the original inputs."""
    lib = _lib.load()
    q4, k4, v4, o = s.q, s.k, s.v, s.o
    B, H, N, D = q4.shape
    Hk, M = k4.shape[1], k4.shape[2]
    dims = (B, H, Hk, N, M, D)
    dev, dt = q4.device, q4.dtype
    do4 = d_out.unsqueeze(1) if d_out.dim() == 3 else d_out
    if do4.dtype != dt:
        do4 = do4.to(dt)
    do4 = _prep(do4)
    with _on_device(dev):
        dq = torch.empty((B, H, N, D), device=dev, dtype=dt)
        dk = torch.empty((B, Hk, M, D), device=dev, dtype=dt)
        dv = torch.empty((B, Hk, M, D), device=dev, dtype=dt)
        db32 = None
        if s.attn_bias is not None and need_bias_grad:
            db32 = torch.zeros(s.attn_bias.shape, device=dev, dtype=torch.float32)       # cu:1827 (f32 atomics target)
        prob = _problem(dt, dims, s.causal, s.attn_bias_batch_dim, s.l2norm_qk, s.groups, s.scale)
        ws_bytes = int(lib.fcsa_backward_workspace_bytes(C.byref(prob)))
        ws = torch.empty((max(ws_bytes, 256),), device=dev, dtype=torch.uint8)
        args = _lib.BackwardArgs(
            prob, _tensor4(do4), _tensor4(o), _ptr(s.inv_l),
            _tensor4(q4), _tensor4(k4), _tensor4(v4), _ptr(s.mask), _ptr(s.attn_bias),
            _lib.NormState(_ptr(s.qn), _ptr(s.kn), _ptr(s.rq), _ptr(s.rk)),
            _tensor4(dq), _tensor4(dk), _tensor4(dv), _ptr(db32),
            ws.data_ptr(), ws.numel(), _stream_ptr(dev))
        _lib.check(lib.fcsa_backward(C.byref(args)), "fcsa_backward")
    db = db32.to(dt) if db32 is not None else None                                       # cu:1912
    return dq.reshape(q_shape), dk.reshape(k_shape), dv.reshape(v_shape), db


def l2norm_device(t: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """Grouped l2norm of a GPU f16/bf16 tensor with the library's row kernel (fcsa_l2norm)."""
    lib = _lib.load()
    shape = t.shape
    D = shape[-1]
    t3 = _prep(t.reshape(1, 1, -1, D) if t.dim() < 4 else t.reshape(-1, shape[-3], shape[-2], D))
    out = torch.empty(t3.shape, device=t.device, dtype=t.dtype)
    with _on_device(t.device):
        x = _tensor4(t3)
        _lib.check(lib.fcsa_l2norm(_DTYPES[t.dtype], t3.shape[0], t3.shape[1], t3.shape[2], D, groups,
                                   C.byref(x), out.data_ptr(), None, _stream_ptr(t.device)), "fcsa_l2norm")
    return out.reshape(shape)
