"""Drop-in replacement for the reference's native extension module
`flash_cosine_sim_attention_cuda_0_1_40` (pybind, flash_cosine_sim_attention_cuda.cu:1928-1933):

    forward(q, k, v, mask, attn_bias, attn_bias_batch_dim, scale, causal) -> (o, inv_l, should_backwards)
    backward(d_out, o, inv_l, q, k, v, mask, attn_bias, attn_bias_batch_dim, scale, causal) -> (dq, dk, dv, db)
    debug()

Same argument order, meaning and return values as cu:1630-1639 / cu:1752-1764, so the
reference's `FlashCosineSimAttention` autograd.Function (flash_cosine_sim_attention.py:245-302)
works unchanged on top of it.  Here q and k are expected ALREADY
normalised, exactly as in the reference; the fused-l2norm entry points live in `ops.py`.
Implemented over the C ABI of libfcsa_hip.so (include/fcsa.h) -- hand-written gfx950 kernels.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, _torch_ops


def forward(q, k, v, mask, attn_bias, attn_bias_batch_dim, scale, causal):
    if not q.is_cuda:
        raise RuntimeError("flash_cosine_sim_attention_amd: q, k, v must be GPU tensors (HIP kernels only, no CPU fallback)")
    should_backwards = any(t is not None and t.requires_grad for t in (q, k, v, attn_bias))       # cu:1689
    o, inv_l, _, _, _, _ = _torch_ops.load().forward(q, k, v, mask, attn_bias, bool(attn_bias_batch_dim), float(scale), bool(causal),
                                                     False, 1, should_backwards)
    if not should_backwards:
        inv_l = torch.empty((q.shape[0], 0), device=q.device, dtype=torch.float32)
    # merged batch-heads (q.dim() == 3): inv_l stays [BH, 1, N]; the reference keeps the unsqueezed head dim too (cu:1698)
    return o, inv_l, should_backwards


def backward(d_out, o, inv_l, q, k, v, mask, attn_bias, attn_bias_batch_dim, scale, causal):
    empty = q.new_empty((0,))
    empty32 = q.new_empty((0,), dtype=torch.float32)
    want_db = attn_bias is not None
    dq, dk, dv, db = _torch_ops.load().backward(d_out, o, inv_l.contiguous(), q, k, v, mask, attn_bias, empty, empty, empty32, empty32,
                                                bool(attn_bias_batch_dim), float(scale), bool(causal), False, 1, want_db)
    return dq, dk, dv, (db if want_db else None)


def debug():
    """Reference: an empty hook (cu:1919-1926).  Here: returns the library's self-description."""
    lib = _lib.load()
    buf = C.create_string_buffer(512)
    lib.fcsa_debug(buf, 512)
    return buf.value.decode()


def l2norm_device(t: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """Grouped l2norm of a GPU tensor with the library's row kernel (the C entry point fcsa_l2norm, include/fcsa.h)."""
    lib = _lib.load()
    shape = t.shape
    D = shape[-1]
    if t.numel() == 0:
        return torch.empty_like(t)
    t4 = t.reshape(1, 1, -1, D) if t.dim() < 4 else t.reshape(-1, shape[-3], shape[-2], D)
    if t4.stride(-1) != 1 or t4.data_ptr() % 16 or any(s % (16 // t4.element_size()) for s in t4.stride()[:-1]):
        t4 = t4.contiguous()
    out = torch.empty(t4.shape, device=t.device, dtype=t.dtype)
    with torch.cuda.device(t.device):
        x = _lib.tensor4(t4)
        _lib.check(lib.fcsa_l2norm(_lib.dtype_code(t.dtype), t4.shape[0], t4.shape[1], t4.shape[2], D, groups, C.byref(x), out.data_ptr(),
                                   None, torch.cuda.current_stream(t.device).cuda_stream), "fcsa_l2norm")
    return out.reshape(shape)
