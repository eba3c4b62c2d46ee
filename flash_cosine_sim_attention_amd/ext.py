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

from . import _core, _lib


def forward(q, k, v, mask, attn_bias, attn_bias_batch_dim, scale, causal):
    should_backwards = any(t is not None and t.requires_grad for t in (q, k, v, attn_bias))       # cu:1689
    o, saved = _core.attention_forward(q, k, v, mask, attn_bias, attn_bias_batch_dim, scale, causal,
                                       l2norm_qk=False, groups=1, need_backward=should_backwards)
    inv_l = saved.inv_l if saved is not None else torch.empty((q.shape[0], 0), device=q.device, dtype=torch.float32)
    # merged batch-heads (q.dim() == 3): inv_l stays [BH, 1, N]; the reference keeps the unsqueezed head dim too (cu:1698)
    return o, inv_l, should_backwards


def backward(d_out, o, inv_l, q, k, v, mask, attn_bias, attn_bias_batch_dim, scale, causal):
    empty = q.new_empty((0,))
    empty32 = q.new_empty((0,), dtype=torch.float32)
    saved = _core.Saved(o, inv_l, q, k, v, mask, attn_bias, empty, empty, empty32, empty32, float(scale), 1, bool(causal), False,
                        bool(attn_bias_batch_dim))
    return _core.attention_backward(d_out, saved, need_bias_grad=attn_bias is not None)


def debug():
    """Reference: an empty hook (cu:1919-1926).  Here: returns the library's self-description."""
    lib = _lib.load()
    buf = C.create_string_buffer(512)
    lib.fcsa_debug(buf, 512)
    return buf.value.decode()
