"""Public operator API -- same names, keywords and defaults as the reference package
(flash_cosine_sim_attention/__init__.py:1, flash_cosine_sim_attention.py:308-334).

Differences from the reference, all inside the boundary:
  * the (grouped) l2norm of q and k is FUSED into the op (q in the forward kernel's prologue, k in
    a library row kernel, inverse norms saved; the l2norm backward in the dQ / dKV epilogues)
    instead of two eager F.normalize passes outside the autograd.Function
    (flash_cosine_sim_attention.py:320-321), so `FlashCosineSimAttention.backward` returns
    gradients w.r.t. the RAW q, k;
  * GPU tensors only ever run on the hand-written gfx950 kernels; there is no silent fallback
    (a missing libfcsa_hip.so raises ImportError).  CPU tensors take this package's own
    forward-only blockwise path (`cpu.py`), as they do in the reference (py:322-323).

Logit range: with l2norm_qk the logits lie in +-|scale|*groups.  Inside the library's static exponent window (f16: <= 11 and no attn_bias,
bf16 / f32: <= 75) the kernels use one constant shift like the reference (whose kernel overflows / zeroes rows at the far end of that
range); beyond it the forward shifts every row by its own max logit and normalises exactly, like the reference's PyTorch
plain_cosine_sim_attention.  There is no limit on scale (only float16 refuses |scale| * log2(e) > 60000).
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import _torch_ops
from . import cpu as _cpu


# ---------------------------------------------------------------------------------------------
# l2norm helpers: public exports of the reference package (flash_cosine_sim_attention.py:38-65).
# Differentiable torch code; the fused operator does not call them.
# ---------------------------------------------------------------------------------------------

def _norm_floor(t: torch.Tensor) -> float:
    # the reference clamps the norm at F.normalize's 1e-12 on the GPU and, on the CPU, at 1e-3 for the 16-bit types (py:38-48)
    return 1e-12 if (t.is_cuda or t.dtype == torch.float32) else 1e-3


def l2norm(t):
    """t / max(||t||_2, eps) over the last dimension."""
    length = torch.linalg.vector_norm(t, dim=-1, keepdim=True)
    return t / length.clamp_min(_norm_floor(t))


def grouped_l2norm(t, groups=1):
    """l2norm of each of `groups` equal slices of the last dimension."""
    width = t.shape[-1]
    if groups < 1 or width % groups:
        raise ValueError(f"groups ({groups}) must divide the last dimension ({width})")
    return l2norm(t.unflatten(-1, (groups, width // groups))).flatten(-2)


def l2norm_tensors(*tensors, groups=1):
    """Grouped l2norm of every tensor, each cast to the dtype of the FIRST one (py:57-65)."""
    if not tensors:
        raise ValueError("l2norm_tensors needs at least one tensor")
    target = tensors[0].dtype
    return tuple(grouped_l2norm(t, groups).to(target) for t in tensors)


# ---------------------------------------------------------------------------------------------
# O(N*M)-memory attention in plain PyTorch ops: the reference's `plain_cosine_sim_attention`
# (py:75-126), kept as a public export.  Runs wherever its inputs live.  Not used by the fused op.
# ---------------------------------------------------------------------------------------------

def plain_cosine_sim_attention(q, k, v, mask=None, attn_bias=None, scale=8, groups=1, causal=False,
                               l2norm_qk=True, attn_bias_batch_dim=False):
    if causal and mask is not None:
        raise AssertionError("mask should not be supplied if causality is needed")
    squeeze_heads = q.dim() == 3                       # merged batch-heads queries
    if squeeze_heads:
        if k.dim() != 3 or v.dim() != 3:
            raise AssertionError("if batch and heads are merged for queries, keys and values must also similarly have only 3 dimensions")
        attn_bias_batch_dim, q = True, q.unsqueeze(1)
    q, k = l2norm_tensors(q, k, groups=groups) if l2norm_qk else (q, k)
    keys = k.unsqueeze(1) if k.dim() == 3 else k       # single-headed K/V broadcast over the heads
    values = v.unsqueeze(1) if v.dim() == 3 else v
    logits = torch.matmul(q, keys.transpose(-1, -2)) * scale
    if attn_bias is not None:
        logits = logits + attn_bias.unsqueeze(1 if attn_bias_batch_dim else 0)
    lowest = -torch.finfo(logits.dtype).max
    n, m = logits.shape[-2:]
    if causal:                                         # key j is visible to query i iff j - (m - n) <= i
        future = torch.ones((n, m), dtype=torch.bool, device=logits.device).triu(m - n + 1)
        logits = logits.masked_fill(future, lowest)
    if mask is not None:
        logits = logits.masked_fill(~mask[:, None, None, :], lowest)
    out = torch.matmul(logits.softmax(dim=-1), values)
    return out.squeeze(1) if squeeze_heads else out


# ---------------------------------------------------------------------------------------------
# autograd.Function (flash_cosine_sim_attention.py:245-304) with the l2norm fused in
# ---------------------------------------------------------------------------------------------

class FlashCosineSimAttention(Function):
    """The reference's autograd.Function name (py:245), kept as a public export: a thin Python Function over the compiled
    dispatcher ops `torch.ops.fcsa.forward / backward`.  `flash_cosine_sim_attention` itself uses the same two ops through a
    C++ autograd node (`torch.ops.fcsa.attention`), which saves the ~60 us a Python Function costs per forward+backward."""

    @staticmethod
    def forward(ctx, q, k, v, mask, attn_bias, scale, groups, causal, l2norm_qk, attn_bias_batch_dim):
        if not q.is_cuda:
            raise RuntimeError("flash_cosine_sim_attention_amd: q, k, v must be GPU tensors (HIP kernels only, no CPU fallback)")
        fc = _torch_ops.load()
        should_backwards = any(ctx.needs_input_grad[i] for i in (0, 1, 2, 4))              # q, k, v, attn_bias (cu:1689)
        o, inv_l, qn, kn, rq, rk = fc.forward(q, k, v, mask, attn_bias, bool(attn_bias_batch_dim), float(scale), bool(causal),
                                              bool(l2norm_qk), int(groups), should_backwards)
        ctx.should_backwards = should_backwards
        if not should_backwards:
            return o
        # tensors go through save_for_backward (py:270): no reference cycle through ctx, in-place modification of a saved
        # input is detected, saved-tensor hooks (checkpointing, offload) apply; scalars stay plain attributes
        ctx.save_for_backward(o, inv_l, q, k, v, mask, attn_bias, qn, kn, rq, rk)
        ctx.scalars = (bool(attn_bias_batch_dim), float(scale), bool(causal), bool(l2norm_qk), int(groups))
        ctx.bias_grad = bool(attn_bias is not None and ctx.needs_input_grad[4])
        return o

    @staticmethod
    def backward(ctx, do):
        assert ctx.should_backwards
        o, inv_l, q, k, v, mask, attn_bias, qn, kn, rq, rk = ctx.saved_tensors
        dq, dk, dv, db = _torch_ops.load().backward(do, o, inv_l, q, k, v, mask, attn_bias, qn, kn, rq, rk, *ctx.scalars,
                                                    ctx.bias_grad)
        return dq, dk, dv, None, (db if ctx.bias_grad else None), None, None, None, None, None


flash_cosine_sim_attention_hip = FlashCosineSimAttention.apply


def flash_cosine_sim_attention(q, k, v, mask=None, attn_bias=None, scale=8, groups=1, causal=False,
                               l2norm_qk=True, attn_bias_batch_dim=False):
    """Fused cosine-similarity attention; signature of flash_cosine_sim_attention.py:308-319.

    GPU tensors: hand-written gfx950 kernels, forward and backward (gradients w.r.t. the raw q, k, v, attn_bias).
    CPU tensors: forward-only blockwise path (`cpu.attention_forward_cpu`), like the reference (py:322-323).
    Any finite scale runs (see the module docstring for the exponent-shift regimes)."""
    if q.device.type == "cpu":
        return _cpu.attention_forward_cpu(q, k, v, mask=mask, attn_bias=attn_bias, scale=scale, groups=groups, causal=causal,
                                          l2norm_qk=l2norm_qk, attn_bias_batch_dim=attn_bias_batch_dim)
    # the differentiable dispatcher op: autograd node, checks, allocation and launches all in C++ (csrc/fcsa_torch.cpp)
    return _torch_ops.load().attention(q, k, v, mask, attn_bias, bool(attn_bias_batch_dim), float(scale), bool(causal),
                                       bool(l2norm_qk), int(groups))
