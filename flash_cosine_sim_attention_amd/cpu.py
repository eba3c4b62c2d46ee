"""CPU-tensor behaviour of `flash_cosine_sim_attention` -- the operator's forward-only host path.

The reference dispatches non-CUDA tensors to a tiled pure-PyTorch forward
(flash_cosine_sim_attention.py:322-323 -> py:130-241) that never materialises the N x M logits and
refuses inputs that require gradients (py:142-144).  This module is this package's own version
of that path, so that callers which feed CPU tensors keep working:

  * blockwise accumulation of the un-normalised pair (P~ V, rowsum P~) with a CONSTANT exponent shift --
    the logits are bounded by scale * groups because q, k are l2-normalised, so no running max and no
    rescaling is needed (same algebra as the GPU kernels);
  * the key range of a row block is clipped to what causality allows BEFORE looping over key blocks, so
    blocks above the diagonal are never touched (the reference's own skip test, py:215, is inverted --
    it is wrong for causal N > 512, see DESIGN.md "Known reference defect");
  * float32 arithmetic inside, result cast back to the input dtype (py:147, py:241);
  * rows without any valid key come out as 0 (kernel semantics).

It is NOT a fallback for GPU tensors: those only ever run on the HIP kernels (`ops.py`), and a missing
libfcsa_hip.so raises.  Nothing here imports `oracle/` (test infrastructure).
"""
from __future__ import annotations

import torch


def normalise_groups(t: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """Grouped l2norm over the last dim with the reference's CPU clamp (py:38-42: 1e-12 for float32, 1e-3 for the
    16-bit types), computed in float32 and rounded back to t's dtype (py:63-64)."""
    d = t.shape[-1]
    if groups < 1 or d % groups:
        raise ValueError(f"groups ({groups}) must divide the head dimension ({d})")
    floor = 1e-12 if t.dtype == torch.float32 else 1e-3
    g = t.float().reshape(*t.shape[:-1], groups, d // groups)
    length = torch.linalg.vector_norm(g, dim=-1, keepdim=True)
    return (g / length.clamp_min(floor)).reshape(t.shape).to(t.dtype)


def attention_forward_cpu(q, k, v, mask=None, attn_bias=None, scale=8.0, groups=1, causal=False, l2norm_qk=True,
                          attn_bias_batch_dim=False, row_block=256, key_block=1024):
    """Forward-only blockwise cosine-sim attention on host tensors.  Same argument meaning as the GPU operator."""
    for name, t in (("q", q), ("k", k), ("v", v), ("attn_bias", attn_bias)):
        if t is not None and t.requires_grad:
            raise RuntimeError(f"{name} requires grad: the CPU path of flash_cosine_sim_attention is forward-only "
                               "(like the reference's, flash_cosine_sim_attention.py:142-144)")
    if causal and mask is not None:
        raise ValueError("mask should not be supplied if causality is needed")
    out_dtype, out_shape = q.dtype, q.shape
    merged = q.dim() == 3
    if merged:
        if k.dim() != 3 or v.dim() != 3:
            raise ValueError("if batch and heads are merged for queries, keys and values must also have 3 dimensions")
        attn_bias_batch_dim = True
        q = q.unsqueeze(1)
    if q.dim() != 4:
        raise ValueError(f"q must have 3 or 4 dimensions, got {q.dim()}")
    if l2norm_qk:
        q, k = normalise_groups(q, groups), normalise_groups(k, groups)
    k4 = k.unsqueeze(1) if k.dim() == 3 else k           # [B, 1 or H, M, D]: single-headed K/V broadcast over heads
    v4 = v.unsqueeze(1) if v.dim() == 3 else v
    B, H, N, D = q.shape
    M = k4.shape[2]
    qf, kt, vf = q.float(), k4.float().transpose(-1, -2), v4.float()
    bias = None
    if attn_bias is not None:
        bias = attn_bias.float().unsqueeze(1 if attn_bias_batch_dim else 0)          # [B,1,N,M] or [1,H,N,M]
    keep_keys = None if mask is None else mask.to(torch.bool)[:, None, None, :]      # [B,1,1,M]
    offset = M - N                                        # key j is visible to query i iff j <= i + offset (cu:1210)

    # Running row max with rescale (the exponent shift of a row is its own largest visible logit so far): exact for ANY scale,
    # groups, bias and for unnormalised q, k (l2norm_qk=False) -- a constant shift of scale * groups underflows whole rows to 0 once
    # scale * groups is large, and overflows without the l2norm.
    acc = torch.zeros((B, H, N, D), dtype=torch.float32)
    total = torch.zeros((B, H, N, 1), dtype=torch.float32)
    top = torch.full((B, H, N, 1), float("-inf"), dtype=torch.float32)
    for r0 in range(0, N, row_block):
        r1 = min(N, r0 + row_block)
        last = M if not causal else min(M, r1 + offset)   # keys >= last are invisible to every row of this block
        rows = qf[:, :, r0:r1]
        a, s, t = acc[:, :, r0:r1], total[:, :, r0:r1], top[:, :, r0:r1]
        for c0 in range(0, max(last, 0), key_block):
            c1 = min(last, c0 + key_block)
            logits = torch.matmul(rows, kt[..., c0:c1]) * scale
            if bias is not None:
                logits = logits + bias[:, :, r0:r1, c0:c1]
            visible = None
            if causal and c1 - 1 > r0 + offset:           # the block touches the diagonal
                ii = torch.arange(r0, r1).unsqueeze(1) + offset
                jj = torch.arange(c0, c1).unsqueeze(0)
                visible = (jj <= ii)
            if keep_keys is not None:
                visible = keep_keys[..., c0:c1] if visible is None else (visible & keep_keys[..., c0:c1])
            if visible is not None:
                logits = logits.masked_fill(~visible, float("-inf"))
            t_new = torch.maximum(t, logits.amax(dim=-1, keepdim=True))
            safe = torch.where(torch.isinf(t_new), torch.zeros_like(t_new), t_new)      # rows that have seen no key yet
            w = torch.exp(logits - safe)
            rescale = torch.exp(torch.where(torch.isinf(t), torch.zeros_like(t), t) - safe)
            a.mul_(rescale)
            s.mul_(rescale)
            a += torch.matmul(w, vf[:, :, c0:c1])
            s += w.sum(dim=-1, keepdim=True)
            t.copy_(t_new)
    out = acc / total.clamp_min(1e-30)                    # rows without a valid key: 0 / tiny = 0
    return out.reshape(out_shape).to(out_dtype)
