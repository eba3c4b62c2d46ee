"""Loads the compiled PyTorch binding (`_fcsa_torch.so`, csrc/fcsa_torch.cpp) and registers what the dispatcher needs besides
the kernels: fake (meta) implementations so that `torch.compile` can trace `torch.ops.fcsa.forward / backward` without running
them.  The binding is host-only C++ over the C ABI of libfcsa_hip.so (include/fcsa.h); it replaces the reference's pybind
module (flash_cosine_sim_attention_cuda.cu:1928-1933).  There is no fallback: a missing binding raises ImportError."""
from __future__ import annotations

import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
BINDING_PATH = os.path.join(_HERE, "_fcsa_torch.so")
_loaded = False


def load():
    global _loaded
    if _loaded:
        return torch.ops.fcsa
    if not os.path.exists(BINDING_PATH):
        raise ImportError(f"{BINDING_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or `make -C flash_cosine_sim_attention_amd/csrc`).  There is no non-HIP fallback for GPU tensors.")
    torch.ops.load_library(BINDING_PATH)
    alt = os.environ.get("FCSA_LIB")          # measurement only: route the ops to another build of the same C ABI (A/B runs)
    if alt:
        import ctypes
        if ctypes.CDLL(BINDING_PATH).fcsa_torch_use_library(alt.encode()) != 0:
            raise ImportError(f"FCSA_LIB={alt}: not a loadable build of libfcsa_hip.so")
    _register_fakes()
    _loaded = True
    return torch.ops.fcsa


def _canon_dims(q, k):
    merged = q.dim() == 3
    B, H, N, D = (q.shape[0], 1, q.shape[1], q.shape[2]) if merged else q.shape
    Hk, M = (1, k.shape[1]) if k.dim() == 3 else (k.shape[1], k.shape[2])
    return merged, B, H, Hk, N, M, D


def _register_fakes():
    @torch.library.register_fake("fcsa::forward")
    def _(q, k, v, mask, attn_bias, attn_bias_batch_dim, scale, causal, l2norm_qk, groups, need_backward):
        merged, B, H, Hk, N, M, D = _canon_dims(q, k)
        f32 = dict(device=q.device, dtype=torch.float32)
        o = q.new_empty(q.shape)
        none32, none = q.new_empty((0,), dtype=torch.float32), q.new_empty((0,))
        inv_l = torch.empty((B, H, N), **f32) if need_backward else none32
        # (an inference call of the 16-bit kernels saves no normalised q: fcsa_forward_needs_qn, include/fcsa.h)
        blocks = (D // groups) // 8          # fcsa_capi.hip log2_blocks_per_group: 8 * 2^k features per group, or ONE group of any width (D = 96)
        fusable = D % groups == 0 and (D // groups) % 8 == 0 and (blocks & (blocks - 1) == 0 or groups == 1)
        qn = q.new_empty((B, H, N, D)) if (l2norm_qk and (need_backward or q.dtype == torch.float32 or not fusable)) else none
        kn = q.new_empty((B, Hk, M, D)) if l2norm_qk else none
        rq = torch.empty((B, H, N, groups), **f32) if (l2norm_qk and need_backward) else none32
        rk = torch.empty((B, Hk, M, groups), **f32) if (l2norm_qk and need_backward) else none32
        return o, inv_l, qn, kn, rq, rk

    @torch.library.register_fake("fcsa::attention")
    def _(q, k, v, mask, attn_bias, attn_bias_batch_dim, scale, causal, l2norm_qk, groups):
        return q.new_empty(q.shape)

    @torch.library.register_fake("fcsa::backward")
    def _(d_out, o, inv_l, q, k, v, mask, attn_bias, qn, kn, rq, rk, attn_bias_batch_dim, scale, causal, l2norm_qk, groups,
          need_bias_grad):
        db = attn_bias.new_empty(attn_bias.shape) if (attn_bias is not None and need_bias_grad) else q.new_empty((0,))
        return q.new_empty(q.shape), k.new_empty(k.shape), v.new_empty(v.shape), db
