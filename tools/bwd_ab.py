#!/usr/bin/env python3
"""Backward-kernel times for a list of shapes (library HIP events).  Run once per library (FCSA_LIB=...) to A/B."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import _lib
SH = [(4, 8, 4096, 64, True), (4, 8, 4096, 32, True), (4, 8, 4096, 64, False), (4, 8, 4096, 32, False), (2, 8, 8192, 64, True),
      (8, 16, 1024, 64, False), (8, 16, 1024, 64, True), (16, 16, 512, 64, True)]
for (B, H, N, D, causal) in SH:
    q, k, v = (torch.randn(B, H, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    do = torch.randn(B, H, N, D, device="cuda", dtype=torch.bfloat16)
    def step():
        q.grad = k.grad = v.grad = None
        F.flash_cosine_sim_attention(q, k, v, causal=causal).backward(do)
    for _ in range(5): step()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(20): step()
    torch.cuda.synchronize()
    st = {s["name"]: s["total_ms"] / s["calls"] * 1e3 for s in _lib.profile_collect()}
    _lib.profile_enable(False)
    unit = B * H * N * N * D * (0.5 if causal else 1.0)
    print(f"B{B} H{H} N{N} D{D} causal={int(causal)}: dkv {st['bwd_dkv']:8.1f} us {8*unit/st['bwd_dkv']/1e6:7.1f} TF   dq {st['bwd_dq']:8.1f} us   fwd {st['fwd']:8.1f} us")
