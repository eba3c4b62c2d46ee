#!/usr/bin/env python3
"""Forward-kernel time for a list of shapes (library HIP events).  Run once per library (FCSA_LIB=...) to A/B."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import _lib
SH = [(4, 8, 4096, 16, True), (4, 8, 4096, 32, True), (4, 8, 4096, 64, True), (4, 8, 4096, 96, True),
      (4, 8, 4096, 64, False), (4, 8, 2048, 64, False), (8, 16, 1024, 64, False), (8, 16, 1024, 64, True), (2, 8, 8192, 64, True),
      (4, 8, 4096, 32, False), (4, 8, 4096, 96, False), (16, 16, 512, 64, True)]
for dt in (torch.bfloat16,):
    for (B, H, N, D, causal) in SH:
        q, k, v = (torch.randn(B, H, N, D, device="cuda", dtype=dt) for _ in range(3))
        for _ in range(5): F.flash_cosine_sim_attention(q, k, v, causal=causal)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(20): F.flash_cosine_sim_attention(q, k, v, causal=causal)
        torch.cuda.synchronize()
        st = {s["name"]: s["total_ms"] / s["calls"] * 1e3 for s in _lib.profile_collect()}
        _lib.profile_enable(False)
        unit = B * H * N * N * D * (0.5 if causal else 1.0)
        print(f"B{B} H{H} N{N} D{D} causal={int(causal)}: fwd {st['fwd']:8.1f} us  {4*unit/st['fwd']/1e6:7.1f} TF")
