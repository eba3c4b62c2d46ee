#!/usr/bin/env python3
"""Forward time, constant exponent shift vs the per-row (online) reference, same process: dtype x scale at the C3 shape.
bf16 switches regime at scale * groups > 75, f16 at > 11.  Measurement tool (GPU box)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F

def t_us(fn, it=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3

for D in (64, 128, 32):
    for causal in (True, False):
        for dt, scales in ((torch.bfloat16, (8, 80)), (torch.float16, (8, 16))):
            B, H, N = 4, 8, 4096 if causal else 2048
            q, k, v = (torch.randn(B, H, N, D, device="cuda", dtype=dt) for _ in range(3))
            row = []
            with torch.no_grad():
                for sc in scales:
                    row.append((sc, t_us(lambda: F.flash_cosine_sim_attention(q, k, v, causal=causal, scale=sc))))
            print(f"D{D} N{N} causal={int(causal)} {str(dt)[6:]:9s}" + "".join(f"   scale {sc:3d}: {us:7.1f} us" for sc, us in row)
                  + f"   online / constant = {row[1][1] / row[0][1]:.3f}")
