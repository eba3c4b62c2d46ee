#!/usr/bin/env python3
"""Forward time, constant exponent shift vs the per-row (online) reference, same process: dtype x scale at the C3 shape.
bf16 switches regime at scale * groups > 75, f16 at > 11.  Measurement tool (GPU box)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import _lib


def kernel_us(fn, it=20):
    """average duration of the forward kernel itself (HIP events recorded by the library around its launch)"""
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(it): fn()
    torch.cuda.synchronize()
    st = {s["name"]: s["total_ms"] / s["calls"] * 1e3 for s in _lib.profile_collect()}
    _lib.profile_enable(False)
    return st["fwd"]

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
                    fn = lambda: F.flash_cosine_sim_attention(q, k, v, causal=causal, scale=sc)
                    row.append((sc, t_us(fn), kernel_us(fn)))
            print(f"D{D} N{N} causal={int(causal)} {str(dt)[6:]:9s}" + "".join(f"   scale {sc:3d}: op {us:6.1f} us, kernel {ku:6.1f}" for sc, us, ku in row)
                  + f"   online / constant: op {row[1][1] / row[0][1]:.3f}, kernel {row[1][2] / row[0][2]:.3f}")
