#!/usr/bin/env python3
"""Why did bench.py's C2_bias line read 0.67 ms in the round-6 evidence session when its kernels sum to 0.134 ms?  Wall time per step of
the same call (events and host clock), repeated.  Measurement tool."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import flash_cosine_sim_attention_amd as F
c = bench.EXTRA["C2_bias"] if hasattr(bench, "EXTRA") else None
for name in dir(bench):
    v = getattr(bench, name)
    if isinstance(v, dict) and "C2_bias" in v:
        c = v["C2_bias"]
print(c)
for rep in range(3):
    print(rep, bench.run_extra_config(F, c, sdpa=False))
dt = torch.float16
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn((4, 8, 1024, 64), device="cuda", dtype=dt, generator=g).requires_grad_() for _ in range(3))
do = torch.randn((4, 8, 1024, 64), device="cuda", dtype=dt, generator=g)
bias = (0.5 * torch.randn((8, 1024, 1024), device="cuda", dtype=dt, generator=g)).requires_grad_()
def fb(set_none=True):
    q.grad = k.grad = v.grad = None
    if set_none: bias.grad = None
    F.flash_cosine_sim_attention(q, k, v, attn_bias=bias).backward(do)
for mode in (True, False):
    for _ in range(10): fb(mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fb(mode)
    torch.cuda.synchronize(); print("bias.grad=None each step" if mode else "bias.grad accumulates", (time.perf_counter() - t0) / 50 * 1e6, "us per step")
