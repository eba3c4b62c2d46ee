#!/usr/bin/env python3
"""Per-shape step diagnosis: wall time per forward + backward call next to the kernels the library launched for it (names, calls per
step, HIP-event time) -- separates host-bound steps from launch-count effects.  usage: step_diag.py B,H,N,D,causal[,single_kv] ...
(measurement tool, not part of the product path)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import _lib
for spec in sys.argv[1:]:
    B, H, N, D, causal, *rest = (int(x) for x in spec.split(","))
    single = bool(rest and rest[0])
    dt = torch.bfloat16
    q = torch.randn(B, H, N, D, device="cuda", dtype=dt, requires_grad=True)
    ks = (B, N, D) if single else (B, H, N, D)
    k = torch.randn(ks, device="cuda", dtype=dt, requires_grad=True)
    v = torch.randn(ks, device="cuda", dtype=dt, requires_grad=True)
    do = torch.randn(B, H, N, D, device="cuda", dtype=dt)
    def fb():
        q.grad = k.grad = v.grad = None
        F.flash_cosine_sim_attention(q, k, v, causal=bool(causal)).backward(do)
    for _ in range(10): fb()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): fb()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 50 * 1e6
    _lib.profile_enable(True)
    for _ in range(20): fb()
    torch.cuda.synchronize()
    st = _lib.profile_collect()
    _lib.profile_enable(False)
    print(f"{spec}: wall {wall:7.1f} us/step | " + "  ".join(f"{s['name']} x{s['calls'] / 20:g} {s['total_ms'] / 20 * 1e3:6.1f}" for s in st), flush=True)
