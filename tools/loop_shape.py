#!/usr/bin/env python3
"""fwd+bwd loop on one shape for rocprofv3 (usage: loop_shape.py B H N D dtype causal groups singlekv iters)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
B, H, N, D = map(int, sys.argv[1:5]); dt = getattr(torch, sys.argv[5]); causal = bool(int(sys.argv[6])); groups = int(sys.argv[7])
skv = bool(int(sys.argv[8])); iters = int(sys.argv[9])
q = torch.randn(B, H, N, D, device="cuda", dtype=dt).requires_grad_()
kvs = (B, N, D) if skv else (B, H, N, D)
k = torch.randn(kvs, device="cuda", dtype=dt).requires_grad_(); v = torch.randn(kvs, device="cuda", dtype=dt).requires_grad_()
do = torch.randn(B, H, N, D, device="cuda", dtype=dt)
def step():
    q.grad = k.grad = v.grad = None
    F.flash_cosine_sim_attention(q, k, v, causal=causal, groups=groups).backward(do)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(iters): step()
torch.cuda.synchronize(); print("ms/step wall", (time.time() - t0) / iters * 1e3)
