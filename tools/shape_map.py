#!/usr/bin/env python3
"""Efficiency map over a grid of shapes (round 6): algorithmic TFLOP/s of the forward and of forward + backward for
batch*heads x sequence length x head dim x causal, bf16, self-attention -- to find shapes where the dispatch (forms, split counts,
last-round rule) leaves a hole next to its neighbours.  Protocol: 5 warm-ups, mean of ITERS event-timed calls.
usage: shape_map.py [--dims 64,128] [--single-kv]         (measurement tool, not part of the product path)"""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F

ap = argparse.ArgumentParser()
ap.add_argument("--dims", default="64,128")
ap.add_argument("--bh", default="4,8,16,32,64,128,256")
ap.add_argument("--seq", default="256,512,1024,2048,4096,8192,16384")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--single-kv", action="store_true")
ap.add_argument("--max-gflop", type=float, default=4000.0, help="skip shapes whose forward + backward exceeds this")
a = ap.parse_args()
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]


def timeit(fn, iters):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for D in (int(x) for x in a.dims.split(",")):
    for causal in (True, False):
        print(f"== D {D} {a.dtype} causal={int(causal)}{' single-headed K/V' if a.single_kv else ''}: forward | forward + backward, algorithmic TFLOP/s (ms); rows = batch x heads, columns = N")
        seqs = [int(x) for x in a.seq.split(",")]
        print("   B x H   " + "".join(f"{n:>26d}" for n in seqs))
        for bh in (int(x) for x in a.bh.split(",")):
            H = min(bh, 8); B = bh // H
            cells = []
            for N in seqs:
                unit = B * H * N * N * D * ((N + 1) / (2.0 * N) if causal else 1.0)
                if 14 * unit / 1e9 > a.max_gflop or B * H * N * D * 2 * 12 > 8e9:
                    cells.append(f"{'-':>26s}"); continue
                q = torch.randn(B, H, N, D, device="cuda", dtype=dt, requires_grad=True)
                ks = (B, N, D) if a.single_kv else (B, H, N, D)
                k = torch.randn(ks, device="cuda", dtype=dt, requires_grad=True)
                v = torch.randn(ks, device="cuda", dtype=dt, requires_grad=True)
                do = torch.randn(B, H, N, D, device="cuda", dtype=dt)
                def fwd():
                    with torch.no_grad(): return F.flash_cosine_sim_attention(q, k, v, causal=causal)
                def fb():
                    q.grad = k.grad = v.grad = None
                    F.flash_cosine_sim_attention(q, k, v, causal=causal).backward(do)
                iters = 30 if unit < 2e10 else 10
                tf, tb = timeit(fwd, iters), timeit(fb, iters)
                cells.append(f"{4 * unit / tf / 1e9:7.0f} | {14 * unit / tb / 1e9:5.0f} ({tb:6.3f})")
                del q, k, v, do
            print(f"   {B:3d} x {H:<3d} " + "".join(f"{c:>26s}" for c in cells), flush=True)
