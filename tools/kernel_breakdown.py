#!/usr/bin/env python3
"""Per-kernel time breakdown (library HIP events) for a few shapes.  Measurement tool.

Regime (state it wherever these numbers are quoted): every launch is bracketed by a HIP-event pair recorded by the library
(fcsa_profile_enable), after WARM (default 50) un-instrumented warm-up steps, mean of ITERS (default 30) steps.  The event pairs
serialise the launches and add ~1.5 - 2.5 us per kernel; with only a handful of warm-up steps (round 5 used 5) the chip is also still
ramping its clocks, which is why round 5's table read 15 - 25 % above the rocprofv3 --kernel-trace durations of the same kernels.
Use these numbers to compare kernels of ONE table; quote absolute kernel times from the rocprofv3 kernel trace."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import _lib

SHAPES = {
    "d64":   dict(q=(4, 8, 4096, 64), kv=(4, 8, 4096, 64), dtype=torch.bfloat16, causal=True, groups=1),
    "d128":  dict(q=(4, 8, 4096, 128), kv=(4, 8, 4096, 128), dtype=torch.bfloat16, causal=True, groups=1),
    "d128nc": dict(q=(4, 8, 2048, 128), kv=(4, 8, 2048, 128), dtype=torch.bfloat16, causal=False, groups=1),
    "C5":    dict(q=(4, 8, 2048, 128), kv=(4, 2048, 128), dtype=torch.bfloat16, causal=True, groups=8, scale=1),
    "C4":    dict(q=(1, 8, 1024, 64), kv=(1, 8, 8192, 64), dtype=torch.float16, causal=False, groups=1, mask=True),
    "C2":    dict(q=(4, 8, 1024, 64), kv=(4, 8, 1024, 64), dtype=torch.float16, causal=False, groups=1, fwd_only=True),
    "C5s8":  dict(q=(4, 8, 2048, 128), kv=(4, 2048, 128), dtype=torch.bfloat16, causal=True, groups=8, scale=8),
    "d32":   dict(q=(4, 8, 4096, 32), kv=(4, 8, 4096, 32), dtype=torch.bfloat16, causal=True, groups=1),
    "d96":   dict(q=(4, 8, 4096, 96), kv=(4, 8, 4096, 96), dtype=torch.bfloat16, causal=True, groups=1),
    "d64f32": dict(q=(2, 8, 2048, 64), kv=(2, 8, 2048, 64), dtype=torch.float32, causal=True, groups=1),
    "C2bias": dict(q=(4, 8, 1024, 64), kv=(4, 8, 1024, 64), dtype=torch.float16, causal=False, groups=1, bias=True),
    # per-row exponent reference (online) regimes: f16 beyond scale * groups = 11, bf16 beyond 75
    "f16s16": dict(q=(4, 8, 4096, 64), kv=(4, 8, 4096, 64), dtype=torch.float16, causal=True, groups=1, scale=16),
    "f16s16d128": dict(q=(4, 8, 4096, 128), kv=(4, 8, 4096, 128), dtype=torch.float16, causal=True, groups=1, scale=16),
    "C5s16": dict(q=(4, 8, 2048, 128), kv=(4, 2048, 128), dtype=torch.bfloat16, causal=True, groups=8, scale=16),
    "T5bias": dict(q=(8, 12, 512, 64), kv=(8, 12, 512, 64), dtype=torch.bfloat16, causal=False, groups=1, bias=True),
}
sel = sys.argv[1:] or list(SHAPES)
def run_shape(name):
    c = SHAPES[name]
    q = torch.randn(c["q"], device="cuda", dtype=c["dtype"]).requires_grad_()
    k = torch.randn(c["kv"], device="cuda", dtype=c["dtype"]).requires_grad_()
    v = torch.randn(c["kv"], device="cuda", dtype=c["dtype"]).requires_grad_()
    do = torch.randn(c["q"], device="cuda", dtype=c["dtype"])
    bias = torch.randn(c["q"][1], c["q"][2], c["kv"][-2], device="cuda", dtype=c["dtype"]).requires_grad_() if c.get("bias") else None
    mask = (torch.rand((c["q"][0], c["kv"][-2]), device="cuda") > 0.25) if c.get("mask") else None
    def step():
        q.grad = k.grad = v.grad = None
        if bias is not None: bias.grad = None
        if c.get("fwd_only"):
            with torch.no_grad():
                F.flash_cosine_sim_attention(q, k, v, mask=mask, attn_bias=bias, causal=c["causal"], groups=c["groups"], scale=c.get("scale", 8))
            return
        F.flash_cosine_sim_attention(q, k, v, mask=mask, attn_bias=bias, causal=c["causal"], groups=c["groups"], scale=c.get("scale", 8)).backward(do)
    iters = int(os.environ.get("ITERS", "30"))
    for _ in range(int(os.environ.get("WARM", "50"))): step()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(iters): step()
    torch.cuda.synchronize()
    st = _lib.profile_collect()
    _lib.profile_enable(False)
    B, H, N, D = c["q"]; M = c["kv"][-2]
    unit = B * H * N * M * D * (0.5 if c["causal"] else 1.0)
    print(name, c["q"], "unit GFLOP", round(unit / 1e9, 2))
    for s in st:
        us = s["total_ms"] / s["calls"] * 1e3
        mult = {"fwd": 4, "bwd_dq": 2, "bwd_dkv": 8}.get(s["name"], 0)
        print(f"   {s['name']:<20} calls/step {s['calls']/iters:.0f}  avg {us:8.1f} us" + (f"   {mult*unit/us/1e6:7.1f} TF" if mult else ""))


print("regime: HIP-event pair around every launch (library hook), WARM=%s warm-up steps, mean of ITERS=%s; not comparable with rocprofv3 kernel-trace durations"
      % (os.environ.get("WARM", "50"), os.environ.get("ITERS", "30")))
for name in sel:
    try:
        run_shape(name)
    except Exception as e:      # e.g. an older library (FCSA_LIB) that refuses the configuration
        print(name, "failed:", str(e)[:200])
