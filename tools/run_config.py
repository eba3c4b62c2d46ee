#!/usr/bin/env python3
"""Run ITERS un-instrumented training steps (or forward calls) of one named shape -- the command rocprofv3 --kernel-trace --stats wraps to
get per-kernel durations of a configuration other than the bench workload.  usage: run_config.py NAME [ITERS] [WARM]   (shapes:
tools/kernel_breakdown.py SHAPES).  Measurement tool."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv, argv = sys.argv[:1], sys.argv[1:]
os.environ.setdefault("ITERS", "1"); os.environ.setdefault("WARM", "0")
import importlib.util
spec = importlib.util.spec_from_file_location("kb", os.path.join(ROOT, "tools", "kernel_breakdown.py"))
import flash_cosine_sim_attention_amd as F

SHAPES = None
src = open(os.path.join(ROOT, "tools", "kernel_breakdown.py")).read()
ns = {"torch": torch}
exec(src[src.index("SHAPES = {"):src.index("sel = sys.argv")], ns)
SHAPES = ns["SHAPES"]
name = argv[0]
iters = int(argv[1]) if len(argv) > 1 else 50
warm = int(argv[2]) if len(argv) > 2 else 20
c = SHAPES[name]
q = torch.randn(c["q"], device="cuda", dtype=c["dtype"]).requires_grad_()
k = torch.randn(c["kv"], device="cuda", dtype=c["dtype"]).requires_grad_()
v = torch.randn(c["kv"], device="cuda", dtype=c["dtype"]).requires_grad_()
do = torch.randn(c["q"], device="cuda", dtype=c["dtype"])
bias = torch.randn(c["q"][1], c["q"][2], c["kv"][-2], device="cuda", dtype=c["dtype"]).requires_grad_() if c.get("bias") else None
mask = (torch.rand((c["q"][0], c["kv"][-2]), device="cuda") > 0.25) if c.get("mask") else None
kw = dict(mask=mask, attn_bias=bias, causal=c["causal"], groups=c["groups"], scale=c.get("scale", 8))
def step():
    q.grad = k.grad = v.grad = None
    if bias is not None: bias.grad = None
    if c.get("fwd_only"):
        with torch.no_grad():
            F.flash_cosine_sim_attention(q, k, v, **kw)
    else:
        F.flash_cosine_sim_attention(q, k, v, **kw).backward(do)
for _ in range(warm): step()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters): step()
e.record(); torch.cuda.synchronize()
print(f"{name}: {s.elapsed_time(e) / iters * 1e3:.1f} us per step ({iters} steps)")
