#!/usr/bin/env python3
"""Per-config timings for BASELINE.json configs C2-C5: fused op vs torch SDPA on the same box.
Protocol of the reference (flash_cosine_sim_attention/benchmark.py:7-56): 10 warm-ups, mean of 20 event-timed runs.
Writes gpurun_out/configs.json and prints a table.  Measurement tool (not part of the product path)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F

CFG = {
    "C2": dict(q=(4, 8, 1024, 64), kv=(4, 8, 1024, 64), dtype=torch.float16, causal=False, mask=False, scale=8, groups=1, bwd=False),
    "C3": dict(q=(4, 8, 4096, 64), kv=(4, 8, 4096, 64), dtype=torch.bfloat16, causal=True, mask=False, scale=8, groups=1, bwd=True),
    "C4": dict(q=(1, 8, 1024, 64), kv=(1, 8, 8192, 64), dtype=torch.float16, causal=False, mask=True, scale=8, groups=1, bwd=True),
    "C5": dict(q=(4, 8, 2048, 128), kv=(4, 2048, 128), dtype=torch.bfloat16, causal=True, mask=False, scale=1, groups=8, bwd=True),
    "C3-f16-noncausal": dict(q=(4, 8, 4096, 64), kv=(4, 8, 4096, 64), dtype=torch.float16, causal=False, mask=False, scale=8, groups=1, bwd=True),
    "C2-bias": dict(q=(4, 8, 1024, 64), kv=(4, 8, 1024, 64), dtype=torch.float16, causal=False, mask=False, scale=8, groups=1, bwd=True, bias=True),
    "C3-d128": dict(q=(4, 8, 4096, 128), kv=(4, 8, 4096, 128), dtype=torch.bfloat16, causal=True, mask=False, scale=8, groups=1, bwd=True),
}

def frac(n, m, causal):
    if not causal: return 1.0
    d = m - n
    return sum(min(m, max(0, i + d + 1)) for i in range(n)) / float(n * m)

def timeit(fn, iters=20, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

out = {}
sel = sys.argv[1:] or list(CFG)
for name in sel:
    c = CFG[name]
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(c["q"], device="cuda", dtype=c["dtype"], generator=g).requires_grad_(c["bwd"])
    k = torch.randn(c["kv"], device="cuda", dtype=c["dtype"], generator=g).requires_grad_(c["bwd"])
    v = torch.randn(c["kv"], device="cuda", dtype=c["dtype"], generator=g).requires_grad_(c["bwd"])
    mask = (torch.rand((c["q"][0], c["kv"][-2]), device="cuda", generator=g) > 0.25) if c["mask"] else None
    do = torch.randn(c["q"], device="cuda", dtype=c["dtype"], generator=g)
    B, H, N, D = c["q"]; M = c["kv"][-2]
    unit = B * H * N * M * D * frac(N, M, c["causal"])
    bias = None
    if c.get("bias"):        # per-head bias with gradient (d_bias): reference test grid item, tests/test.py:32
        bias = (0.5 * torch.randn((H, N, M), device="cuda", dtype=c["dtype"], generator=g)).requires_grad_()
    kw = dict(mask=mask, attn_bias=bias, causal=c["causal"], scale=c["scale"], groups=c["groups"])
    def fwd():
        with torch.no_grad(): return F.flash_cosine_sim_attention(q, k, v, **kw)
    def fb():
        q.grad = k.grad = v.grad = None
        if bias is not None: bias.grad = None
        F.flash_cosine_sim_attention(q, k, v, **kw).backward(do)
    t_f = timeit(fwd)
    r = dict(fwd_ms=round(t_f, 4), fwd_tflops=round(4 * unit / t_f / 1e9, 1))
    if c["bwd"]:
        t_fb = timeit(fb)
        r.update(fwdbwd_ms=round(t_fb, 4), fwdbwd_tflops=round(14 * unit / t_fb / 1e9, 1))
    # torch SDPA (softmax attention) on the same shapes, same protocol
    try:
        if os.environ.get('NO_SDPA'): raise RuntimeError('sdpa skipped')
        ke, ve = (k, v) if k.dim() == 4 else (k[:, None].expand(B, H, M, D), v[:, None].expand(B, H, M, D))
        am = None if mask is None else mask[:, None, None, :].expand(B, 1, N, M)
        if bias is not None: am = bias.detach()[None].expand(B, H, N, M)          # additive float mask (no bias gradient in SDPA)
        def sfwd():
            with torch.no_grad(): return torch.nn.functional.scaled_dot_product_attention(q, ke, ve, attn_mask=am, is_causal=c["causal"])
        def sfb():
            q.grad = k.grad = v.grad = None
            torch.nn.functional.scaled_dot_product_attention(q, ke, ve, attn_mask=am, is_causal=c["causal"]).backward(do)
        ts = timeit(sfwd); r.update(sdpa_fwd_ms=round(ts, 4), fwd_speedup_vs_sdpa=round(ts / t_f, 2))
        if c["bwd"]:
            tsb = timeit(sfb); r.update(sdpa_fwdbwd_ms=round(tsb, 4), fwdbwd_speedup_vs_sdpa=round(tsb / t_fb, 2))
    except Exception as ex:
        r["sdpa_error"] = str(ex)[:120]
    out[name] = r
    print(name, r, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)
