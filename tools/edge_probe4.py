#!/usr/bin/env python3
"""Fourth exploratory probe: the reference pybind surface (ext.forward / ext.backward, q and k pre-normalised) against float32 math --
3-D merged inputs, single-headed K/V, bias + batch dim, the `should_backwards` flag and the inference return values."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import ext
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(13)
R = lambda *s, dt=torch.float16: torch.randn(*s, device=dev, dtype=dt, generator=g)
nrm = lambda t: torch.nn.functional.normalize(t.float(), dim=-1).to(t.dtype)

def run(name, q, k, v, mask=None, bias=None, bias_batch=False, scale=8.0, causal=False):
    try:
        q, k = nrm(q), nrm(k)
        for t in (q, k, v): t.requires_grad_()
        if bias is not None: bias.requires_grad_()
        o, inv_l, sb = ext.forward(q, k, v, mask, bias, bias_batch, scale, causal)
        do = torch.randn(o.shape, device=dev, dtype=o.dtype, generator=g)
        dq, dk, dv, db = ext.backward(do, o, inv_l, q, k, v, mask, bias, bias_batch, scale, causal)
        qf, kf, vf = (t.detach().float().requires_grad_() for t in (q, k, v))
        bf = bias.detach().float().requires_grad_() if bias is not None else None
        q4, k4, v4 = (t[:, None] if t.dim() == 3 and q.dim() == 3 else t for t in (qf, kf, vf))
        of = F.plain_cosine_sim_attention(q4, k4, v4, mask=mask, attn_bias=bf, attn_bias_batch_dim=bias_batch or q.dim() == 3, scale=scale, causal=causal, l2norm_qk=False)
        if q.dim() == 3: of = of[:, 0]
        of.backward(do.float())
        pairs = [("o", o, of), ("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)] + ([("db", db, bf.grad)] if bias is not None else [])
        print(f"{name:44s} sb={sb} inv_l{tuple(inv_l.shape)} " + "  ".join(f"{n} {(a.float() - b).abs().max().item():.1e}/{b.abs().max().item():.1e}" for n, a, b in pairs), flush=True)
    except Exception as ex:
        print(f"{name:44s} EXC {type(ex).__name__}: {str(ex)[:170]}", flush=True)

run("4-D", R(2, 3, 70, 64), R(2, 3, 90, 64), R(2, 3, 90, 64))
run("4-D causal bf16 D=128", R(2, 3, 130, 128, dt=torch.bfloat16), R(2, 3, 130, 128, dt=torch.bfloat16), R(2, 3, 130, 128, dt=torch.bfloat16), causal=True)
run("3-D merged", R(6, 70, 64), R(6, 90, 64), R(6, 90, 64))
run("3-D merged + mask + bias", R(6, 70, 64), R(6, 90, 64), R(6, 90, 64), mask=torch.rand(6, 90, device=dev, generator=g) > 0.3, bias=0.5 * R(6, 70, 90))
run("single-headed K/V + bias per head", R(2, 3, 70, 64), R(2, 90, 64), R(2, 90, 64), bias=0.5 * R(3, 70, 90))
run("bias with batch dim, f32", R(2, 3, 70, 32, dt=torch.float32), R(2, 3, 90, 32, dt=torch.float32), R(2, 3, 90, 32, dt=torch.float32), bias=0.5 * R(2, 70, 90, dt=torch.float32), bias_batch=True)
q, k, v = nrm(R(1, 2, 20, 64)), nrm(R(1, 2, 30, 64)), R(1, 2, 30, 64)
o, inv_l, sb = ext.forward(q, k, v, None, None, False, 8.0, False)
print("inference (nothing requires grad): should_backwards", sb, "inv_l", tuple(inv_l.shape), "o finite", bool(torch.isfinite(o).all()))
print("debug():", ext.debug()[:100])
