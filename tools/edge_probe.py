#!/usr/bin/env python3
"""Exploratory probe of API-level edge cases against the package's plain_cosine_sim_attention (f32 math on the same inputs): prints
max-abs deltas; anything surprising becomes a test.  (measurement / triage tool, not part of the product path)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flash_cosine_sim_attention_amd as F

def ref(q, k, v, **kw):
    qf, kf, vf = (t.detach().float().requires_grad_() for t in (q, k, v))
    b = kw.get("attn_bias")
    kw2 = dict(kw)
    if b is not None: kw2["attn_bias"] = b.detach().float()
    o = F.plain_cosine_sim_attention(qf, kf, vf, **kw2)
    return o, (qf, kf, vf)

def case(name, q, k, v, do=None, **kw):
    try:
        for t in (q, k, v): t.requires_grad_()
        o = F.flash_cosine_sim_attention(q, k, v, **kw)
        do = torch.randn_like(o) if do is None else do
        o.backward(do)
        orf, (qf, kf, vf) = ref(q, k, v, **kw)
        orf.backward(do.float())
        d = [(o.float() - orf).abs().max().item()] + [(a.grad.float() - b.grad).abs().max().item() for a, b in zip((q, k, v), (qf, kf, vf))]
        print(f"{name:55s} max|d| o/dq/dk/dv = " + " ".join(f"{x:.2e}" for x in d), flush=True)
    except Exception as ex:
        print(f"{name:55s} EXC {type(ex).__name__}: {str(ex)[:160]}", flush=True)

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(5)
R = lambda *s, dt=torch.float16: torch.randn(*s, device=dev, dtype=dt, generator=g)
case("scale=0 (uniform attention)", R(1, 2, 100, 64), R(1, 2, 130, 64), R(1, 2, 130, 64), scale=0.0)
case("groups = D (every feature its own group) D=16", R(1, 2, 70, 16), R(1, 2, 90, 16), R(1, 2, 90, 16), groups=16, scale=1.0)
case("groups = D/2 D=32 (pairs)", R(1, 2, 70, 32), R(1, 2, 90, 32), R(1, 2, 90, 32), groups=16, scale=1.0)
case("merged batch*heads 3-D q,k,v", R(6, 100, 64), R(6, 120, 64), R(6, 120, 64))
case("merged 3-D causal", R(6, 130, 64), R(6, 130, 64), R(6, 130, 64), causal=True)
m = torch.rand(2, 260, device=dev, generator=g) > 0.4
case("non-contiguous mask view", R(2, 3, 100, 64), R(2, 3, 130, 64), R(2, 3, 130, 64), mask=m[:, ::2])
case("mask with a fully masked batch row", R(2, 3, 50, 64), R(2, 3, 70, 64), R(2, 3, 70, 64), mask=torch.stack([torch.ones(70, dtype=torch.bool, device=dev), torch.zeros(70, dtype=torch.bool, device=dev)]))
case("long N tiny D (1,1,20000,16) causal bf16", R(1, 1, 20000, 16, dt=torch.bfloat16), R(1, 1, 20000, 16, dt=torch.bfloat16), R(1, 1, 20000, 16, dt=torch.bfloat16), causal=True)
q = R(2, 3, 100, 64); case("dO expanded ones (sum().backward())", q, R(2, 3, 130, 64), R(2, 3, 130, 64), do=torch.ones(1, device=dev, dtype=torch.float16).expand(2, 3, 100, 64))
case("q sliced in the last dim (misaligned base)", R(2, 3, 100, 72)[..., 4:68], R(2, 3, 130, 64), R(2, 3, 130, 64))
case("k, v as transposed views b n h d", R(2, 3, 100, 64), R(2, 130, 3, 64).transpose(1, 2), R(2, 130, 3, 64).transpose(1, 2))
case("f32 bias batch dim + causal", R(2, 3, 100, 64, dt=torch.float32), R(2, 3, 100, 64, dt=torch.float32), R(2, 3, 100, 64, dt=torch.float32), attn_bias=R(2, 100, 100, dt=torch.float32), attn_bias_batch_dim=True, causal=True)
case("scale 100 f16 (online shift, saturated softmax)", R(1, 2, 100, 64), R(1, 2, 130, 64), R(1, 2, 130, 64), scale=100.0)
case("no l2norm, q,k pre-normalised, scale 10", torch.nn.functional.normalize(R(1, 2, 100, 64).float(), dim=-1).half(), torch.nn.functional.normalize(R(1, 2, 130, 64).float(), dim=-1).half(), R(1, 2, 130, 64), l2norm_qk=False, scale=10.0)
for bad, kw in [("float64 q", dict(q=torch.randn(1, 2, 8, 64, device=dev, dtype=torch.float64))), ("D=48", dict(q=R(1, 2, 8, 48), k=R(1, 2, 8, 48), v=R(1, 2, 8, 48))),
                ("mask + causal", dict(mask=torch.ones(1, 8, dtype=torch.bool, device=dev), causal=True)), ("int mask", dict(mask=torch.ones(1, 8, dtype=torch.int32, device=dev))),
                ("groups not dividing D", dict(groups=5)), ("cpu k", dict(k=torch.randn(1, 2, 8, 64, dtype=torch.float16)))]:
    a = dict(q=R(1, 2, 8, 64), k=R(1, 2, 8, 64), v=R(1, 2, 8, 64)); a.update({x: y for x, y in kw.items() if x in ("q", "k", "v")})
    rest = {x: y for x, y in kw.items() if x not in ("q", "k", "v")}
    try:
        F.flash_cosine_sim_attention(a["q"], a["k"], a["v"], **rest); print(f"{bad:55s} NO ERROR RAISED")
    except Exception as ex:
        print(f"{bad:55s} raises {type(ex).__name__}: {str(ex)[:110]}")
