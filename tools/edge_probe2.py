#!/usr/bin/env python3
"""Second exploratory probe (see edge_probe.py): zero / tiny-norm rows, -inf and huge biases, single positions, many tiny problems."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flash_cosine_sim_attention_amd as F
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(7)
R = lambda *s, dt=torch.float16: torch.randn(*s, device=dev, dtype=dt, generator=g)

def case(name, q, k, v, do=None, bias=None, **kw):
    try:
        q, k, v = (t.detach().clone().requires_grad_() for t in (q, k, v))
        if bias is not None: bias = bias.detach().clone().requires_grad_(); kw["attn_bias"] = bias
        o = F.flash_cosine_sim_attention(q, k, v, **kw)
        do = torch.randn(o.shape, device=dev, dtype=o.dtype, generator=g) if do is None else do
        o.backward(do)
        qf, kf, vf = (t.detach().float().requires_grad_() for t in (q, k, v))
        kw2 = dict(kw)
        if bias is not None: bf = bias.detach().float().requires_grad_(); kw2["attn_bias"] = bf
        of = F.plain_cosine_sim_attention(qf, kf, vf, **kw2)
        of.backward(do.float())
        pairs = [("o", o, of), ("dq", q.grad, qf.grad), ("dk", k.grad, kf.grad), ("dv", v.grad, vf.grad)] + ([("db", bias.grad, bf.grad)] if bias is not None else [])
        msg = []
        for nm, a, b in pairs:
            a = a.float()
            fin_a, fin_b = torch.isfinite(a), torch.isfinite(b)
            both = fin_a & fin_b
            d = (a[both] - b[both]).abs().max().item() if both.any() else 0.0
            msg.append(f"{nm} {d:.1e}/{b[fin_b].abs().max().item() if fin_b.any() else 0:.1e}" + (f" nonfinite ours {int((~fin_a).sum())} ref {int((~fin_b).sum())}" if (~fin_a).any() or (~fin_b).any() else ""))
        print(f"{name:52s} " + "  ".join(msg), flush=True)
    except Exception as ex:
        print(f"{name:52s} EXC {type(ex).__name__}: {str(ex)[:160]}", flush=True)

q = R(1, 2, 40, 64); q[0, 0, 3] = 0; q[0, 1, 7] = 0
case("two all-zero q rows (f16)", q, R(1, 2, 50, 64), R(1, 2, 50, 64))
k = R(1, 2, 50, 64, dt=torch.float32); k[0, 0, 5] = 0
case("an all-zero k row (f32)", R(1, 2, 40, 64, dt=torch.float32), k, R(1, 2, 50, 64, dt=torch.float32))
case("tiny-norm q rows 1e-3 (bf16)", 1e-3 * R(1, 2, 40, 64, dt=torch.bfloat16), R(1, 2, 50, 64, dt=torch.bfloat16), R(1, 2, 50, 64, dt=torch.bfloat16))
case("huge-norm k rows 1e3 (f16)", R(1, 2, 40, 64), 300.0 * R(1, 2, 50, 64), R(1, 2, 50, 64))
b = 0.5 * R(2, 40, 50); b[:, :, 10:20] = float("-inf")
case("bias with a -inf band (f16)", R(1, 2, 40, 64), R(1, 2, 50, 64), R(1, 2, 50, 64), bias=b)
b = 0.5 * R(2, 40, 50, dt=torch.float32); b[0, 3, :] = float("-inf")
case("bias with a whole row -inf (f32)", R(1, 2, 40, 64, dt=torch.float32), R(1, 2, 50, 64, dt=torch.float32), R(1, 2, 50, 64, dt=torch.float32), bias=b)
case("bias +-3000 (f16, online shift)", R(1, 2, 40, 64), R(1, 2, 50, 64), R(1, 2, 50, 64), bias=3000.0 * torch.sign(R(2, 40, 50)))
case("N = M = 1", R(2, 3, 1, 64), R(2, 3, 1, 64), R(2, 3, 1, 64))
case("N = M = 1 causal D=128 bf16", R(2, 3, 1, 128, dt=torch.bfloat16), R(2, 3, 1, 128, dt=torch.bfloat16), R(2, 3, 1, 128, dt=torch.bfloat16), causal=True)
case("40000 tiny problems (B=5000,H=8,N=M=8) bf16", R(5000, 8, 8, 32, dt=torch.bfloat16), R(5000, 8, 8, 32, dt=torch.bfloat16), R(5000, 8, 8, 32, dt=torch.bfloat16), causal=True)
case("H = 1 (kv_heads == heads == 1)", R(3, 1, 70, 64), R(3, 1, 90, 64), R(3, 1, 90, 64))
case("v with 1e4 entries (f16)", R(1, 2, 40, 64), R(1, 2, 50, 64), 1e4 * torch.sign(R(1, 2, 50, 64)))
case("dO = 0", R(1, 2, 40, 64), R(1, 2, 50, 64), R(1, 2, 50, 64), do=torch.zeros(1, 2, 40, 64, device=dev, dtype=torch.float16))
case("scale 1e-3", R(1, 2, 40, 64), R(1, 2, 50, 64), R(1, 2, 50, 64), scale=1e-3)
case("scale -0.0", R(1, 2, 40, 64), R(1, 2, 50, 64), R(1, 2, 50, 64), scale=-0.0)
case("groups = D/4 (D = 64, groups 16: 4-feature groups)", R(1, 2, 40, 64), R(1, 2, 50, 64), R(1, 2, 50, 64), groups=16, scale=1.0)
