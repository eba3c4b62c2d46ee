#!/usr/bin/env python3
"""Third exploratory probe (see edge_probe.py): merged batch*heads inputs with mask / bias, backward twice, gradient accumulation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flash_cosine_sim_attention_amd as F
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(9)
R = lambda *s, dt=torch.float16: torch.randn(*s, device=dev, dtype=dt, generator=g)

def cmp(name, outs, refs):
    print(f"{name:50s} " + "  ".join(f"{(a.float() - b.float()).abs().max().item():.1e}/{b.float().abs().max().item():.1e}" for a, b in zip(outs, refs)), flush=True)

# merged [BH, N, D] with a mask [BH, M] and a bias [BH, N, M] (bias_batch forced true, cu:1652)
BH, N, M, D = 6, 70, 90, 64
q, k, v = R(BH, N, D).requires_grad_(), R(BH, M, D).requires_grad_(), R(BH, M, D).requires_grad_()
mask = torch.rand(BH, M, device=dev, generator=g) > 0.3; mask[:, 0] = True
bias = (0.5 * R(BH, N, M)).requires_grad_()
do = R(BH, N, D)
try:
    o = F.flash_cosine_sim_attention(q, k, v, mask=mask, attn_bias=bias)
    o.backward(do)
    qf, kf, vf, bf = (t.detach().float().requires_grad_() for t in (q, k, v, bias))
    of = F.plain_cosine_sim_attention(qf[:, None], kf[:, None], vf[:, None], mask=mask, attn_bias=bf, attn_bias_batch_dim=True)[:, 0]
    of.backward(do.float())
    cmp("merged 3-D + mask + bias", (o, q.grad, k.grad, v.grad, bias.grad), (of, qf.grad, kf.grad, vf.grad, bf.grad))
except Exception as ex:
    print("merged 3-D + mask + bias EXC", type(ex).__name__, str(ex)[:200])

# backward twice with retain_graph, and accumulation into .grad
q, k, v = R(2, 3, 100, 64).requires_grad_(), R(2, 3, 130, 64).requires_grad_(), R(2, 3, 130, 64).requires_grad_()
do = R(2, 3, 100, 64)
o = F.flash_cosine_sim_attention(q, k, v, causal=False)
o.backward(do, retain_graph=True)
g1 = [t.grad.clone() for t in (q, k, v)]
o.backward(do)
cmp("backward twice: grad == 2 x first", [t.grad for t in (q, k, v)], [2 * x for x in g1])

# in-place modification of an input between forward and backward must be caught by autograd
q, k, v = R(1, 2, 50, 64).requires_grad_(), R(1, 2, 60, 64).requires_grad_(), R(1, 2, 60, 64).requires_grad_()
kk = k * 1.0
o = F.flash_cosine_sim_attention(q, kk, v)
kk.mul_(2.0)
try:
    o.backward(torch.ones_like(o)); print("in-place modification of k after forward: NOT caught")
except RuntimeError as ex:
    print("in-place modification of k after forward: caught:", str(ex)[:90])

# only v requires grad; only q requires grad
for who in ("q", "k", "v"):
    t = dict(q=R(1, 2, 50, 64), k=R(1, 2, 60, 64), v=R(1, 2, 60, 64))
    t[who].requires_grad_()
    o = F.flash_cosine_sim_attention(t["q"], t["k"], t["v"], causal=False)
    o.backward(torch.ones_like(o))
    print(f"only {who} requires grad:", [x.grad is not None for x in t.values()], "finite", bool(torch.isfinite(t[who].grad.float()).all()))

# functional gradcheck-style finite difference in float32 on a tiny problem (directional derivative)
torch.manual_seed(1)
q, k, v = (torch.randn(1, 2, 9, 16, device=dev, dtype=torch.float32, requires_grad=True) for _ in range(3))
bias = torch.randn(2, 9, 9, device=dev, dtype=torch.float32, requires_grad=True)
do = torch.randn(1, 2, 9, 16, device=dev)
f = lambda q, k, v, b: (F.flash_cosine_sim_attention(q, k, v, attn_bias=b, causal=True, scale=3.0) * do).sum()
f(q, k, v, bias).backward()
for nm, t in (("q", q), ("k", k), ("v", v), ("bias", bias)):
    u = torch.randn_like(t); eps = 1e-2
    args = dict(q=q.detach(), k=k.detach(), v=v.detach(), b=bias.detach())
    key = "b" if nm == "bias" else nm
    ap, am = dict(args), dict(args)
    ap[key] = args[key] + eps * u; am[key] = args[key] - eps * u
    fd = (f(ap["q"], ap["k"], ap["v"], ap["b"]) - f(am["q"], am["k"], am["v"], am["b"])).item() / (2 * eps)
    an = (t.grad * u).sum().item()
    print(f"finite difference d/d{nm}: analytic {an:+.5f}  numeric {fd:+.5f}")
