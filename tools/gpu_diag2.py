#!/usr/bin/env python3
"""Targeted f32 dq diagnostics: exact bad positions, determinism, variants."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from oracle import cosine_sim_oracle as O
npf = lambda t: t.detach().cpu().double().numpy()

def run(n, m, d, causal, rep=2, dtype=torch.float32, seed=0):
    torch.manual_seed(seed)
    q = torch.randn(1, 1, n, d, device="cuda", dtype=dtype, requires_grad=True)
    k = torch.randn(1, 1, m, d, device="cuda", dtype=dtype, requires_grad=True)
    v = torch.randn(1, 1, m, d, device="cuda", dtype=dtype, requires_grad=True)
    do = torch.randn(1, 1, n, d, device="cuda", dtype=dtype)
    kw = dict(causal=causal, l2norm_qk=False, scale=0.125)
    rdq, rdk, rdv, _ = O.attention_backward(npf(do), npf(q), npf(k), npf(v), **kw)
    for r in range(rep):
        q.grad = k.grad = v.grad = None
        o = F.flash_cosine_sim_attention(q, k, v, **kw)
        o.backward(do)
        torch.cuda.synchronize()
        e = np.abs(npf(q.grad) - rdq)[0, 0]
        bad = np.argwhere(e > 1e-4)
        print(f"N{n} M{m} D{d} causal={causal} run{r}: dq max {e.max():.3e} nbad {len(bad)}  dk {np.abs(npf(k.grad)-rdk).max():.2e} dv {np.abs(npf(v.grad)-rdv).max():.2e}")
        if len(bad):
            rows = sorted(set(bad[:, 0].tolist()))
            print("   bad rows:", rows[:40])
            for rr in rows[:3]:
                cols = bad[bad[:, 0] == rr][:, 1].tolist()
                print(f"   row {rr}: cols {cols[:64]}")
                print(f"      got {npf(q.grad)[0,0,rr,cols[:6]]}  ref {rdq[0,0,rr,cols[:6]]}")

for (n, m, d, c) in ((32, 32, 64, False), (32, 32, 64, True), (32, 64, 64, False), (64, 32, 64, False), (32, 40, 64, False),
                     (128, 100, 64, False), (63, 63, 64, False), (63, 63, 128, False), (63, 63, 32, False), (32, 32, 96, False)):
    run(n, m, d, c)
