#!/usr/bin/env python3
"""Long-sequence probe: forward + backward at N = 65536 / 131072 (one head, causal, bf16), sampled query rows of o and sampled key rows of
dk / dv against float64 math on the same inputs (row-wise, so the N x N logits never exist).  (measurement / triage tool)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flash_cosine_sim_attention_amd as F
dev = "cuda"
for N, D in ((65536, 64), (131072, 64), (65536, 128)):
    g = torch.Generator(device=dev).manual_seed(N + D)
    q, k, v, do = (torch.randn(1, 1, N, D, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(4))
    for t in (q, k, v): t.requires_grad_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = F.flash_cosine_sim_attention(q, k, v, causal=True)
    o.backward(do)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fin = all(bool(torch.isfinite(t.float()).all()) for t in (o, q.grad, k.grad, v.grad))
    qh = torch.nn.functional.normalize(q.detach().double()[0, 0], dim=-1)
    kh = torch.nn.functional.normalize(k.detach().double()[0, 0], dim=-1)
    vd, dod, od = v.detach().double()[0, 0], do.double()[0, 0], o.detach().double()[0, 0]
    rows = torch.tensor([0, 1, 127, 128, 4095, N // 2 + 3, N - 129, N - 1], device=dev)
    worst_o = 0.0
    for i in rows.tolist():
        p = torch.softmax(8.0 * (kh[: i + 1] @ qh[i]), dim=0)
        worst_o = max(worst_o, ((p @ vd[: i + 1]) - od[i]).abs().max().item())
    # dv_j = sum_{i >= j} P_ij dO_i for sampled j: needs column j of P: P_ij = exp(s_ij) / l_i with l_i for all i >= j -> compute l in chunks
    l = torch.empty(N, device=dev, dtype=torch.float64)
    for a in range(0, N, 2048):
        s = 8.0 * (qh[a:a + 2048] @ kh[: a + 2048].T)
        idx = torch.arange(a, min(a + 2048, N), device=dev)[:, None]
        s = s.masked_fill(torch.arange(s.shape[1], device=dev)[None, :] > idx, float("-inf"))
        l[a:a + 2048] = torch.logsumexp(s, dim=1)
    worst_dv = 0.0
    for j in (0, 129, N // 2, N - 2):
        pcol = torch.exp(8.0 * (qh[j:] @ kh[j]) - l[j:])
        worst_dv = max(worst_dv, ((pcol @ dod[j:]) - v.grad.double()[0, 0, j]).abs().max().item())
    print(f"N {N} D {D}: fwd+bwd {dt * 1e3:.1f} ms (first call), finite {fin}, max|d| o rows {worst_o:.2e}, dv rows {worst_dv:.2e} (|dv| max {v.grad.abs().max().item():.2e})", flush=True)
    del q, k, v, do, o, l
