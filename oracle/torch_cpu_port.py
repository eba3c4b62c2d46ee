"""Multi-threaded CPU port of the reference's pure-PyTorch attention, used ONLY as the
`cpu_baseline` leg of bench.py (and never by the product path).

TEST / MEASUREMENT INFRASTRUCTURE.  Restates plain_cosine_sim_attention
(flash_cosine_sim_attention.py:75-126) with torch CPU ops so that it runs on all host cores
(OpenMP/MKL) and differentiates through torch.autograd exactly like the reference's own
tests do (tests/test.py:90-96).  tests/test_oracle_golden.py::test_cpu_port_matches_numpy_oracle pins
it to oracle/cosine_sim_oracle.py, which is itself pinned to the reference's golden vectors.
"""
import time

import torch
import torch.nn.functional as F


def plain_attention_cpu(q, k, v, mask=None, scale=8.0, groups=1, causal=False, l2norm_qk=True):
    single = k.dim() == 3
    if l2norm_qk:
        def nrm(t):
            shp = t.shape
            return F.normalize(t.reshape(*shp[:-1], groups, shp[-1] // groups), dim=-1).reshape(shp).type(t.dtype)
        q, k = nrm(q), nrm(k)
    eq = 'bjd' if single else 'bhjd'
    sim = torch.einsum(f'bhid,{eq}->bhij', q, k) * scale
    neg = -torch.finfo(sim.dtype).max
    if causal:
        i, j = sim.shape[-2:]
        sim = sim.masked_fill(torch.ones(i, j, dtype=torch.bool).triu(j - i + 1), neg)
    if mask is not None:
        sim = sim.masked_fill(~mask[:, None, None, :], neg)
    attn = sim.softmax(dim=-1)
    return torch.einsum(f'bhij,{eq}->bhid', attn, v)


def tiled_forward_cpu(q, k, v, mask=None, scale=8.0, groups=1, causal=False, l2norm_qk=True, row_tile=512, col_tile=512):
    """Multi-threaded torch restatement of the reference's tiled CPU forward (flash_cosine_sim_attention.py:130-241): float32
    inside, blockwise (o, l) accumulation with the constant shift exp(w - scale), o /= clamp(l, 1e-12), with the INTENDED
    causal tile skip (see oracle/cosine_sim_oracle.py::tiled_attention for the reference defect).  4-D q; k, v 3-D or 4-D."""
    dtype = q.dtype
    if l2norm_qk:
        def nrm(t):
            shp = t.shape
            eps = 1e-12 if t.dtype == torch.float32 else 1e-3                                  # l2norm_cpu, py:38-42
            tg = t.reshape(*shp[:-1], groups, shp[-1] // groups)
            n = tg.norm(dim=-1, keepdim=True)
            return (tg / torch.where(n > eps, n, torch.full_like(n, eps))).reshape(shp).type(t.dtype)
        q, k = nrm(q), nrm(k)
    q, k, v = q.float(), k.float(), v.float()
    eq = 'bjd' if k.dim() == 3 else 'bhjd'
    n, m = q.shape[-2], k.shape[-2]
    diff = m - n
    o = torch.zeros_like(q)
    l = torch.zeros((*q.shape[:-1], 1))
    for r0 in range(0, n, row_tile):
        r1 = min(n, r0 + row_tile)
        for c0 in range(0, m, col_tile):
            c1 = min(m, c0 + col_tile)
            if causal and (r1 - 1 + diff) < c0:
                continue
            w = torch.einsum(f'bhid,{eq}->bhij', q[:, :, r0:r1], k[..., c0:c1, :]) * scale
            e = torch.exp(w - scale)
            if mask is not None:
                e = e.masked_fill(~mask[:, None, None, c0:c1], 0.)
            if causal and r0 + diff < c1 - 1:
                e = e.masked_fill(torch.ones(r1 - r0, c1 - c0, dtype=torch.bool).triu(r0 + diff - c0 + 1), 0.)
            o[:, :, r0:r1] += torch.einsum(f'bhij,{eq}->bhid', e, v[..., c0:c1, :])
            l[:, :, r0:r1] += e.sum(dim=-1, keepdim=True)
    return (o / l.clamp(min=1e-12)).type(dtype)


def time_forward(fn, shape_q, shape_kv, dtype, causal, reps=2, seed=0, **kw):
    """Best-of-`reps` wall time (s) of one forward of `fn` (plain_attention_cpu or tiled_forward_cpu) on the host CPU."""
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(shape_q, generator=g).to(dtype)
    k = torch.randn(shape_kv, generator=g).to(dtype)
    v = torch.randn(shape_kv, generator=g).to(dtype)
    best = float('inf')
    with torch.no_grad():
        for r in range(reps + 1):                       # first pass = warm-up
            t0 = time.perf_counter()
            fn(q, k, v, causal=causal, **kw)
            dt = time.perf_counter() - t0
            if r > 0:
                best = min(best, dt)
    return best


def time_fwd_bwd(shape_q, shape_kv, dtype, causal, scale=8.0, groups=1, reps=3, seed=0):
    """Best-of-`reps` wall time (s) of one forward+backward on the host CPU."""
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(shape_q, generator=g).to(dtype).requires_grad_()
    k = torch.randn(shape_kv, generator=g).to(dtype).requires_grad_()
    v = torch.randn(shape_kv, generator=g).to(dtype).requires_grad_()
    do = torch.randn(shape_q, generator=g).to(dtype)
    best = float('inf')
    for r in range(reps + 1):                       # first pass = warm-up
        q.grad = k.grad = v.grad = None
        t0 = time.perf_counter()
        o = plain_attention_cpu(q, k, v, scale=scale, groups=groups, causal=causal)
        o.backward(do)
        dt = time.perf_counter() - t0
        if r > 0:
            best = min(best, dt)
    return best
