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
