"""The forward kernel's per-row exponent reference (fcsa_fwd.hip online_recentre) under inputs BUILT to move it: the logits of a row
ramp along the key axis by more than the re-centring threshold per tile (upwards: every tile rescales what the row has accumulated;
downwards: the first block holds the max and the tail underflows gracefully), a float16 bias ramps across the whole key range, and
the many-head shapes reach the 8-wave and the lean (D = 128) forms through the normal dispatch.  Random data (test_gpu_fuzz.py,
test_gpu_cabi_direct.py at scale * groups = 120 / 128) only moves the reference in the first tiles of a row.

Oracle: the float64 restatement with the row-sum clamp off (rows are normalised exactly in this regime, like the reference's PyTorch
plain_cosine_sim_attention; see test_gpu_fuzz.py).  Tolerances: test_gpu_parity.py's, times max(1, scale * groups / 16) for the
16-bit types (the rounded q^, k^ move a logit by ~scale * 2^-9 / 2^-12).
"""
import numpy as np
import pytest
import torch

from oracle import cosine_sim_oracle as O

pytestmark = pytest.mark.gpu

DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}
import tolerances as T
FWD_TOL = {d: t[:2] for d, t in T.FWD_TOL.items()}      # (atol, rtol) -- tests/tolerances.py
GRAD_TOL = T.GRAD_TOL


def _ramp_inputs(dtype, B, H, N, M, D, order, seed):
    """q rows near one direction u, key j at cosine c_j to it, c ramping over [-0.9, 0.9] along (or against) the key axis"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    u = torch.nn.functional.normalize(torch.randn((B, H, 1, D), device="cuda", generator=g), dim=-1)
    q = u + 0.05 * torch.randn((B, H, N, D), device="cuda", generator=g)
    c = torch.linspace(-0.9, 0.9, M, device="cuda")
    if order == "down":
        c = c.flip(0)
    w = torch.randn((B, H, M, D), device="cuda", generator=g)
    w = torch.nn.functional.normalize(w - (w * u).sum(-1, keepdim=True) * u, dim=-1)
    k = c[None, None, :, None] * u + (1 - c * c).sqrt()[None, None, :, None] * w
    k = k * (0.5 + torch.rand((B, H, M, 1), device="cuda", generator=g))          # the l2norm removes it again
    v = torch.randn((B, H, M, D), device="cuda", generator=g)
    do = torch.randn((B, H, N, D), device="cuda", generator=g)
    dt = DT[dtype]
    return q.to(dt), k.to(dt), v.to(dt), do.to(dt)


CASES = [
    # dtype, B, H, N, M, D, scale, causal, order, bias ramp (0 = none)
    ("f16", 1, 2, 100, 512, 64, 64.0, False, "up", 0.0),          # +23 log2 units per 64-key tile against a threshold of 10
    ("f16", 1, 2, 100, 512, 64, 64.0, False, "down", 0.0),
    ("f16", 2, 2, 300, 300, 64, 48.0, True, "up", 0.0),           # causal: every row ends at its own max
    ("f16", 1, 2, 70, 640, 32, 1.0, False, "up", 40.0),           # the bias alone spans e^80: constant shifts cannot hold it in f16
    ("f16", 1, 2, 70, 640, 128, 1.0, False, "down", 40.0),
    ("bf16", 1, 2, 100, 512, 64, 128.0, False, "up", 0.0),        # bf16 threshold 64: +41 log2 units per tile, moves every other tile
    ("bf16", 7, 32, 256, 256, 128, 128.0, True, "up", 0.0),       # lean 8-wave form (224 heads)
    ("f16", 7, 32, 256, 256, 64, 64.0, True, "up", 0.0),          # 8-wave prefetching form
    ("f16", 7, 32, 256, 320, 128, 64.0, False, "up", 0.0),        # lean form, float16 threshold
    ("f32", 1, 2, 90, 400, 32, 120.0, False, "up", 0.0),
    ("f32", 1, 1, 64, 256, 128, 100.0, True, "down", 0.0),
    # large POSITIVE biases (round 4 advice): bf16 / f32 keep the constant shift up to scale * groups = 40, where include/fcsa.h documents
    # +45 of exponent headroom for a bias; beyond 40 a bias sends the problem to the per-row form, which has no limit
    ("bf16", 1, 2, 100, 512, 64, 40.0, False, "up", 30.0),        # static window, bias up to +30 on top of the largest logits
    ("bf16", 2, 2, 300, 300, 64, 8.0, True, "down", 44.0),        # static, the default scale, bias up to +44
    ("bf16", 1, 2, 100, 512, 64, 48.0, False, "up", 60.0),        # bound 48 with a bias: per-row reference
    ("f32", 1, 2, 90, 400, 32, 40.0, False, "up", 30.0),
]


def _npf(t):
    return t.detach().cpu().double().numpy()


@pytest.mark.parametrize("dtype,B,H,N,M,D,scale,causal,order,bias_ramp", CASES,
                         ids=[f"{c[0]}-d{c[5]}-{c[8]}-{'bias' if c[9] else 'plain'}-{i}" for i, c in enumerate(CASES)])
def test_moving_reference_matches_oracle(dtype, B, H, N, M, D, scale, causal, order, bias_ramp):
    import flash_cosine_sim_attention_amd as F
    q, k, v, do = _ramp_inputs(dtype, B, H, N, M, D, order, seed=N * M + D)
    bias = None
    if bias_ramp:
        r = torch.linspace(-bias_ramp, bias_ramp, M, device="cuda")
        r = r.flip(0) if order == "down" else r
        bias = (r[None, None, :] + 0.25 * torch.randn((H, N, M), device="cuda")).to(DT[dtype]).requires_grad_()
    q.requires_grad_(); k.requires_grad_(); v.requires_grad_()
    o = F.flash_cosine_sim_attention(q, k, v, attn_bias=bias, scale=scale, causal=causal)
    o.backward(do)
    torch.cuda.synchronize()
    assert torch.isfinite(o).all() and all(torch.isfinite(t.grad).all() for t in (q, k, v))

    pairs = [(0, 0), (B - 1, H - 1)] if B * H > 8 else [None]
    # f32: a logit of magnitude scale * log2(e) carries ~2^-24 of it as absolute error into the exponent (1e-5 at scale 120), and
    # in these inputs every row sits at that magnitude: dq measured 2.4e-5 against the 2e-5 stated for scale 8
    cond = max(1.0, scale / 16.0) if dtype != "f32" else max(1.0, scale / 64.0)
    atol, rtol = FWD_TOL[dtype]
    for pr in pairs:
        sl = (slice(None), slice(None)) if pr is None else (slice(pr[0], pr[0] + 1), slice(pr[1], pr[1] + 1))
        bs = None if bias is None else _npf(bias)
        okw = dict(attn_bias=bs, scale=scale, causal=causal, eps=1e-300)
        ro, _ = O.attention_forward_stats(_npf(q)[sl], _npf(k)[sl], _npf(v)[sl], **okw)
        excess = (np.abs(_npf(o)[sl] - ro) - rtol * np.abs(ro)).max()
        assert T.check("online/forward excess", dtype, excess, cond * atol * max(1.0, np.abs(_npf(v)[sl]).max())), f"{pr}: forward excess {excess:.3e}"
        grads = O.attention_backward(_npf(do)[sl], _npf(q)[sl], _npf(k)[sl], _npf(v)[sl], **okw)
        gots = [_npf(q.grad)[sl], _npf(k.grad)[sl], _npf(v.grad)[sl]] + ([_npf(bias.grad)] if bias is not None else [])
        for name, gg, rr in zip(["dq", "dk", "dv", "d_bias"], gots, grads):
            rel = np.linalg.norm(gg - rr) / max(np.linalg.norm(rr), 1e-3 * np.sqrt(rr.size))
            lim = cond * GRAD_TOL[dtype] * (1.5 if name == "d_bias" else 1.0)
            assert T.check("online/" + ("d_bias" if name == "d_bias" else "grad") + " rel-L2", dtype, rel, lim), f"{pr}: {name} rel-L2 {rel:.3e} > {lim}"
