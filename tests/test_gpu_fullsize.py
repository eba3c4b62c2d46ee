"""Full-size checks at BASELINE.json's configs C2-C5 where the float64 oracle is too slow.

Two kinds of evidence:
  * a plain PyTorch float32 evaluation of the same math on (batch, head) SLICES on the GPU
    (the floating-point reference the tier allows beside the oracle), and
  * size-independent identities of the operator:
      - rows of P sum to 1:      V = 1  =>  O = 1 on every row that has a valid key
      - linearity in V:          O(a v1 + b v2) = a O(v1) + b O(v2)
      - key-permutation invariance (non-causal): permuting (k, v, mask) together leaves O unchanged
      - colsum identity:         sum_j dV[j] = sum_i dO[i]            (rows of P sum to 1)
      - dS rows sum to 0:        sum_j d_bias[i, j] = 0
      - l2norm tangent space:    <dq_i, q_i> = 0 and <dk_j, k_j> = 0 per group (gradient of a
                                 scale-invariant function is orthogonal to its argument)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_slice(q, k, v, mask, causal, scale, groups, ref_dtype=torch.float32, operand_dtype=None):
    """PyTorch evaluation of softmax(scale * qn kn^T) v for [n, d] x [m, d] slices in `ref_dtype` (float32 or float64).
    operand_dtype: round the normalised operands to that dtype first -- c1 * q^ (c1 = scale * log2 e, as the kernels fold it) and
    k^ -- i.e. the "operand-faithful" reference: what exact arithmetic gives on the 16-bit operands every implementation of this op
    (the reference's too, py:57-65) feeds its S product.  It separates the error inherent to 16-bit operands, which grows with
    scale * groups, from everything else, so that check needs no range-dependent tolerance."""
    q, k, v = q.to(ref_dtype), k.to(ref_dtype), v.to(ref_dtype)
    d = q.shape[-1]

    def nrm(t):
        tg = t.reshape(t.shape[0], groups, d // groups)
        return torch.nn.functional.normalize(tg, dim=-1).reshape(t.shape)

    qn, kn = nrm(q), nrm(k)
    if operand_dtype is not None:
        c1 = scale * 1.4426950408889634
        qn = (qn * c1).to(operand_dtype).to(ref_dtype) / c1
        kn = kn.to(operand_dtype).to(ref_dtype)
    s = (qn @ kn.t()) * scale
    n, m = s.shape
    if causal:
        s = s.masked_fill(torch.ones(n, m, dtype=torch.bool, device=s.device).triu(m - n + 1), float("-inf"))
    if mask is not None:
        s = s.masked_fill(~mask[None, :], float("-inf"))
    return torch.softmax(s, dim=-1) @ v


def _grads_operand_faithful(q, k, v, do, mask, causal, scale, groups, dtype):
    """Gradients of one (batch, head) slice by the kernel's own formulas in float64 on the 16-bit OPERANDS:
    c1 * q^ and k^ rounded to `dtype` feed S, dQ^ = scale dS K^, dK^ = scale dS^T Q^ and the projection of the l2norm backward
    (what oracle.attention_backward(operand_dtype=...) computes, here in torch on the GPU so that full-size slices take
    milliseconds).  Returns (dq, dk, dv) w.r.t. the raw slices; pinned to the numpy oracle by test_operand_faithful_helper..."""
    q, k, v, do = (t.double() for t in (q, k, v, do))
    n, d = q.shape
    m = k.shape[0]

    def nrm(t):
        tg = t.reshape(t.shape[0], groups, d // groups)
        inv = 1.0 / tg.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        return (tg * inv).reshape(t.shape), inv

    qn, rq = nrm(q)
    kn, rk = nrm(k)
    c1 = abs(scale) * 1.4426950408889634
    qr = (qn * c1).to(dtype).double() / c1
    kr = kn.to(dtype).double()
    s = (qr @ kr.t()) * scale
    if causal:
        s = s.masked_fill(torch.ones(n, m, dtype=torch.bool, device=s.device).triu(m - n + 1), float("-inf"))
    if mask is not None:
        s = s.masked_fill(~mask[None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = p @ v
    delta = (do * o).sum(-1, keepdim=True)
    dv = p.t() @ do
    ds = p * (do @ v.t() - delta)
    dqh, dkh = scale * (ds @ kr), scale * (ds.t() @ qr)

    def nrm_bwd(g, xh, inv):
        gg, xg = g.reshape(g.shape[0], groups, -1), xh.reshape(g.shape[0], groups, -1)
        return (inv * (gg - xg * (gg * xg).sum(-1, keepdim=True))).reshape(g.shape)

    return nrm_bwd(dqh, qr, rq), nrm_bwd(dkh, kr, rk), dv


def _configs():
    return {
        "C2": dict(q=(4, 8, 1024, 64), kv=(4, 8, 1024, 64), dtype=torch.float16, causal=False, mask=False, scale=8, groups=1),
        "C3": dict(q=(4, 8, 4096, 64), kv=(4, 8, 4096, 64), dtype=torch.bfloat16, causal=True, mask=False, scale=8, groups=1),
        "C4": dict(q=(1, 8, 1024, 64), kv=(1, 8, 8192, 64), dtype=torch.float16, causal=False, mask=True, scale=8, groups=1),
        "C5": dict(q=(4, 8, 2048, 128), kv=(4, 2048, 128), dtype=torch.bfloat16, causal=True, mask=False, scale=1, groups=8),
        # C5 also at the default scale = 8 -> scale * groups = 64: since round 3 inside the static exponent window
        # (bf16: up to 75); "C5s10" below runs the per-row (dynamic) shift forward at full size
        "C5s8": dict(q=(4, 8, 2048, 128), kv=(4, 2048, 128), dtype=torch.bfloat16, causal=True, mask=False, scale=8, groups=8),
        "C5s10": dict(q=(4, 8, 2048, 128), kv=(4, 2048, 128), dtype=torch.bfloat16, causal=True, mask=False, scale=10, groups=8),
    }


def _make(cfg, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    q = torch.randn(cfg["q"], device="cuda", dtype=cfg["dtype"], generator=g)
    k = torch.randn(cfg["kv"], device="cuda", dtype=cfg["dtype"], generator=g)
    v = torch.randn(cfg["kv"], device="cuda", dtype=cfg["dtype"], generator=g)
    mask = None
    if cfg["mask"]:
        mask = torch.rand((cfg["q"][0], cfg["kv"][-2]), device="cuda", generator=g) > 0.25     # benchmark.py:120
    return q, k, v, mask


@pytest.mark.parametrize("name", ["C2", "C3", "C4", "C5", "C5s8", "C5s10"])
def test_forward_fullsize_vs_f32_slices_and_identities(name):
    import flash_cosine_sim_attention_amd as F
    cfg = _configs()[name]
    q, k, v, mask = _make(cfg)
    kw = dict(mask=mask, causal=cfg["causal"], scale=cfg["scale"], groups=cfg["groups"])
    o = F.flash_cosine_sim_attention(q, k, v, **kw)
    assert torch.isfinite(o).all()
    cond = max(1.0, cfg["scale"] * cfg["groups"] / 16.0)     # logit error = scale * groups * (16-bit rounding of q^, k^)
    atol = (2e-3 if cfg["dtype"] == torch.float16 else 1.5e-2) * cond
    rtol = 2.0 ** -10 if cfg["dtype"] == torch.float16 else 2.0 ** -7      # one output ulp
    single = k.dim() == 3
    atol1 = 2e-3 if cfg["dtype"] == torch.float16 else 1.5e-2      # the same bar WITHOUT the range factor, for the operand-faithful check
    for (b, h) in ((0, 0), (cfg["q"][0] - 1, cfg["q"][1] - 1), (0, 3)):
        kk, vv = (k[b], v[b]) if single else (k[b, h], v[b, h])
        mk = None if mask is None else mask[b]
        # (1) exact float64 math on the raw inputs: includes the rounding of q^, k^ to 16 bit, which scales with the logit range
        ref = _ref_slice(q[b, h], kk, vv, mk, cfg["causal"], cfg["scale"], cfg["groups"], ref_dtype=torch.float64)
        err = ((o[b, h].double() - ref).abs() - rtol * ref.abs()).max().item()
        assert err <= 1.25 * atol, f"{name} slice {(b, h)} vs exact float64: max-abs {err:.3e}"
        # (2) exact float64 math on the 16-bit operands: a fixed bar at every logit range
        ref = _ref_slice(q[b, h], kk, vv, mk, cfg["causal"], cfg["scale"], cfg["groups"], ref_dtype=torch.float64, operand_dtype=cfg["dtype"])
        err = ((o[b, h].double() - ref).abs() - rtol * ref.abs()).max().item()
        assert err <= atol1, f"{name} slice {(b, h)} vs float64 on the 16-bit operands: max-abs {err:.3e}"
    # rows of P sum to one
    ones = torch.ones_like(v)
    o1 = F.flash_cosine_sim_attention(q, k, ones, **kw)
    assert (o1.float() - 1).abs().max().item() <= (2e-3 if cfg["dtype"] == torch.float16 else 8e-3)
    # linearity in V
    v2 = torch.randn_like(v)
    o2 = F.flash_cosine_sim_attention(q, k, v2, **kw)
    o12 = F.flash_cosine_sim_attention(q, k, (v + v2), **kw)
    assert (o12.float() - (o.float() + o2.float())).abs().max().item() <= 4 * atol
    if not cfg["causal"]:
        perm = torch.randperm(k.shape[-2], device="cuda")
        kp, vp = k.index_select(-2, perm), v.index_select(-2, perm)
        mp = None if mask is None else mask.index_select(-1, perm)
        op = F.flash_cosine_sim_attention(q, kp, vp, mask=mp, causal=False, scale=cfg["scale"], groups=cfg["groups"])
        assert (op.float() - o.float()).abs().max().item() <= 2 * atol


@pytest.mark.parametrize("name", ["C3", "C5", "C4", "C5s8", "C5s10"])
def test_backward_fullsize_identities_and_slices(name):
    import flash_cosine_sim_attention_amd as F
    cfg = _configs()[name]
    q, k, v, mask = _make(cfg, seed=1)
    q.requires_grad_(); k.requires_grad_(); v.requires_grad_()
    kw = dict(mask=mask, causal=cfg["causal"], scale=cfg["scale"], groups=cfg["groups"])
    o = F.flash_cosine_sim_attention(q, k, v, **kw)
    do = torch.randn_like(o)
    o.backward(do)
    dq, dk, dv = q.grad, k.grad, v.grad
    for g in (dq, dk, dv):
        assert torch.isfinite(g).all()
    bf = cfg["dtype"] == torch.bfloat16
    # colsum identity: sum_j dV[b,h,j,:] == sum_i dO[b,h,i,:]   (summed over heads too for single-head kv)
    lhs = dv.float().sum(-2)
    rhs = do.float().sum(-2)
    if k.dim() == 3:
        rhs = rhs.sum(1)
    scale_ref = do.float().abs().sum(-2).max().item()
    assert (lhs - rhs).abs().max().item() <= (1.5e-2 if bf else 4e-3) * scale_ref
    # tangent-space identity of the fused l2norm backward
    G = cfg["groups"]
    for x, gx in ((q, dq), (k, dk)):
        xg = x.detach().float().reshape(*x.shape[:-1], G, -1)
        gg = gx.float().reshape(*x.shape[:-1], G, -1)
        dot = (xg * gg).sum(-1).abs()
        mag = (xg.norm(dim=-1) * gg.norm(dim=-1)) + 1e-20
        assert (dot / mag).max().item() <= (6e-2 if bf else 1.5e-2)
    # torch autograd float32 on (b, h) slices.  Single-headed K/V: dk / dv of a batch element are the SUM over its heads, so
    # every head of that batch element is evaluated and the per-head torch gradients are added up.
    b = cfg["q"][0] - 1
    single = k.dim() == 3
    rel = lambda a, r: ((a.float() - r).norm() / r.norm()).item()
    cond = max(1.0, cfg["scale"] * cfg["groups"] / 16.0)
    import tolerances as T
    dtn = "bf16" if bf else "f16"
    tol = T.GRAD_TOL[dtn] * cond
    dk_ref = dv_ref = None
    for h in (range(cfg["q"][1]) if single else (1,)):
        qs = q.detach()[b, h].float().requires_grad_()
        ks = (k.detach()[b] if single else k.detach()[b, h]).float().requires_grad_()
        vs = (v.detach()[b] if single else v.detach()[b, h]).float().requires_grad_()
        ref = _ref_slice(qs, ks, vs, None if mask is None else mask[b], cfg["causal"], cfg["scale"], cfg["groups"])
        (ref * do[b, h].float()).sum().backward()
        if h in (1, cfg["q"][1] - 1):
            assert T.check("fullsize/grad raw", dtn, rel(dq[b, h], qs.grad), tol, name), (h, rel(dq[b, h], qs.grad))
        dk_ref = ks.grad if dk_ref is None else dk_ref + ks.grad
        dv_ref = vs.grad if dv_ref is None else dv_ref + vs.grad
    dk_got, dv_got = (dk[b], dv[b]) if single else (dk[b, 1], dv[b, 1])
    assert T.check("fullsize/grad raw", dtn, rel(dk_got, dk_ref), tol, name), rel(dk_got, dk_ref)
    assert T.check("fullsize/grad raw", dtn, rel(dv_got, dv_ref), tol, name), rel(dv_got, dv_ref)
    # the same slices against exact float64 math on the 16-bit OPERANDS: the FIXED bars, at every scale * groups (the twin of the
    # range-scaled comparison above; round 3 had it for the forward only)
    tol1 = T.GRAD_TOL[dtn]
    dk_ref = dv_ref = None
    for h in (range(cfg["q"][1]) if single else (1,)):
        kk, vv = (k.detach()[b], v.detach()[b]) if single else (k.detach()[b, h], v.detach()[b, h])
        rdq, rdk, rdv = _grads_operand_faithful(q.detach()[b, h], kk, vv, do[b, h], None if mask is None else mask[b], cfg["causal"],
                                                cfg["scale"], cfg["groups"], cfg["dtype"])
        if h in (1, cfg["q"][1] - 1):
            assert T.check("fullsize/grad 16-bit operands", dtn, rel(dq[b, h], rdq), tol1, name), ("dq vs 16-bit operands", h, rel(dq[b, h], rdq))
        dk_ref = rdk if dk_ref is None else dk_ref + rdk
        dv_ref = rdv if dv_ref is None else dv_ref + rdv
    assert T.check("fullsize/grad 16-bit operands", dtn, rel(dk_got, dk_ref), tol1, name), ("dk vs 16-bit operands", rel(dk_got, dk_ref))
    assert T.check("fullsize/grad 16-bit operands", dtn, rel(dv_got, dv_ref), tol1, name), ("dv vs 16-bit operands", rel(dv_got, dv_ref))


def test_operand_faithful_helper_equals_the_numpy_oracle():
    """_grads_operand_faithful (torch, float64) == oracle.attention_backward(operand_dtype=...) on a small slice: the full-size twin
    checks above stand on the pinned oracle."""
    from oracle import cosine_sim_oracle as O
    g = torch.Generator(device="cuda").manual_seed(5)
    for dtype, name, causal, groups, scale in ((torch.bfloat16, "bf16", True, 4, 8.0), (torch.float16, "f16", False, 1, 16.0)):
        q, k, v, do = (torch.randn((1, 1, n_, 32), device="cuda", dtype=dtype, generator=g) for n_ in (50, 70, 70, 50))
        mask = None if causal else (torch.rand((1, 70), device="cuda", generator=g) > 0.3)
        got = _grads_operand_faithful(q[0, 0], k[0, 0], v[0, 0], do[0, 0], None if mask is None else mask[0], causal, scale, groups, dtype)
        npf = lambda t: t.detach().cpu().double().numpy()
        ref = O.attention_backward(npf(do), npf(q), npf(k), npf(v), mask=None if mask is None else mask.cpu().numpy(), scale=scale,
                                   groups=groups, causal=causal, operand_dtype=name, eps=1e-300)
        for a, r in zip(got, ref[:3]):
            assert np.abs(npf(a) - r[0, 0]).max() <= 1e-9 * max(1.0, np.abs(r).max())


def test_dbias_rows_sum_to_zero():
    import flash_cosine_sim_attention_amd as F
    torch.manual_seed(3)
    b, h, n, m, d = 2, 4, 300, 420, 64
    q = torch.randn(b, h, n, d, device="cuda", dtype=torch.float16)
    k = torch.randn(b, h, m, d, device="cuda", dtype=torch.float16)
    v = torch.randn(b, h, m, d, device="cuda", dtype=torch.float16)
    bias = (0.5 * torch.randn(h, n, m, device="cuda", dtype=torch.float16)).requires_grad_()
    o = F.flash_cosine_sim_attention(q, k, v, attn_bias=bias)
    o.backward(torch.randn_like(o))
    rowsum = bias.grad.float().sum(-1).abs().max().item()
    assert rowsum <= 2e-2 * bias.grad.float().abs().sum(-1).max().item() + 1e-3


def test_c1_shape_f32_vs_float64_oracle():
    """BASELINE config C1: (1, 8, 1024, 64) float32, non-causal, no mask -- the HIP path against the float64 numpy oracle,
    forward and all three gradients (the oracle needs a few seconds at this size)."""
    import flash_cosine_sim_attention_amd as F
    from oracle import cosine_sim_oracle as O
    g = torch.Generator(device="cuda").manual_seed(11)
    q, k, v = (torch.randn((1, 8, 1024, 64), device="cuda", dtype=torch.float32, generator=g).requires_grad_() for _ in range(3))
    do = torch.randn((1, 8, 1024, 64), device="cuda", dtype=torch.float32, generator=g)
    o = F.flash_cosine_sim_attention(q, k, v)
    o.backward(do)
    npf = lambda t: t.detach().cpu().double().numpy()
    ro = O.plain_attention(npf(q), npf(k), npf(v))
    rdq, rdk, rdv, _ = O.attention_backward(npf(do), npf(q), npf(k), npf(v))
    assert np.abs(npf(o) - ro).max() <= 1e-4                 # the reference's own f32 bound (tests/test.py:49); typical 5e-7
    rel = lambda a, r: np.linalg.norm(a - r) / np.linalg.norm(r)
    assert rel(npf(o), ro) <= 1e-5
    for name, got, ref in (("dq", q.grad, rdq), ("dk", k.grad, rdk), ("dv", v.grad, rdv)):
        assert rel(npf(got), ref) <= 2e-5, (name, rel(npf(got), ref))


def test_tensors_beyond_4_gib_address_the_last_head_correctly():
    """Maximum sizes: q, k, v, o and the gradients of 4.8 GB each (B72 H32 N8192 D128 bf16), so the last (batch, head) slices start
    beyond 4 GiB from their tensor's base -- every per-slice offset in the library has to be 64-bit.  The first and the last slice
    are compared with the same op run on contiguous copies of those slices alone (other launch forms, small offsets)."""
    import flash_cosine_sim_attention_amd as F
    B, H, N, D = 72, 32, 8192, 128
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2**30:
        pytest.skip("needs ~60 GiB of device memory")
    g = torch.Generator(device="cuda").manual_seed(41)
    q, k, v, do = (torch.randn((B, H, N, D), device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(4))
    assert q.numel() * q.element_size() > 4.4 * 2**30
    q.requires_grad_(); k.requires_grad_(); v.requires_grad_()
    o = F.flash_cosine_sim_attention(q, k, v, causal=True)
    o.backward(do)
    torch.cuda.synchronize()
    for b, h in ((0, 0), (B - 1, H - 1), (B // 2, 3)):
        sq, sk, sv = (t.detach()[b:b + 1, h:h + 1].clone().requires_grad_() for t in (q, k, v))
        so = F.flash_cosine_sim_attention(sq, sk, sv, causal=True)
        so.backward(do[b:b + 1, h:h + 1].clone())
        assert torch.isfinite(o[b, h]).all()
        assert (o[b, h].float() - so[0, 0].float()).abs().max().item() <= 2e-2
        for name, big, small in (("dq", q.grad, sq.grad), ("dk", k.grad, sk.grad), ("dv", v.grad, sv.grad)):
            rel = ((big[b, h].float() - small[0, 0].float()).norm() / small[0, 0].float().norm()).item()
            assert rel <= 1e-2, f"slice ({b}, {h}): {name} differs from the slice-only run by rel-L2 {rel:.3e}"


def test_long_sequence_65536_sampled_rows_vs_float64():
    """N = 65536, one head, causal, bf16 (512 row tiles: 256 balanced pairs; 16.8 MB per tensor): sampled query rows of o and sampled key
    rows of dv against float64 math on the same inputs, row-wise, so the N x N logits never exist (tools/long_seq_probe.py does the same
    up to N = 131072)."""
    import flash_cosine_sim_attention_amd as F
    N, D, dev = 65536, 64, "cuda"
    g = torch.Generator(device=dev).manual_seed(65)
    q, k, v, do = (torch.randn(1, 1, N, D, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(4))
    for t in (q, k, v): t.requires_grad_()
    o = F.flash_cosine_sim_attention(q, k, v, causal=True)
    o.backward(do)
    for t in (o, q.grad, k.grad, v.grad): assert torch.isfinite(t.float()).all()
    qh = torch.nn.functional.normalize(q.detach().double()[0, 0], dim=-1)
    kh = torch.nn.functional.normalize(k.detach().double()[0, 0], dim=-1)
    vd, dod, od = v.detach().double()[0, 0], do.double()[0, 0], o.detach().double()[0, 0]
    for i in (0, 1, 127, 128, 4095, N // 2 + 3, N - 129, N - 1):
        p = torch.softmax(8.0 * (kh[: i + 1] @ qh[i]), dim=0)
        assert ((p @ vd[: i + 1]) - od[i]).abs().max().item() <= 2e-2, i          # (bf16 output: measured 4e-3)
    l = torch.empty(N, device=dev, dtype=torch.float64)
    for a in range(0, N, 4096):
        s = 8.0 * (qh[a:a + 4096] @ kh[: a + 4096].T)
        idx = torch.arange(a, a + 4096, device=dev)[:, None]
        l[a:a + 4096] = torch.logsumexp(s.masked_fill(torch.arange(s.shape[1], device=dev)[None, :] > idx, float("-inf")), dim=1)
    for j in (0, 129, N // 2, N - 2):
        pcol = torch.exp(8.0 * (qh[j:] @ kh[j]) - l[j:])
        assert ((pcol @ dod[j:]) - v.grad.double()[0, 0, j]).abs().max().item() <= 5e-2, j      # (measured 1.1e-2 against |dv| up to 4)
