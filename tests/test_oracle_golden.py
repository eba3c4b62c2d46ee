"""Pins oracle/cosine_sim_oracle.py against golden vectors produced by the REFERENCE
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import cases as C
from oracle import cosine_sim_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _np(t):
    return None if t is None else t.double().numpy() if t.is_floating_point() else t.numpy()


def _load(case):
    inp = {k: _np(v) for k, v in C.make_inputs(case).items()}
    gold = dict(np.load(os.path.join(GOLD, case["name"] + ".npz")))
    return inp, gold


def _valid_rows(case, inp):
    """[B,1,N] (or [B,N] merged) bool: rows with at least one valid key."""
    n, m, b = case["n"], case["m"], case["b"]
    ok = np.ones((b, n), dtype=bool)
    if case["causal"]:
        ok &= (np.arange(n)[None, :] + (m - n)) >= 0
    if inp["mask"] is not None:
        ok &= inp["mask"].any(-1)[:, None]
    return ok if case["merged"] else ok[:, None, :]


@pytest.mark.parametrize("case", C.CASES, ids=lambda c: c["name"])
def test_plain_forward_matches_reference(case):
    inp, gold = _load(case)
    o = O.plain_attention(inp["q"], inp["k"], inp["v"], mask=inp["mask"], attn_bias=inp["attn_bias"], **C.op_kwargs(case))
    assert o.shape == gold["o_plain"].shape
    assert np.abs(o - gold["o_plain"]).max() <= 2e-6


@pytest.mark.parametrize("case", C.CASES, ids=lambda c: c["name"])
def test_kernel_form_forward_matches_reference_on_nonempty_rows(case):
    """exp(S - scale) / rowsum form (cu:1216-1246) == softmax form wherever a row has a valid key;
    rows without one are 0 (kernel semantics)."""
    inp, gold = _load(case)
    o, inv_l = O.attention_forward_stats(inp["q"], inp["k"], inp["v"], mask=inp["mask"], attn_bias=inp["attn_bias"],
                                         **C.op_kwargs(case))
    ok = _valid_rows(case, inp)
    okb = np.broadcast_to(ok[..., None], o.shape)
    assert np.abs(np.where(okb, o - gold["o_plain"], 0.0)).max() <= 2e-6
    assert np.abs(np.where(okb, 0.0, o)).max() == 0.0
    assert np.isfinite(inv_l).all()


@pytest.mark.parametrize("case", [c for c in C.CASES if c["tiled_ok"]], ids=lambda c: c["name"])
def test_tiled_forward_matches_reference_cpu_path(case):
    inp, gold = _load(case)
    kw = C.op_kwargs(case)
    q, k = inp["q"], inp["k"]
    if kw["l2norm_qk"]:
        q, k = O.l2norm(q, kw["groups"]), O.l2norm(k, kw["groups"])
    for tile in (512, 64):
        o = O.tiled_attention(q, k, inp["v"], mask=inp["mask"], attn_bias=inp["attn_bias"], scale=kw["scale"],
                              causal=kw["causal"], attn_bias_batch_dim=kw["attn_bias_batch_dim"],
                              row_tile=tile, col_tile=tile)
        assert np.abs(o - gold["o_tiled"]).max() <= 2e-5     # reference computes this path in f32


@pytest.mark.parametrize("case", C.CASES, ids=lambda c: c["name"])
def test_backward_matches_reference_autograd(case):
    inp, gold = _load(case)
    dq, dk, dv, db = O.attention_backward(inp["do"], inp["q"], inp["k"], inp["v"], mask=inp["mask"],
                                          attn_bias=inp["attn_bias"], **C.op_kwargs(case))
    ok = _valid_rows(case, inp)
    empty = not ok.all()
    okq = np.broadcast_to(ok[..., None], dq.shape)
    tol = 5e-6
    assert np.abs(np.where(okq, dq - gold["dq"], 0.0)).max() <= tol * max(1.0, np.abs(gold["dq"]).max())
    if not empty:
        # rows with no valid key: plain's softmax is uniform there and feeds dv / d_bias; the kernel form gives 0
        assert np.abs(dv - gold["dv"]).max() <= tol * max(1.0, np.abs(gold["dv"]).max())
        assert np.abs(dk - gold["dk"]).max() <= tol * max(1.0, np.abs(gold["dk"]).max())
        if db is not None:
            assert np.abs(db - gold["db"]).max() <= tol * max(1.0, np.abs(gold["db"]).max())
    else:
        assert np.abs(dk - gold["dk"]).max() <= tol * max(1.0, np.abs(gold["dk"]).max())


@pytest.mark.parametrize("case", [c for c in C.CASES if c["name"][:3] in ("g01", "g12", "g16", "g17")],
                         ids=lambda c: c["name"])
def test_l2norm_matches_reference(case):
    inp, gold = _load(case)
    assert np.abs(O.l2norm(inp["q"], case["groups"]) - gold["qn"]).max() <= 1e-6
    assert np.abs(O.l2norm(inp["k"], case["groups"]) - gold["kn"]).max() <= 1e-6


def test_l2norm_backward_finite_difference():
    rs = np.random.RandomState(0)
    x = rs.standard_normal((3, 5, 12))
    g = rs.standard_normal((3, 5, 12))
    for groups in (1, 3):
        ana = O.l2norm_backward(g, x, groups)
        num = np.zeros_like(x)
        eps = 1e-6
        it = np.nditer(x, flags=["multi_index"])
        for _ in it:
            i = it.multi_index
            xp, xm = x.copy(), x.copy()
            xp[i] += eps
            xm[i] -= eps
            num[i] = ((O.l2norm(xp, groups) - O.l2norm(xm, groups)) * g).sum() / (2 * eps)
        assert np.abs(ana - num).max() < 1e-6


def test_flop_convention():
    # C3 = (4,8,4096,4096,64) causal fwd+bwd = 240.58 GFLOP; dense 481.04
    assert abs(O.algorithmic_flops(4, 8, 4096, 4096, 64) / 1e9 - 481.04) < 0.01
    assert abs(O.algorithmic_flops(4, 8, 4096, 4096, 64, causal=True) / 1e9 - 240.58) < 0.01
    assert O.causal_valid_count(4, 6) == 3 + 4 + 5 + 6
    assert O.causal_valid_count(3, 2) == 0 + 1 + 2


def test_cpu_port_matches_numpy_oracle():
    """oracle/torch_cpu_port.py (the cpu_baseline leg of bench.py) == the numpy oracle, forward and grads."""
    import torch
    from oracle import torch_cpu_port as P
    torch.manual_seed(0)
    for single, causal, groups in ((False, True, 1), (True, False, 2)):
        q = torch.randn(2, 3, 50, 32, dtype=torch.float64, requires_grad=True)
        kv = (2, 70, 32) if single else (2, 3, 70, 32)
        k = torch.randn(kv, dtype=torch.float64, requires_grad=True)
        v = torch.randn(kv, dtype=torch.float64, requires_grad=True)
        do = torch.randn(2, 3, 50, 32, dtype=torch.float64)
        o = P.plain_attention_cpu(q, k, v, causal=causal, groups=groups)
        o.backward(do)
        n = lambda t: t.detach().numpy()
        ro = O.plain_attention(n(q), n(k), n(v), causal=causal, groups=groups)
        rdq, rdk, rdv, _ = O.attention_backward(n(do), n(q), n(k), n(v), causal=causal, groups=groups)
        assert np.abs(n(o) - ro).max() < 1e-10
        for a, r in ((q.grad, rdq), (k.grad, rdk), (v.grad, rdv)):
            assert np.abs(n(a) - r).max() < 1e-9


def test_tiled_cpu_port_matches_numpy_oracle():
    """oracle/torch_cpu_port.tiled_forward_cpu (a cpu_baseline leg of bench.py) == the numpy oracle's tiled path, incl. causal
    N > tile (where the reference's own skip is wrong) and single-headed K/V with a key mask."""
    import torch
    from oracle import torch_cpu_port as P
    torch.manual_seed(0)
    q = torch.randn(2, 3, 150, 32)
    k, v = torch.randn(2, 3, 150, 32), torch.randn(2, 3, 150, 32)
    o = P.tiled_forward_cpu(q, k, v, causal=True, row_tile=64, col_tile=32)
    ref = O.plain_attention(q.double().numpy(), k.double().numpy(), v.double().numpy(), causal=True)
    assert np.abs(o.double().numpy() - ref).max() < 2e-5
    k1, v1 = torch.randn(2, 90, 32), torch.randn(2, 90, 32)
    mask = torch.rand(2, 90) > 0.3
    o = P.tiled_forward_cpu(q, k1, v1, mask=mask, groups=2, scale=4.0, row_tile=64, col_tile=32)
    ref = O.plain_attention(q.double().numpy(), k1.double().numpy(), v1.double().numpy(), mask=mask.numpy(), groups=2, scale=4.0)
    assert np.abs(o.double().numpy() - ref).max() < 2e-5


def test_backward_o_saved_is_the_backward_input():
    """`o` is an input of the backward (reference backward_preprocess, cu:1256-1335: delta = rowsum(dO * o)).  With the exact output the
    option changes nothing; with a perturbed one the gradients move exactly as the analytic dependence on delta says: dS changes by
    -P * d(delta), so dv (which does not depend on delta) stays, and dq / dk change by the corresponding products."""
    rng = np.random.RandomState(7)
    b, h, n, m, d = 2, 2, 5, 9, 16
    q, k, v, do = (rng.randn(b, h, x, d) for x in (n, m, m, n))
    kw = dict(scale=8, groups=2, causal=True)
    o, _ = O.attention_forward_stats(q, k, v, **kw)
    base = O.attention_backward(do, q, k, v, **kw)
    same = O.attention_backward(do, q, k, v, o_saved=o, **kw)
    for a, c in zip(base[:3], same[:3]):
        np.testing.assert_allclose(a, c, rtol=0, atol=1e-12)
    o2 = o + 1e-3 * rng.randn(*o.shape)
    moved = O.attention_backward(do, q, k, v, o_saved=o2, **kw)
    np.testing.assert_allclose(moved[2], base[2], rtol=0, atol=1e-12)          # dv: no delta in it
    assert np.abs(moved[0] - base[0]).max() > 1e-6                            # dq, dk follow delta
    assert np.abs(moved[1] - base[1]).max() > 1e-6
    # first-order check through the normalised operands: l2norm_qk = False makes dq^ = scale * dS K^ directly
    qh, kh = O.l2norm(q, 1), O.l2norm(k, 1)
    kw2 = dict(scale=8, groups=1, causal=True, l2norm_qk=False)
    o3, st = O.attention_forward_stats(qh, kh, v, **kw2)
    g0 = O.attention_backward(do, qh, kh, v, **kw2)
    g1 = O.attention_backward(do, qh, kh, v, o_saved=o3 + 1e-3, **kw2)
    ddelta = (do * 1e-3).sum(-1)                                               # delta moves by rowsum(dO) * 1e-3
    # P from the forward statistics: dq^ changes by -scale * (P * ddelta) K^ = -scale * ddelta * (P K^)
    s = np.einsum("bhid,bhjd->bhij", qh, kh) * 8
    valid = np.tril(np.ones((n, m), dtype=bool), k=m - n)
    p = np.where(valid, np.exp(s - 8), 0.0)
    p = p / np.maximum(p.sum(-1, keepdims=True), 1e-10)
    expect = -8 * ddelta[..., None] * np.einsum("bhij,bhjd->bhid", p, kh)
    np.testing.assert_allclose(g1[0] - g0[0], expect, rtol=1e-9, atol=1e-12)
