"""Parity of the 64-rows-per-wave, one-wave-per-SIMD forward for 16-bit D = 128 (csrc/fcsa_fwd3.hip, `use_forward_wide128`):
no bias, no key mask, static exponent shift, grids whose 256-row (causal: paired) workgroups cover the chip.

  * against the float64 oracle on (batch, head) slices -- forward elementwise and all gradients (they consume this kernel's O and inv_l):
    causal / not, N == M, M > N (causal offset), N > M (rows without a visible key come out 0), ragged rows and ragged key tails in
    every residue class that matters for a 64-key tile and a 256-row workgroup, one tile only, single-headed K/V, grouped l2norm,
    l2norm_qk=False, strided `b n (h d)` views, bf16 and f16, the inference path (no saved state);
  * against the form it replaces: the SAME call after `fcsa_debug_forward_form(0)` (the C ABI's debug knob, include/fcsa.h) runs the 32-row lean
    kernels; the two forwards must agree to the rounding of the 16-bit output (they sum rows and P~ in different orders);
  * size-independent properties at C3-D128's size: rows of P sum to one (v == 1 -> o == 1), linearity in v.
"""
import os

import numpy as np
import pytest
import torch

import tolerances as T

pytestmark = pytest.mark.gpu

DTN = {torch.float16: "f16", torch.bfloat16: "bf16"}


def _npf(t):
    return t.detach().cpu().double().numpy()


def _grid_ok(B, H, N, causal, M=0):
    """the dispatch rule of use_forward_wide128 (csrc/fcsa_fwd3.hip): 256-row (causal: paired) workgroups >= 7/8 of the device's CUs, or
    -- round 6 -- more 128-row tiles than CUs and >= 2048 keys"""
    MT, MT4 = (N + 255) // 256, (N + 127) // 128
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if B * H * ((MT + 1) // 2 if causal else MT) >= cus * 7 // 8: return True
    return B * H * ((MT4 + 1) // 2 if causal else MT4) > cus and M >= 2048


CASES = [
    # id,                      B, H,  N,    M,    dtype,          causal, groups, scale, single_kv, l2norm
    ("causal_square",          8, 28, 300,  300,  torch.bfloat16, True,   1, 8.0, False, True),
    ("causal_m_gt_n",          8, 28, 300,  450,  torch.float16,  True,   1, 8.0, False, True),
    ("causal_n_gt_m",          8, 28, 400,  130,  torch.bfloat16, True,   1, 8.0, False, True),      # rows 0..269 see no key -> 0
    ("full_ragged_tail_1",     15, 16, 520, 129,  torch.bfloat16, False,  1, 8.0, False, True),      # last tile holds ONE key
    ("full_ragged_tail_63",    15, 16, 257, 191,  torch.float16,  False,  1, 8.0, False, True),
    ("full_exact_tiles",       15, 16, 512, 256,  torch.bfloat16, False,  1, 8.0, False, True),
    ("full_one_tile",          15, 16, 256, 64,   torch.float16,  False,  1, 8.0, False, True),
    ("full_m_33",              15, 16, 300, 33,   torch.bfloat16, False,  1, 8.0, False, True),      # one key in the second block
    ("causal_one_row_block",   16, 16, 64,  64,   torch.bfloat16, True,   1, 8.0, False, True),      # waves 1 - 3 own no row
    ("causal_n_1",             16, 16, 1,   70,   torch.float16,  True,   1, 8.0, False, True),
    ("causal_five_row_tiles",  4, 20, 1100, 1100, torch.bfloat16, True,   1, 8.0, False, True),      # odd tile count: the middle tile is unpaired
    ("causal_groups8_scale1",  4, 60, 384,  384,  torch.float16,  True,   8, 1.0, False, True),      # C5's head shape
    ("causal_single_kv",       8, 28, 333,  333,  torch.bfloat16, True,   8, 1.0, True,  True),
    ("full_single_kv_ragged",  8, 28, 270,  200,  torch.float16,  False,  1, 8.0, True,  True),
    ("full_no_l2norm",         15, 16, 300, 300,  torch.bfloat16, False,  1, 0.125, False, False),   # the reference extension's contract: q, k as given
    ("causal_scale_m8",        8, 28, 260,  260,  torch.bfloat16, True,   1, -8.0, False, True),
    ("ring_wraps_twice",       15, 16, 256, 460,  torch.bfloat16, False,  1, 8.0, False, True),      # 8 tiles: every ring slot is refilled twice
    ("half_grid_causal",       4, 10, 2048, 2048, torch.bfloat16, True,   1, 8.0, False, True),      # 160 paired workgroups on 256 CUs (round 6's range)
    ("half_grid_full_ragged",  1, 36, 1000, 2100, torch.float16,  False,  1, 8.0, False, True),      # 144 workgroups, ragged rows and keys
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_wide128_matches_oracle(case):
    import flash_cosine_sim_attention_amd as F
    from oracle import cosine_sim_oracle as O
    name, B, H, N, M, dtype, causal, groups, scale, single, l2norm = case
    assert _grid_ok(B, H, N, causal, M), "shape would not dispatch to the wide D = 128 forward"
    D = 128
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + N + M)
    q = torch.randn((B, H, N, D), device="cuda", dtype=dtype, generator=g)
    kshape = (B, M, D) if single else (B, H, M, D)
    k = torch.randn(kshape, device="cuda", dtype=dtype, generator=g)
    v = torch.randn(kshape, device="cuda", dtype=dtype, generator=g)
    if not l2norm:
        q = torch.nn.functional.normalize(q.float(), dim=-1).to(dtype)
        k = torch.nn.functional.normalize(k.float(), dim=-1).to(dtype)
        scale = 8.0
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    kw = dict(scale=scale, groups=groups, causal=causal, l2norm_qk=l2norm)
    o = F.flash_cosine_sim_attention(q, k, v, **kw)
    do = torch.randn(o.shape, device="cuda", dtype=dtype, generator=g)
    o.backward(do)
    torch.cuda.synchronize()
    assert torch.isfinite(o).all()
    dt = DTN[dtype]
    atol, rtol, _ = T.FWD_TOL[dt]
    if single:
        pairs = [None]                                       # dk / dv sum over the heads: one whole batch element
    else:
        pairs = [(0, 0), (B - 1, H - 1), (B // 2, H // 3)]
    for pr in pairs:
        if pr is None:
            b = B - 1
            qs, ks, vs, dos = (_npf(t[b:b + 1]) for t in (q, k, v, do))
            gots = (_npf(o[b:b + 1]), _npf(q.grad[b:b + 1]), _npf(k.grad[b:b + 1]), _npf(v.grad[b:b + 1]))
        else:
            b, h = pr
            qs, ks, vs, dos = (_npf(t[b:b + 1, h:h + 1]) for t in (q, k, v, do))
            gots = (_npf(o[b:b + 1, h:h + 1]), _npf(q.grad[b:b + 1, h:h + 1]), _npf(k.grad[b:b + 1, h:h + 1]), _npf(v.grad[b:b + 1, h:h + 1]))
        ro, _ = O.attention_forward_stats(qs, ks, vs, **kw)
        excess = (np.abs(gots[0] - ro) - rtol * np.abs(ro)).max()
        assert T.check("wide128/forward excess", dt, excess, atol * max(np.abs(vs).max(), 1.0), name), f"{pr}: forward excess {excess:.3e}"
        if causal and N > M:                                   # rows without a visible key: exactly 0 (kernel semantics)
            dead = N - M
            assert np.abs(gots[0][..., :dead, :]).max() == 0.0
        grads = O.attention_backward(dos, qs, ks, vs, **kw)
        for gname, gg, rr in zip(("dq", "dk", "dv"), gots[1:], grads):
            rel = np.linalg.norm(gg - rr) / max(np.linalg.norm(rr), 1e-3 * np.sqrt(rr.size))
            assert T.check("wide128/grad rel-L2", dt, rel, T.GRAD_TOL[dt], name), f"{pr}: {gname} rel-L2 {rel:.3e}"


@pytest.mark.parametrize("case", [c for c in CASES if c[0] in ("causal_square", "causal_m_gt_n", "full_ragged_tail_1", "causal_five_row_tiles",
                                                               "causal_single_kv", "ring_wraps_twice", "causal_n_gt_m")], ids=lambda c: c[0])
def test_wide128_agrees_with_the_lean_form(case):
    """Same inputs through both forwards.  They differ in summation order (row sums of the un-rounded vs the rounded P~, 64-row waves),
    so o may differ by the rounding of the 16-bit output and inv_l by a few f32 ulps -- not more."""
    import flash_cosine_sim_attention_amd as F
    name, B, H, N, M, dtype, causal, groups, scale, single, l2norm = case
    D = 128
    g = torch.Generator(device="cuda").manual_seed(77 + N)
    q = torch.randn((B, H, N, D), device="cuda", dtype=dtype, generator=g)
    kshape = (B, M, D) if single else (B, H, M, D)
    k = torch.randn(kshape, device="cuda", dtype=dtype, generator=g)
    v = torch.randn(kshape, device="cuda", dtype=dtype, generator=g)
    kw = dict(scale=scale, groups=groups, causal=causal)
    from flash_cosine_sim_attention_amd import _lib
    prev = _lib.forward_form(1)
    try:
        o_new = F.flash_cosine_sim_attention(q, k, v, **kw)
        _lib.forward_form(0)
        o_old = F.flash_cosine_sim_attention(q, k, v, **kw)
        torch.cuda.synchronize()
    finally:
        _lib.forward_form(prev)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    d = (o_new.float() - o_old.float()).abs()
    bar = 1.01 * ulp * o_old.float().abs() + 4 * ulp * 2.0 ** -7      # one ulp of the value (+ a sliver near zero)
    assert (d <= bar).all(), f"max |new - old| {d.max().item():.3e}"
    frac = (d > 0).float().mean().item()
    assert frac <= 0.25, f"{frac:.3f} of the outputs differ between the two forms"


def test_wide128_strided_views_and_inference():
    """`b n (h d) -> b h n d` views (transformer.py:100) are consumed in place; under no_grad nothing is saved (q^ / inverse norms are not
    written: qn_out == nullptr) and the result is the same bits as the training-mode forward."""
    import flash_cosine_sim_attention_amd as F
    B, H, N, D = 8, 28, 300, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    qkv = torch.randn((B, N, 3, H, D), device="cuda", dtype=torch.bfloat16, generator=g)
    q, k, v = (t.transpose(1, 2) for t in qkv.unbind(2))
    assert not q.is_contiguous()
    with torch.no_grad():
        o_inf = F.flash_cosine_sim_attention(q, k, v, causal=True)
    qc, kc, vc = (t.contiguous().requires_grad_() for t in (q, k, v))
    o_train = F.flash_cosine_sim_attention(qc, kc, vc, causal=True)
    assert torch.equal(o_inf, o_train.detach())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_wide128_fullsize_properties(dtype):
    """C3 at D = 128, (4, 8, 4096, 128) causal: P rows sum to one, linearity in v, and slices against float32 PyTorch."""
    import flash_cosine_sim_attention_amd as F
    B, H, N, D = 4, 8, 4096, 128
    g = torch.Generator(device="cuda").manual_seed(11)
    q, k, v = (torch.randn((B, H, N, D), device="cuda", dtype=dtype, generator=g) for _ in range(3))
    o = F.flash_cosine_sim_attention(q, k, v, causal=True)
    ones = F.flash_cosine_sim_attention(q, k, torch.ones_like(v), causal=True)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    # sum_j P~_ij / l_i with l_i the sum of the UN-rounded P~ (fcsa_fwd3.hip): 1 up to the mean rounding of P~, far inside an output ulp
    assert (ones.float() - 1).abs().max().item() <= 2 * ulp
    v2 = torch.randn_like(v)
    o2 = F.flash_cosine_sim_attention(q, k, v2, causal=True)
    o12 = F.flash_cosine_sim_attention(q, k, (v.float() + v2.float()).to(dtype), causal=True)
    atol = T.FWD_TOL[DTN[dtype]][0]
    assert (o12.float() - (o.float() + o2.float())).abs().max().item() <= 4 * atol
    for (b, h) in ((0, 0), (B - 1, H - 1), (1, 5)):
        qs, ks, vs = q[b, h].float(), k[b, h].float(), v[b, h].float()
        s = torch.nn.functional.normalize(qs, dim=-1) @ torch.nn.functional.normalize(ks, dim=-1).t() * 8.0
        s = s.masked_fill(torch.ones_like(s, dtype=torch.bool).triu(1), float("-inf"))
        ref = torch.softmax(s, -1) @ vs
        err = ((o[b, h].float() - ref).abs() - ulp * ref.abs()).max().item()
        assert T.check("wide128/fullsize excess", DTN[dtype], err, atol), f"slice {(b, h)} excess {err:.3e}"
