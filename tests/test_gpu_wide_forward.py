"""Parity of the WIDE forward kernel (64 query rows per wave, rotating MFMA / softmax pipeline; fcsa_fwd.hip fwd2_kernel).

launch_forward picks it when the 256-row workgroups cover the chip (>= 224 of them), dim_head is 32 and
  * not causal with >= 4096 keys, or
  * causal with N >= 8192
(rounds 1 - 5 also sent dim_head 64 and non-causal problems of any length there; round 6 re-measured: the 32-row kernel at two waves
per SIMD is 13 - 16 % faster at dim_head 64 since its round-4 changes -- fcsa_fwd.hip use_wide_fwd.  dim_head 96 went to the lean
two-wave 32-row kernel in round 3.  The other cases below stay as parity cases of whatever kernel the dispatch gives them).
The shapes below are chosen to land on it through the normal dispatch:
  * many small heads (batch*heads = 224+), checked against the float64 oracle elementwise with the stated tolerances,
  * the long causal shapes, checked on (batch, head) slices against a float32 PyTorch evaluation on the GPU
    (the oracle is O(N^2) float64 numpy: too slow at N = 8192) and through the rows-sum-to-one identity.
Gradients flow through the (unchanged) backward kernels and are compared as well, so a wrong inv_l or O would show.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# forward tolerances stated in DESIGN.md §5: |o - ref| <= atol * max|v| + rtol * |ref|
import tolerances as T
DTN = {torch.float16: "f16", torch.bfloat16: "bf16"}
TOL = {dt: T.FWD_TOL[n][:2] for dt, n in DTN.items()}          # (atol, rtol) -- tests/tolerances.py
GRAD_REL = {dt: T.GRAD_TOL[n] for dt, n in DTN.items()}


def _npf(t):
    return t.detach().cpu().double().numpy()


def _inputs(B, H, N, M, D, dtype, seed, kv_heads=None):
    g = torch.Generator(device="cuda").manual_seed(seed)
    Hk = H if kv_heads is None else kv_heads
    q = torch.randn((B, H, N, D), device="cuda", dtype=dtype, generator=g)
    k = torch.randn((B, Hk, M, D), device="cuda", dtype=dtype, generator=g)
    v = torch.randn((B, Hk, M, D), device="cuda", dtype=dtype, generator=g)
    return q, k, v


WIDE_SMALL = [
    # B, H, N,   M,   D,  dtype,           mask,  groups, scale
    (7, 32, 300, 300, 64, torch.bfloat16, False, 1, 8),
    (7, 32, 300, 333, 64, torch.float16, True, 1, 8),       # ragged N and M, key padding mask
    (8, 28, 257, 190, 32, torch.bfloat16, False, 2, 10),    # N just past one 256-row tile, grouped l2norm
    (4, 60, 256, 64, 96, torch.float16, False, 1, 8),       # exactly one key tile
    (4, 60, 320, 129, 96, torch.bfloat16, True, 3, 6),      # tail key tile of 1 key
    (14, 16, 512, 512, 64, torch.float16, False, 1, 16),    # larger scale
    (5, 48, 270, 70, 32, torch.float16, True, 1, 8),
    (8, 28, 300, 4170, 32, torch.float16, False, 1, 8),     # the wide kernel's non-causal regime since round 6: >= 4096 keys, ragged tail
    (8, 28, 257, 4096, 32, torch.bfloat16, False, 4, 2),    # ... grouped l2norm, N just past one 256-row tile
]


@pytest.mark.parametrize("B,H,N,M,D,dtype,use_mask,groups,scale", WIDE_SMALL)
def test_wide_forward_matches_oracle(B, H, N, M, D, dtype, use_mask, groups, scale):
    assert B * H * ((N + 255) // 256) >= 224, "shape would not dispatch to the wide (D <= 64) / lean (D = 96) kernel"
    _many_heads_vs_oracle(B, H, N, M, D, dtype, use_mask, groups, scale, causal=False)


# Round 3: 16-bit D = 96 / 128 on grids that cover the chip with 8-wave workgroups run the LEAN forms -- forward (no cross-block
# prefetch), dQ (4-wave workgroups, two per CU) and dK/dV (V fragments from an LDS copy of the workgroup's keys), all at two waves
# per SIMD.  Same check as above: forward on five (batch, head) pairs and all gradients of one pair against the float64 oracle.
LEAN_FORMS = [
    # B, H, N,   M,   D,   dtype,          mask,  groups, scale, causal
    (8, 28, 300, 300, 128, torch.bfloat16, False, 1, 8, True),
    (8, 28, 300, 333, 128, torch.float16, True, 1, 8, False),     # ragged M, key padding mask, one batch element without keys
    (7, 32, 260, 300, 96, torch.bfloat16, False, 3, 6, True),      # M > N: causal offset; grouped l2norm (group size 32)
    (4, 60, 384, 384, 128, torch.float16, False, 8, 1, True),      # C5's head shape: groups = 8, scale 1
    (15, 16, 520, 520, 128, torch.bfloat16, False, 1, 8, False),   # 240 heads x 3 row tiles, non-causal
    (4, 60, 256, 129, 96, torch.float16, True, 1, 8, False),       # tail key tile of 1 key
]


@pytest.mark.parametrize("B,H,N,M,D,dtype,use_mask,groups,scale,causal", LEAN_FORMS)
def test_lean_two_wave_forms_match_oracle(B, H, N, M, D, dtype, use_mask, groups, scale, causal):
    MT = (N + 255) // 256
    assert B * H * ((MT + 1) // 2 if causal else MT) >= 224, "shape would not dispatch to the 8-wave (lean) kernels"
    _many_heads_vs_oracle(B, H, N, M, D, dtype, use_mask, groups, scale, causal=causal)


# Causal with many small heads: the narrow kernels in their 8-waves-per-workgroup form (forward, dQ, dK/dV), which
# the launchers choose when 256-row / 256-key workgroups still cover the chip.
EIGHT_WAVE_CAUSAL = [
    (8, 28, 300, 300, 64, torch.bfloat16, 1, 8),
    (8, 28, 260, 300, 64, torch.float16, 1, 8),       # M > N: causal offset
    (16, 16, 200, 200, 32, torch.bfloat16, 2, 10),
    (6, 40, 257, 257, 16, torch.float16, 1, 8),
]


@pytest.mark.parametrize("B,H,N,M,D,dtype,groups,scale", EIGHT_WAVE_CAUSAL)
def test_eight_wave_causal_matches_oracle(B, H, N, M, D, dtype, groups, scale):
    _many_heads_vs_oracle(B, H, N, M, D, dtype, False, groups, scale, causal=True)


def _many_heads_vs_oracle(B, H, N, M, D, dtype, use_mask, groups, scale, causal):
    import flash_cosine_sim_attention_amd as F
    from oracle import cosine_sim_oracle as O
    q, k, v = _inputs(B, H, N, M, D, dtype, seed=B * 1000 + N)
    mask = None
    if use_mask:
        g = torch.Generator(device="cuda").manual_seed(7)
        mask = torch.rand((B, M), device="cuda", generator=g) > 0.3
        mask[:, 0] = True
        mask[1, :] = False                     # one batch element without any valid key: rows must come out 0
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    o = F.flash_cosine_sim_attention(q, k, v, mask=mask, scale=scale, groups=groups, causal=causal)
    do = torch.randn(o.shape, device="cuda", dtype=dtype, generator=torch.Generator(device="cuda").manual_seed(3))
    o.backward(do)
    torch.cuda.synchronize()
    # oracle on a subset of the (batch, head) pairs: every batch element that has a special role + a spread of heads
    atol, rtol = TOL[dtype]
    pairs = [(0, 0), (1, 1), (B - 1, H - 1), (B // 2, H // 3), (2 % B, 5 % H)]
    for (b, h) in pairs:
        mk = None if mask is None else _npf(mask[b:b + 1]).astype(bool)
        ro, inv_l = O.attention_forward_stats(_npf(q[b:b + 1, h:h + 1]), _npf(k[b:b + 1, h:h + 1]), _npf(v[b:b + 1, h:h + 1]),
                                              mask=mk, scale=scale, groups=groups, l2norm_qk=True, causal=causal)
        got = _npf(o[b:b + 1, h:h + 1])
        vmax = np.abs(_npf(v[b, h])).max()
        err = np.abs(got - ro) - rtol * np.abs(ro)
        assert T.check("wide/forward excess", DTN[dtype], err.max(), atol * vmax), f"(b,h)={(b, h)} max excess {err.max():.3e}"
        assert np.isfinite(got).all()
    if mask is not None:
        assert o[1].abs().max().item() == 0.0                 # no valid key -> zeros (oracle: attention_forward_stats)
    # gradients against the oracle's backward on one pair (they consume the wide kernel's O and inv_l)
    b, h = pairs[2]
    mk = None if mask is None else _npf(mask[b:b + 1]).astype(bool)
    gq, gk, gv, _ = O.attention_backward(_npf(do[b:b + 1, h:h + 1]), _npf(q[b:b + 1, h:h + 1]), _npf(k[b:b + 1, h:h + 1]),
                                         _npf(v[b:b + 1, h:h + 1]), mask=mk, scale=scale, groups=groups, l2norm_qk=True, causal=causal)
    for name, got, ref in (("dq", q.grad[b:b + 1, h:h + 1], gq), ("dk", k.grad[b:b + 1, h:h + 1], gk), ("dv", v.grad[b:b + 1, h:h + 1], gv)):
        rel = np.linalg.norm(_npf(got) - ref) / max(np.linalg.norm(ref), 1e-3 * np.sqrt(ref.size))
        assert T.check("wide/grad rel-L2", DTN[dtype], rel, GRAD_REL[dtype]), f"{name} rel-L2 {rel:.3e}"


def _ref_slice(q, k, v, causal, scale):
    q, k, v = q.float(), k.float(), v.float()
    s = (torch.nn.functional.normalize(q, dim=-1) @ torch.nn.functional.normalize(k, dim=-1).t()) * scale
    n, m = s.shape
    if causal:
        s = s.masked_fill(torch.ones(n, m, dtype=torch.bool, device=s.device).triu(m - n + 1), float("-inf"))
    return torch.softmax(s, dim=-1) @ v


WIDE_CAUSAL = [
    # B, H, N,    M,    D,  dtype
    (2, 8, 8192, 8192, 64, torch.bfloat16),
    (2, 7, 8192, 8200, 32, torch.float16),      # M > N: causal offset (seq_len_diff) and a ragged key tail
    (7, 8, 2048, 2048, 96, torch.bfloat16),
    (8, 8, 2050, 2050, 96, torch.float16),      # ragged rows
]


@pytest.mark.parametrize("B,H,N,M,D,dtype", WIDE_CAUSAL)
def test_wide_forward_causal_vs_f32_slices(B, H, N, M, D, dtype):
    import flash_cosine_sim_attention_amd as F
    MT = (N + 255) // 256
    assert B * H * ((MT + 1) // 2) >= 224, "shape would not dispatch to the wide (D <= 64) / lean (D = 96) kernel"
    q, k, v = _inputs(B, H, N, M, D, dtype, seed=N + D)
    o = F.flash_cosine_sim_attention(q, k, v, causal=True)
    assert torch.isfinite(o).all()
    atol, rtol = TOL[dtype]
    for (b, h) in ((0, 0), (B - 1, H - 1), (B // 2, 3)):
        ref = _ref_slice(q[b, h], k[b, h], v[b, h], True, 8)
        err = ((o[b, h].float() - ref).abs() - rtol * ref.abs()).max().item()
        assert T.check("wide/causal slice excess", DTN[dtype], err, atol * v[b, h].float().abs().max().item()), f"slice {(b, h)} excess {err:.3e}"
    # rows of P sum to one (every row has at least one valid key when M >= N)
    o1 = F.flash_cosine_sim_attention(q, k, torch.ones_like(v), causal=True)
    assert (o1.float() - 1).abs().max().item() <= (2e-3 if dtype == torch.float16 else 8e-3)


@pytest.mark.parametrize("dtype,D,groups", [(torch.bfloat16, 128, 8), (torch.float16, 96, 1)])
def test_lean_forms_with_single_headed_kv(dtype, D, groups):
    """Single-headed K/V (head stride 0) through the lean kernels of a chip-covering grid: the forward must equal, bit for bit, the
    same call with K / V expanded to every head, and dk / dv must equal the head sums of that call's gradients (the f32 slabs + finalize
    against 60 separately rounded 16-bit gradients: compared at the 16-bit rounding of the summands)."""
    import flash_cosine_sim_attention_amd as F
    B, H, N = 4, 60, 300
    g = torch.Generator(device="cuda").manual_seed(11)
    q = torch.randn((B, H, N, D), device="cuda", dtype=dtype, generator=g).requires_grad_()
    k1 = torch.randn((B, N, D), device="cuda", dtype=dtype, generator=g).requires_grad_()
    v1 = torch.randn((B, N, D), device="cuda", dtype=dtype, generator=g).requires_grad_()
    do = torch.randn((B, H, N, D), device="cuda", dtype=dtype, generator=g)
    kw = dict(causal=True, groups=groups, scale=1.0 if groups > 1 else 8.0)
    o1 = F.flash_cosine_sim_attention(q, k1, v1, **kw)
    o1.backward(do)
    dq1, dk1, dv1 = q.grad.clone(), k1.grad.clone(), v1.grad.clone()
    q.grad = None
    kf = k1.detach()[:, None].expand(B, H, N, D).contiguous().requires_grad_()
    vf = v1.detach()[:, None].expand(B, H, N, D).contiguous().requires_grad_()
    of = F.flash_cosine_sim_attention(q, kf, vf, **kw)
    of.backward(do)
    assert torch.equal(o1, of)
    assert torch.equal(dq1, q.grad)
    for name, single, full in (("dk", dk1, kf.grad), ("dv", dv1, vf.grad)):
        ref = full.float().sum(1)
        scale_ = full.float().abs().sum(1).max().item()          # rounding of the 60 summands, each to 2^-9 (bf16) / 2^-11 (f16) relative
        err = (single.float() - ref).abs().max().item()
        assert err <= (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10) * scale_, f"{name}: {err:.3e} vs scale {scale_:.3e}"
