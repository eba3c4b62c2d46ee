"""Split-key forward (fcsa_fwd.hip: fwd_kernel with gridDim.y = splits + fwd_combine_kernel; chosen by fcsa_capi.hip
forward_splits when a problem's 128-row tiles (causal: pairs of them) cannot fill the chip and the caller passes the optional
workspace): parity with the float64 oracle, forward and -- through the saved inv_l -- backward."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import cosine_sim_oracle as O

pytestmark = pytest.mark.gpu

DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}
import tolerances as T
FWD_TOL = {d: t[:2] for d, t in T.FWD_TOL.items()}      # (atol, rtol) -- tests/tolerances.py
GRAD_TOL = T.GRAD_TOL

CASES = [
    # dtype, B, H, N,   M,    D,  mask,  single_kv, l2norm, groups
    ("bf16", 1, 2, 40, 2500, 64, False, False, True, 1),      # 4 splits, ragged last split
    ("f16", 1, 8, 300, 2100, 64, True, False, True, 1),       # key mask, 3 row tiles per head
    ("bf16", 1, 4, 200, 2048, 64, False, False, True, 4),     # grouped l2norm (bf16: static shift up to scale * groups = 60)
    ("f32", 1, 1, 8, 4096, 128, False, False, True, 1),       # 8 splits, f32 MFMA path
    ("bf16", 2, 3, 129, 2300, 32, True, True, True, 1),       # single-headed K/V, 4 splits
    ("f16", 1, 4, 64, 2100, 96, False, False, False, 1),      # reference contract: q, k already normalised
    ("bf16", 1, 1, 1, 8192, 16, False, False, True, 1),       # one query row ("decode"), 13 splits
]


def _npf(t):
    return t.detach().cpu().double().numpy()


@pytest.mark.parametrize("dtype,B,H,N,M,D,use_mask,single_kv,l2norm,groups", CASES)
def test_split_forward_matches_oracle(dtype, B, H, N, M, D, use_mask, single_kv, l2norm, groups):
    _split_forward_case(dtype, B, H, N, M, D, use_mask, single_kv, l2norm, groups, False)


# Causal problems whose PAIRS of 128-row tiles cannot fill the chip (round 6): the key range of each row tile -- up to its diagonal -- is
# split, so a workgroup of split y sees a different key window in each of its two passes and some windows are empty.  The same holds for
# the backward of these problems: dQ splits each row tile's keys, dK/dV each key tile's queries (from its diagonal down).
CAUSAL_CASES = [
    # dtype, B, H, N,    M,    D,  single_kv, l2norm, groups
    ("bf16", 1, 2, 2200, 2200, 64, False, True, 1),       # 9 pairs; ragged rows and keys
    ("f16", 1, 3, 1500, 2600, 32, False, True, 1),        # M > N: every row sees the first 1100 keys as well
    ("bf16", 2, 2, 3000, 2100, 128, False, True, 8),      # N > M: rows 0 .. 899 see no key at all (zeros), 256-byte rows
    ("f16", 1, 4, 2300, 2300, 64, True, True, 1),         # single-headed K/V
    ("bf16", 1, 1, 4096, 4096, 64, False, False, 1),      # q, k already normalised (the reference extension's contract); 16 pairs
]


@pytest.mark.parametrize("dtype,B,H,N,M,D,single_kv,l2norm,groups", CAUSAL_CASES)
def test_causal_split_forward_matches_oracle(dtype, B, H, N, M, D, single_kv, l2norm, groups):
    _split_forward_case(dtype, B, H, N, M, D, False, single_kv, l2norm, groups, True)


@pytest.mark.parametrize("dtype,B,H,N,M,D,use_mask,l2norm,groups,causal", [
    ("bf16", 1, 4, 3000, 200, 64, False, True, 1, False),       # 8 key tiles, 5 query splits: 20 slabs per gradient summed to one K/V head
    ("f16", 2, 2, 2500, 100, 128, True, True, 8, False),        # key mask, 256-byte rows
    ("bf16", 1, 8, 2048, 2048, 128, False, True, 8, True),      # C5's shape at batch 1: 64 pairs, causal
])
def test_split_query_dkv_single_headed_kv(dtype, B, H, N, M, D, use_mask, l2norm, groups, causal):
    """Single-headed K/V with a split dK/dV launch (round 6): the per-head f32 slabs become heads x splits slabs, one finalize launch sums them."""
    _split_forward_case(dtype, B, H, N, M, D, use_mask, True, l2norm, groups, causal, fwd_split=False)


def _split_forward_case(dtype, B, H, N, M, D, use_mask, single_kv, l2norm, groups, causal, fwd_split=True):
    import flash_cosine_sim_attention_amd as F
    from flash_cosine_sim_attention_amd import _lib
    dt = DT[dtype]
    prob = _lib.problem(dt, (B, H, 1 if single_kv else H, N, M, D), causal, False, l2norm, groups, 8.0 if l2norm else 0.125)
    assert not fwd_split or _lib.load().fcsa_forward_workspace_bytes(C.byref(prob)) > 0, "case would not take the split-key path"
    if single_kv and not fwd_split:   # single-headed K/V: heads slabs per gradient without a split, heads x splits with one
        assert _lib.load().fcsa_backward_workspace_bytes(C.byref(prob)) > (B * H * N * 4 + 255) // 256 * 256 + 2 * ((B * H * M * D * 4 + 255) // 256 * 256), \
            "case would not split the backward"
    if causal and not single_kv:      # the causal cases also run the split-key dQ / split-query dK/dV kernels (f32 slabs behind delta in the workspace)
        assert _lib.load().fcsa_backward_workspace_bytes(C.byref(prob)) > (B * H * N * 4 + 255) // 256 * 256, "case would not split the backward"
    g = torch.Generator(device="cuda").manual_seed(N * 7 + M)
    q = torch.randn((B, H, N, D), device="cuda", dtype=dt, generator=g)
    kv_shape = (B, M, D) if single_kv else (B, H, M, D)
    k = torch.randn(kv_shape, device="cuda", dtype=dt, generator=g)
    v = torch.randn(kv_shape, device="cuda", dtype=dt, generator=g)
    if not l2norm:
        q, k = torch.nn.functional.normalize(q.float(), dim=-1).to(dt), torch.nn.functional.normalize(k.float(), dim=-1).to(dt)
    mask = None
    if use_mask:
        mask = torch.rand((B, M), device="cuda", generator=g) > 0.4
        mask[:, :3] = True
        mask[:, 600:1100] = False            # a whole split (or most of it) without a valid key
    q.requires_grad_(); k.requires_grad_(); v.requires_grad_()
    scale = 8.0 if l2norm else 0.125
    o = F.flash_cosine_sim_attention(q, k, v, mask=mask, scale=scale, groups=groups, l2norm_qk=l2norm, causal=causal)
    do = torch.randn(o.shape, device="cuda", dtype=dt, generator=g)
    o.backward(do)
    torch.cuda.synchronize()
    mk = None if mask is None else _npf(mask).astype(bool)
    kw = dict(mask=mk, scale=scale, groups=groups, l2norm_qk=l2norm, causal=causal)
    ro, _ = O.attention_forward_stats(_npf(q), _npf(k), _npf(v), **kw)
    atol, rtol = FWD_TOL[dtype]
    excess = (np.abs(_npf(o) - ro) - rtol * np.abs(ro)).max()
    assert T.check("split/forward excess", dtype, excess, atol * max(np.abs(_npf(v)).max(), 1.0)), f"forward excess {excess:.3e}"
    rdq, rdk, rdv, _ = O.attention_backward(_npf(do), _npf(q), _npf(k), _npf(v), **kw)
    for name, got, ref in (("dq", q.grad, rdq), ("dk", k.grad, rdk), ("dv", v.grad, rdv)):
        rel = np.linalg.norm(_npf(got) - ref) / max(np.linalg.norm(ref), 1e-3 * np.sqrt(ref.size))
        assert T.check("split/grad rel-L2", dtype, rel, GRAD_TOL[dtype] * T.SPLIT_GRAD_FACTOR[dtype]), f"{name} rel-L2 {rel:.3e}"


def test_split_and_unsplit_agree_through_the_c_abi():
    """Same problem with and without the optional workspace: the C ABI promises the same result."""
    from flash_cosine_sim_attention_amd import _lib
    lib = _lib.load()
    B, H, N, M, D = 1, 4, 96, 3000, 64
    g = torch.Generator(device="cuda").manual_seed(11)
    q, k, v = (torch.randn(s, device="cuda", dtype=torch.bfloat16, generator=g) for s in ((B, H, N, D), (B, H, M, D), (B, H, M, D)))
    outs = []
    for use_ws in (False, True):
        o = torch.empty_like(q)
        inv_l = torch.empty((B, H, N), device="cuda", dtype=torch.float32)
        qn, kn = torch.empty_like(q), torch.empty_like(k)
        prob = _lib.problem(q.dtype, (B, H, H, N, M, D), False, False, True, 1, 8.0)
        nbytes = int(lib.fcsa_forward_workspace_bytes(C.byref(prob)))
        assert nbytes > 0
        ws = torch.empty((nbytes,), device="cuda", dtype=torch.uint8) if use_ws else None
        args = _lib.ForwardArgs(prob, _lib.tensor4(q), _lib.tensor4(k), _lib.tensor4(v), _lib.tensor4(o), inv_l.data_ptr(),
                                None, None, _lib.NormState(qn.data_ptr(), kn.data_ptr(), None, None),
                                None if ws is None else ws.data_ptr(), 0 if ws is None else nbytes,
                                torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.fcsa_forward(C.byref(args)), "fcsa_forward")
        torch.cuda.synchronize()
        outs.append((o.float(), inv_l.clone()))
    (o0, l0), (o1, l1) = outs
    assert (o0 - o1).abs().max().item() <= 2.0 ** -7 * o0.abs().max().item()      # one output ulp: the summation order differs
    assert ((l0 - l1).abs() / l0.abs()).max().item() <= 1e-5


# Split-query dK/dV (fcsa_capi.hip backward_dkv_splits: few keys, many queries, K/V with heads; the causal form: CAUSAL_CASES above): the dK/dV kernel runs
# gridDim.y query ranges that write partial f32 slabs, one finalize launch sums them (and applies the l2norm backward to dK^).
# Through the public op, i.e. with the workspace the compiled binding sizes from fcsa_backward_workspace_bytes.
SPLIT_QUERY_CASES = [
    # dtype, B, H, N,    M,   D,  mask,  l2norm, groups, non-contiguous k / v
    ("bf16", 1, 2, 3000, 150, 64, False, True, 1, False),      # 5 splits; N is not a multiple of the query tile
    ("f16", 2, 2, 1100, 100, 64, True, True, 2, False),        # 2 splits, key mask with a masked tail, 2 l2norm groups
    ("f32", 1, 2, 1536, 80, 32, False, True, 1, False),        # 3 splits, f32 MFMA path
    ("bf16", 1, 3, 2048, 300, 128, False, True, 8, False),     # 256-byte rows (64-row query tiles), 9 key tiles
    ("f16", 1, 2, 1024, 96, 96, False, False, 1, False),       # reference contract: q, k already normalised
    ("bf16", 1, 2, 2048, 128, 64, False, True, 1, True),       # k, v are views of a [B, M, H, D] tensor (dk, dv come back contiguous)
]


@pytest.mark.parametrize("dtype,B,H,N,M,D,use_mask,l2norm,groups,strided", SPLIT_QUERY_CASES)
def test_split_query_dkv_matches_oracle(dtype, B, H, N, M, D, use_mask, l2norm, groups, strided):
    import flash_cosine_sim_attention_amd as F
    from flash_cosine_sim_attention_amd import _lib
    dt = DT[dtype]
    scale = 8.0 / groups if l2norm else 0.125
    al = lambda x: (x + 255) // 256 * 256
    prob = _lib.problem(dt, (B, H, H, N, M, D), False, False, l2norm, groups, scale)
    assert _lib.load().fcsa_backward_workspace_bytes(C.byref(prob)) >= al(B * H * N * 4) + 2 * al(2 * B * H * M * D * 4), \
        "case would not take the split-query path"
    g = torch.Generator(device="cuda").manual_seed(N + 13 * M)
    q = torch.randn((B, H, N, D), device="cuda", dtype=dt, generator=g)
    if strided:
        k = torch.randn((B, M, H, D), device="cuda", dtype=dt, generator=g).transpose(1, 2)
        v = torch.randn((B, M, H, D), device="cuda", dtype=dt, generator=g).transpose(1, 2)
    else:
        k = torch.randn((B, H, M, D), device="cuda", dtype=dt, generator=g)
        v = torch.randn((B, H, M, D), device="cuda", dtype=dt, generator=g)
    if not l2norm:
        q, k = torch.nn.functional.normalize(q.float(), dim=-1).to(dt), torch.nn.functional.normalize(k.float(), dim=-1).to(dt)
    mask = None
    if use_mask:
        mask = torch.rand((B, M), device="cuda", generator=g) > 0.3
        mask[:, 0] = True
        mask[:, M - 20:] = False
    q.requires_grad_(); k.requires_grad_(); v.requires_grad_()
    o = F.flash_cosine_sim_attention(q, k, v, mask=mask, scale=scale, groups=groups, l2norm_qk=l2norm)
    do = torch.randn(o.shape, device="cuda", dtype=dt, generator=g)
    o.backward(do)
    torch.cuda.synchronize()
    mk = None if mask is None else _npf(mask).astype(bool)
    kw = dict(mask=mk, scale=scale, groups=groups, l2norm_qk=l2norm)
    rdq, rdk, rdv, _ = O.attention_backward(_npf(do), _npf(q), _npf(k), _npf(v), **kw)
    for name, got, ref in (("dq", q.grad, rdq), ("dk", k.grad, rdk), ("dv", v.grad, rdv)):
        assert torch.isfinite(got).all(), name
        rel = np.linalg.norm(_npf(got) - ref) / max(np.linalg.norm(ref), 1e-3 * np.sqrt(ref.size))
        assert T.check("split/grad rel-L2", dtype, rel, GRAD_TOL[dtype] * T.SPLIT_GRAD_FACTOR[dtype]), f"{name} rel-L2 {rel:.3e}"
