"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Grid = the reference's own (tests/test.py:31-37) at reduced batch/heads, plus bf16, D=16, N != M,
causal with N != M, single-head-KV gradients, groups, merged batch-heads, l2norm_qk=False.

Stated tolerances.  The oracle is evaluated in float64 on the dtype-rounded inputs; the reference
itself asserts max-abs 1e-4 (f32) / 1e-1 (f16) against its PyTorch path (tests/test.py:12-18,49).
    forward o, elementwise  |got - ref| <= atol + rtol * |ref|   and   rel-L2 = ||got-ref|| / ||ref||
        f32 : atol 1e-5,   rtol 2e-5            rel-L2 <= 4e-6
        f16 : atol 2.5e-3, rtol 2^-10 (1 ulp)   rel-L2 <= 8.5e-4
        bf16: atol 2e-2, rtol 2^-7  (1 ulp)     rel-L2 <= 4.5e-3
      (atol = 2 * eps_P * max|v|: rows with 1-3 keys are a convex combination of values up to |v| ~ 4.5
       whose weights carry the 16-bit rounding of P, eps_P = 2^-9 bf16 / 2^-11 f16 plus the rounded q^, k^)
    gradients (dq, dk, dv; d_bias x1.5)          rel-L2 <= 2e-5 f32 / 1.4e-3 f16 / 8.5e-3 bf16   (tests/tolerances.py: calibration)
bf16 carries 8 significant bits: rounding the OUTPUT alone is 1.1e-3 rel-L2, and q^, k^, P are rounded
to the 16-bit type before each MFMA exactly like the reference rounds them to its input dtype.

Wide logit ranges (scale * groups > 16; golden cases g28-g31, `wide`): the rounding of q^, k^ is amplified by the logit range in
ANY 16-bit evaluation, the reference's included, so the comparison against exact math on the raw inputs scales its bars with
cases.logit_cond() -- and every 16-bit case is compared a second time against exact math on the 16-bit OPERANDS the S product is
fed (oracle `operand_dtype`, "operand-faithful") with the FIXED bars above, forward and gradients, at every range.
"""
import numpy as np
import pytest
import torch

import cases as C
import tolerances as T
from oracle import cosine_sim_oracle as O

pytestmark = pytest.mark.gpu

DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}
FWD_TOL, GRAD_TOL = T.FWD_TOL, T.GRAD_TOL      # (atol, rtol, rel-L2) forward; rel-L2 gradients -- tests/tolerances.py


def _close(got, ref, dtype, cond=1.0, label="parity/fwd-excess"):
    atol, rtol, _ = FWD_TOL[dtype]
    return T.check(label, dtype, float((np.abs(got - ref) - rtol * np.abs(ref)).max()), atol * cond)


def _supported(dtype):
    import flash_cosine_sim_attention_amd._lib as L
    if dtype != "f32":
        return True
    buf = __import__("ctypes").create_string_buffer(512)
    L.load().fcsa_debug(buf, 512)
    return b"f32" in buf.value


def _np(t):
    if t is None:
        return None
    t = t.detach().cpu()
    return t.double().numpy() if t.is_floating_point() else t.numpy()


def _rel(a, b):
    # the floor keeps the ratio meaningful when the exact result is (nearly) zero, e.g. dq for N = M = 1
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-3 * np.sqrt(b.size)))


def _valid_rows(n, m, b, causal, mask, merged):
    ok = np.ones((b, n), dtype=bool)
    if causal:
        ok &= (np.arange(n)[None, :] + (m - n)) >= 0
    if mask is not None:
        ok &= mask.any(-1)[:, None]
    return ok if merged else ok[:, None, :]


def _run_case(case, check_grads=True):
    import flash_cosine_sim_attention_amd as F
    dtype = case["dtype"]
    if not _supported(dtype):
        pytest.skip("float32 kernels not built")
    inp = C.make_inputs(case, device="cuda")
    kw = C.op_kwargs(case)
    q, k, v = (inp[n].clone().requires_grad_(check_grads) for n in ("q", "k", "v"))
    bias = inp["attn_bias"].clone().requires_grad_(check_grads) if inp["attn_bias"] is not None else None
    o = F.flash_cosine_sim_attention(q, k, v, mask=inp["mask"], attn_bias=bias, **kw)
    assert o.shape == q.shape and o.dtype == q.dtype
    npi = {n: _np(t) for n, t in inp.items()}
    # (per-row-reference regime: rows are normalised exactly, like the reference's plain_cosine_sim_attention; the kernel form's
    #  1e-10 clamp, taken in exp(S - scale) units, does not exist there)
    dyn = C.dynamic_shift_regime(dtype, case["scale"], case["groups"], case["l2norm"], case["bias"])
    okw = dict(mask=npi["mask"], attn_bias=npi["attn_bias"], eps=1e-300 if dyn else 1e-10, **kw)
    cond = C.logit_cond(dtype, case["scale"], case["groups"], case["l2norm"])
    got = _np(o)
    assert np.isfinite(got).all()
    if check_grads:
        o.backward(inp["do"])
    # pass 0: exact math on the raw inputs (bars x cond); pass 1 (16-bit types): exact math on the 16-bit operands (fixed bars)
    for operand_dtype in ((None,) if dtype == "f32" else (None, dtype)):
        c = cond if operand_dtype is None else 1.0
        what = "raw inputs" if operand_dtype is None else "16-bit operands"
        ref_o, _ = O.attention_forward_stats(npi["q"], npi["k"], npi["v"], operand_dtype=operand_dtype, **okw)
        assert _close(got, ref_o, dtype, c), f"fwd vs {what}: max-abs {np.abs(got - ref_o).max():.3e}"
        assert T.check("parity/fwd-rel/" + what, dtype, _rel(got, ref_o), FWD_TOL[dtype][2] * c, case.get("name")), f"fwd vs {what}: rel-L2 {_rel(got, ref_o):.3e}"
        if not check_grads:
            continue
        rdq, rdk, rdv, rdb = O.attention_backward(npi["do"], npi["q"], npi["k"], npi["v"], operand_dtype=operand_dtype, **okw)
        gt = GRAD_TOL[dtype] * c
        for name, got_t, ref in (("dq", q.grad, rdq), ("dk", k.grad, rdk), ("dv", v.grad, rdv)):
            g = _np(got_t)
            assert g.shape == ref.shape, name
            assert np.isfinite(g).all(), name
            assert T.check("parity/grad/" + what, dtype, _rel(g, ref), gt, case.get("name")), f"{name} vs {what}: rel-L2 {_rel(g, ref):.3e}"
        if bias is not None:
            g = _np(bias.grad)
            assert np.isfinite(g).all()
            assert T.check("parity/dbias/" + what, dtype, _rel(g, rdb), gt * 1.5, case.get("name")), f"db vs {what}: rel-L2 {_rel(g, rdb):.3e}"


# ------------------------------------------------------------------------------------------------
# 1. golden-vector cases (the same inputs the reference produced tests/golden/*.npz from)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", C.CASES, ids=lambda c: c["name"])
def test_golden_case_vs_oracle(case):
    _run_case(case)


@pytest.mark.parametrize("case", C.CASES, ids=lambda c: c["name"])
def test_golden_case_vs_reference_fixture(case):
    """HIP forward / grads against the REFERENCE's stored outputs (rows that have a valid key)."""
    import os
    import flash_cosine_sim_attention_amd as F
    gold = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", case["name"] + ".npz")))
    inp = C.make_inputs(case, device="cuda")
    kw = C.op_kwargs(case)
    q, k, v = (inp[n].clone().requires_grad_() for n in ("q", "k", "v"))
    bias = inp["attn_bias"].clone().requires_grad_() if inp["attn_bias"] is not None else None
    o = F.flash_cosine_sim_attention(q, k, v, mask=inp["mask"], attn_bias=bias, **kw)
    o.backward(inp["do"])
    ok = _valid_rows(case["n"], case["m"], case["b"], case["causal"], _np(inp["mask"]), case["merged"])
    okb = np.broadcast_to(ok[..., None], gold["o_plain"].shape)
    cond = C.logit_cond(case["dtype"], case["scale"], case["groups"], case["l2norm"])      # (1 for every case but the `wide` ones)
    assert _close(np.where(okb, _np(o), 0.0), np.where(okb, gold["o_plain"], 0.0), case["dtype"], cond, "fixture/fwd-excess")
    if ok.all():
        gt = GRAD_TOL[case["dtype"]] * cond
        for nm, g_ in (("dq", q.grad), ("dk", k.grad), ("dv", v.grad)):
            assert T.check("fixture/grad", case["dtype"], _rel(_np(g_), gold[nm]), gt, case["name"]), nm
        if bias is not None:
            assert T.check("fixture/dbias", case["dtype"], _rel(_np(bias.grad), gold["db"]), gt * 1.5, case["name"])


# ------------------------------------------------------------------------------------------------
# 2. the reference's test grid (tests/test.py:31-37), forward and gradients
# ------------------------------------------------------------------------------------------------
def _grid():
    out = []
    seed = 100
    for causal, mask in ((True, False), (False, True), (False, False)):
        for bias in (False, True):
            for n in (63, 127):
                for d in (32, 64, 96, 128):
                    for dtype in ("f16", "bf16", "f32"):
                        for bias_batch in (False, True):
                            for single in (False, True):
                                if bias_batch and not bias:
                                    continue          # attn_bias_batch_dim only matters with a bias
                                seed += 1
                                name = f"c{int(causal)}m{int(mask)}b{int(bias)}bb{int(bias_batch)}s{int(single)}_n{n}_d{d}_{dtype}"
                                out.append(C._c(name, b=2, h=3, n=n, d=d, dtype=dtype, causal=causal, mask=mask, bias=bias,
                                                bias_batch=bias_batch, single_kv=single, seed=seed))
    return out


@pytest.mark.parametrize("case", _grid(), ids=lambda c: c["name"])
def test_reference_grid(case):
    _run_case(case)


# ------------------------------------------------------------------------------------------------
# 3. gaps: tile boundaries, cross attention, groups, head dims, strided views, error behaviour
# ------------------------------------------------------------------------------------------------
GAPS = [
    C._c("x_n1_m1_d32", n=1, m=1, d=32, dtype="f16", seed=301),
    C._c("x_n64_m64_d64", n=64, d=64, dtype="bf16", seed=302),
    C._c("x_n65_m129_d64_causal", n=65, m=129, d=64, dtype="bf16", causal=True, seed=303),
    C._c("x_n129_m65_d64_causal", n=129, m=65, d=64, dtype="f16", causal=True, seed=304),
    C._c("x_n300_m300_d128_causal", h=1, n=300, d=128, dtype="bf16", causal=True, seed=305),
    C._c("x_n257_m513_d96_mask", b=2, h=1, n=257, m=513, d=96, dtype="f16", mask=True, seed=306),
    C._c("x_n200_d16_causal", n=200, d=16, dtype="bf16", causal=True, seed=307),
    C._c("x_n200_d16_mask_single", b=2, n=200, d=16, dtype="f16", mask=True, single_kv=True, seed=308),
    C._c("x_groups4_d64", n=100, d=64, dtype="bf16", groups=4, scale=2, seed=309),
    C._c("x_groups8_d128_causal_single", b=2, h=4, n=130, d=128, dtype="bf16", groups=8, scale=1, causal=True,
         single_kv=True, seed=310),
    C._c("x_groups16_d32_generic_path", n=70, d=32, dtype="f16", groups=16, scale=1, seed=311),   # group size 2
    C._c("x_groups3_d96", n=70, d=96, dtype="bf16", groups=3, scale=2, seed=312),
    C._c("x_groups8_d96_generic_path", n=70, d=96, dtype="f16", groups=8, scale=1, seed=313),      # group size 12
    C._c("x_nol2norm_d64", n=100, d=64, dtype="f16", l2norm=False, scale=0.125, seed=314),
    C._c("x_nol2norm_single_d64", b=2, n=100, d=64, dtype="bf16", l2norm=False, scale=0.125, single_kv=True, seed=315),
    C._c("x_merged_d64_causal", b=5, n=100, d=64, dtype="bf16", merged=True, causal=True, seed=316),
    C._c("x_merged_bias_d32", b=3, n=70, d=32, dtype="f16", merged=True, bias=True, seed=317),
    C._c("x_bias_cross_d64", n=70, m=150, d=64, dtype="f16", bias=True, seed=318),
    C._c("x_prefixmask_d64", b=3, n=100, m=260, d=64, dtype="bf16", mask=True, mask_kind="prefix", seed=319),
    C._c("x_n1000_d64_causal_f16", h=1, n=1000, d=64, dtype="f16", causal=True, seed=320),
    C._c("x_scale1_d64", n=100, d=64, dtype="bf16", scale=1, seed=321),
    C._c("x_scale16_d64", n=100, d=64, dtype="f16", scale=16, seed=322),
    # bias rows that keep 8 / 16-byte alignment (M % 8 == 0): block loads in forward / dQ, LDS-transposed blocks in dKV; M = 264
    # and 132 mix full key tiles (block form) with a partial last one (element form) in one launch; M = 512 on a small grid runs
    # the key-split d_bias owners
    C._c("x_bias_vec_m256_d64", b=2, h=3, n=96, m=256, d=64, dtype="f16", bias=True, seed=323),
    C._c("x_bias_vec_m264_d64_bb", b=3, h=2, n=100, m=264, d=64, dtype="bf16", bias=True, bias_batch=True, seed=324),
    C._c("x_bias_vec_m132_d32_f32", b=2, h=2, n=70, m=132, d=32, dtype="f32", bias=True, seed=325),
    C._c("x_bias_vec_causal_n256_d64", b=1, h=2, n=256, d=64, dtype="bf16", causal=True, bias=True, seed=326),
    C._c("x_bias_vec_m512_d128_splits", b=2, h=2, n=130, m=512, d=128, dtype="f16", bias=True, seed=327),
    C._c("x_bias_vec_mask_m384_d96", b=2, h=2, n=65, m=384, d=96, dtype="bf16", bias=True, mask=True, seed=328),
]


@pytest.mark.parametrize("case", GAPS, ids=lambda c: c["name"])
def test_gap_cases(case):
    _run_case(case)


def test_empty_rows_are_zero_and_finite():
    """Rows without any valid key: kernel semantics are 0, gradients finite."""
    import flash_cosine_sim_attention_amd as F
    for name in ("g15_causal_n100_m63_d32_f32", "g20_mask_fullrow_d32_f32"):
        case = dict(C.BY_NAME[name], dtype="bf16")
        inp = C.make_inputs(case, device="cuda")
        q, k, v = (inp[n].clone().requires_grad_() for n in ("q", "k", "v"))
        o = F.flash_cosine_sim_attention(q, k, v, mask=inp["mask"], **C.op_kwargs(case))
        o.backward(inp["do"])
        ok = _valid_rows(case["n"], case["m"], case["b"], case["causal"], _np(inp["mask"]), False)
        got = _np(o)
        assert np.abs(np.where(np.broadcast_to(ok[..., None], got.shape), 0.0, got)).max() == 0.0
        for g in (q.grad, k.grad, v.grad):
            assert torch.isfinite(g).all()


def test_strided_views_are_consumed_in_place():
    """`b n (h d) -> b h n d` views (transformer.py:100) must give the same result as contiguous copies."""
    import flash_cosine_sim_attention_amd as F
    torch.manual_seed(0)
    b, h, n, d = 2, 4, 130, 64
    x = torch.randn(b, n, 3 * h * d, device="cuda", dtype=torch.bfloat16)
    q, k, v = (t.reshape(b, n, h, d).permute(0, 2, 1, 3) for t in x.chunk(3, dim=-1))
    assert not q.is_contiguous()
    o1 = F.flash_cosine_sim_attention(q, k, v, causal=True)
    o2 = F.flash_cosine_sim_attention(q.contiguous(), k.contiguous(), v.contiguous(), causal=True)
    assert torch.equal(o1, o2)


def test_l1_extension_surface_matches_reference_contract():
    """ext.forward / ext.backward (the pybind surface, cu:1630-1639, cu:1752-1764) with pre-normalised q, k."""
    from flash_cosine_sim_attention_amd import ext, l2norm_tensors
    torch.manual_seed(1)
    b, h, n, d = 2, 3, 100, 64
    q, k, v = (torch.randn(b, h, n, d, device="cuda", dtype=torch.float16) for _ in range(3))
    qn, kn = l2norm_tensors(q, k)
    qn.requires_grad_()
    o, inv_l, should = ext.forward(qn, kn, v, None, None, False, 8.0, True)
    assert should and inv_l.shape == (b, h, n) and inv_l.dtype == torch.float32
    do = torch.randn_like(o)
    dq, dk, dv, db = ext.backward(do, o, inv_l, qn.detach(), kn, v, None, None, False, 8.0, True)
    assert db is None
    rdq, rdk, rdv, _ = O.attention_backward(_np(do), _np(qn), _np(kn), _np(v), scale=8.0, causal=True, l2norm_qk=False)
    ro, rinv = O.attention_forward_stats(_np(qn), _np(kn), _np(v), scale=8.0, causal=True, l2norm_qk=False)
    assert _close(_np(o), ro, "f16")
    assert _rel(_np(inv_l), rinv) <= 1e-3
    for g, r in ((dq, rdq), (dk, rdk), (dv, rdv)):
        assert _rel(_np(g), r) <= 3e-3
    o2, l2, should2 = ext.forward(qn.detach(), kn, v, None, None, False, 8.0, True)
    assert not should2 and l2.numel() == 0 and torch.equal(o2, o)
    assert "gfx950" in ext.debug()


def test_error_behaviour():
    import flash_cosine_sim_attention_amd as F
    q = torch.randn(1, 2, 8, 64, device="cuda", dtype=torch.float16)
    with pytest.raises(ValueError):
        F.flash_cosine_sim_attention(q, q, q, mask=torch.ones(1, 8, dtype=torch.bool, device="cuda"), causal=True)
    with pytest.raises(ValueError):
        bad = torch.randn(1, 2, 8, 48, device="cuda", dtype=torch.float16)
        F.flash_cosine_sim_attention(bad, bad, bad)
    with pytest.raises(TypeError):
        F.flash_cosine_sim_attention(q, q.float(), q)
    with pytest.raises(ValueError):            # tensors on different devices are refused, not dereferenced
        F.flash_cosine_sim_attention(q, q.cpu(), q)
    with pytest.raises(ValueError):
        F.flash_cosine_sim_attention(q, q, q, mask=torch.ones(1, 8, dtype=torch.bool))
    with pytest.raises(RuntimeError):          # the CPU path is forward-only (flash_cosine_sim_attention.py:142-144)
        F.flash_cosine_sim_attention(q.cpu().float().requires_grad_(), q.cpu().float(), q.cpu().float())


def test_no_grad_path_skips_saved_state():
    import flash_cosine_sim_attention_amd as F
    q = torch.randn(1, 2, 70, 64, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        o = F.flash_cosine_sim_attention(q, q, q, causal=True)
    assert not o.requires_grad and torch.isfinite(o).all()
    # inputs that require grad, under no_grad: still the inference path -- nothing saved, same output
    qg = q.clone().requires_grad_()
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        o2 = F.flash_cosine_sim_attention(qg, qg, qg, causal=True)
    assert not o2.requires_grad and o2.grad_fn is None and torch.equal(o2, o)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() - base <= o2.numel() * o2.element_size() + 4096      # only `o` outlives the call


def test_hip_at_least_as_accurate_as_pytorch_same_dtype():
    """The metric's 'max-|d| vs PyTorch ref': error of the HIP op vs exact math must not exceed the
    error of the reference-style PyTorch composite run in the same dtype on the same GPU."""
    import flash_cosine_sim_attention_amd as F
    torch.manual_seed(2)
    for dtype in (torch.float16, torch.bfloat16):
        q, k, v = (torch.randn(2, 4, 500, 64, device="cuda", dtype=dtype) for _ in range(3))
        exact, _ = O.attention_forward_stats(_np(q), _np(k), _np(v), causal=True)
        hip = _np(F.flash_cosine_sim_attention(q, k, v, causal=True))
        ref = _np(F.plain_cosine_sim_attention(q, k, v, causal=True))
        e_hip, e_ref = np.abs(hip - exact).max(), np.abs(ref - exact).max()
        assert e_hip <= 1.25 * e_ref + 1e-4, (dtype, e_hip, e_ref)
