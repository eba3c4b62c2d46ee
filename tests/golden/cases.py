"""Golden-vector case table shared by make_golden.py (generator) and the tests.

The grid follows the reference's own test parametrisation (tests/test.py:31-37:
(causal, mask) in {(T,F),(F,T),(F,F)} x attn_bias x seq_len in {63,127} x
dim_head in {32,64,96,128} x dtype x attn_bias_batch_dim x single_head_kv),
shrunk in batch/heads to keep the fixtures small, plus the gaps the reference
never tests: bf16, D=16, N != M, causal with M != N, single-head-KV
gradients, groups > 1, merged batch-heads, l2norm_qk=False, rows without any
valid key.
"""
import numpy as np
import torch

DTYPES = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def _c(name, b=1, h=2, n=63, m=None, d=32, dtype="f32", causal=False, mask=False, bias=False,
       bias_batch=False, single_kv=False, merged=False, scale=8, groups=1, l2norm=True,
       mask_kind="random", tiled_ok=True, seed=0, bias_std=1.0, wide=False):
    return dict(name=name, b=b, h=h, n=n, m=(n if m is None else m), d=d, dtype=dtype, causal=causal,
                mask=mask, bias=bias, bias_batch=bias_batch, single_kv=single_kv, merged=merged,
                scale=scale, groups=groups, l2norm=l2norm, mask_kind=mask_kind, tiled_ok=tiled_ok, seed=seed,
                bias_std=bias_std, wide=wide)


CASES = [
    # --- the reference's own grid (tests/test.py:31-37), shrunk ---------------------------------
    _c("g01_dense_d64_n63_f32", d=64, n=63, seed=1),
    _c("g02_causal_d64_n127_f32", h=1, d=64, n=127, causal=True, seed=2),
    _c("g03_mask_d32_n63_f32", b=2, d=32, n=63, mask=True, seed=3),
    _c("g04_causal_biasH_d32_n63_f32", d=32, n=63, causal=True, bias=True, seed=4),
    _c("g05_mask_biasB_d64_n63_f16", b=2, h=2, d=64, n=63, mask=True, bias=True, bias_batch=True, dtype="f16", seed=5),
    _c("g06_dense_singlekv_d64_n63_f16", b=1, d=64, n=63, single_kv=True, dtype="f16", seed=6),
    _c("g07_causal_singlekv_d128_n63_f32", b=1, h=3, d=128, n=63, causal=True, single_kv=True, seed=7),
    _c("g08_mask_d96_n127_f32", b=2, h=1, d=96, n=127, mask=True, seed=8),
    _c("g09_dense_biasH_d128_n63_f16", h=2, d=128, n=63, bias=True, dtype="f16", seed=9),
    # --- gaps the reference never tests ---------------------------------------------------------
    _c("g10_dense_d64_n127_bf16", h=1, d=64, n=127, dtype="bf16", seed=10),
    _c("g11_causal_d64_n127_bf16", h=1, d=64, n=127, causal=True, dtype="bf16", seed=11),
    _c("g12_dense_d16_n63_f32", d=16, n=63, seed=12),
    _c("g13_cross_n63_m200_d32_f32", d=32, n=63, m=200, seed=13),
    _c("g14_causal_n63_m100_d32_f32", d=32, n=63, m=100, causal=True, seed=14),
    # causal with M < N: the first N-M rows have no valid key (plain -> mean(V); kernel -> 0)
    _c("g15_causal_n100_m63_d32_f32", d=32, n=100, m=63, causal=True, seed=15),
    _c("g16_groups2_scale1_d64_f32", d=64, n=63, groups=2, scale=1, seed=16),
    _c("g17_groups8_d128_causal_singlekv_bf16", b=1, d=128, n=63, groups=8, scale=1, causal=True,
       single_kv=True, dtype="bf16", seed=17),
    _c("g18_merged_bh_d32_f32", b=3, d=32, n=63, merged=True, seed=18),
    _c("g19_nol2norm_d32_f32", d=32, n=63, l2norm=False, scale=0.125, seed=19),
    # a batch element whose key mask is all False (plain -> mean(V); kernel -> 0)
    _c("g20_mask_fullrow_d32_f32", b=2, d=32, n=63, mask=True, mask_kind="one_batch_empty", seed=20),
    # multi-tile causal: the reference's tiled CPU path is wrong here (see oracle), plain is right
    _c("g21_causal_n600_d16_f32", h=1, d=16, n=600, causal=True, tiled_ok=False, seed=21),
    _c("g22_mask_prefix_n150_m260_d64_f16", b=2, h=1, d=64, n=150, m=260, mask=True, mask_kind="prefix",
       dtype="f16", seed=22),
    _c("g23_merged_biasB_mask_d64_f32", b=2, d=64, n=63, merged=True, bias=True, mask=True, seed=23),
    # --- round 3: the shapes its changes are about ----------------------------------------------
    # float16 with a bias of sigma = 3 at scale 1: the configuration class whose P~ overflowed with a constant exponent shift
    _c("g24_dense_biasH_std3_scale1_d64_f16", d=64, n=63, m=100, bias=True, bias_std=3.0, scale=1, dtype="f16", seed=24),
    # few keys, many queries (split-query dK/dV), key mask
    _c("g25_mask_n1100_m70_d16_f32", h=1, d=16, n=1100, m=70, mask=True, seed=25),
    # few queries, many keys (split-key forward / dQ), prefix mask
    _c("g26_mask_prefix_n40_m1100_d16_f16", h=1, d=16, n=40, m=1100, mask=True, mask_kind="prefix", dtype="f16",
       seed=26),
    # ragged non-causal tiles at D = 96 with grouped l2norm (rank-1 tail masks)
    _c("g27_dense_n200_m333_d96_groups2_bf16", h=1, d=96, n=200, m=333, groups=2, scale=4, dtype="bf16", seed=27),
    # --- round 4: wide logit ranges (`wide`: the 16-bit rounding of q^, k^ is amplified by scale * groups, in the reference's own
    # 16-bit evaluation too; tests compare twice -- against these fixtures with the bar times logit_cond(), and against exact
    # arithmetic on the 16-bit operands (oracle operand_dtype) with the FIXED bar).  The reference's tiled CPU path and its kernel
    # form exp(S - scale) leave float32 at these ranges (tiled_ok = False); plain_cosine_sim_attention is the reference here.
    _c("g28_dense_groups8_scale8_n320_d64_bf16", h=1, d=64, n=320, groups=8, scale=8, dtype="bf16", tiled_ok=False, wide=True,
       seed=28),                                                     # scale * groups = 64: top of the static bf16 window region
    _c("g29_causal_groups8_scale12_n300_d128_bf16", h=1, d=128, n=300, groups=8, scale=12, causal=True, dtype="bf16",
       tiled_ok=False, wide=True, seed=29),                          # 96: beyond the window, per-row online reference
    _c("g30_mask_scale16_n320_m400_d64_f16", h=1, d=64, n=320, m=400, mask=True, scale=16, dtype="f16", tiled_ok=False, wide=True,
       seed=30),                                                     # f16 beyond its window (11)
    _c("g31_dense_biasH_std3_n300_m330_d64_f16", h=1, d=64, n=300, m=330, bias=True, bias_std=3.0, dtype="f16", tiled_ok=False,
       wide=True, seed=31),                                          # f16 + bias of sigma 3 at the default scale, N >= 300
]

BY_NAME = {c["name"]: c for c in CASES}


def logit_cond(dtype, scale, groups, l2norm=True):
    """Amplification of the 16-bit rounding of q^, k^ by the logit range: the logits are scale * sum_g cos_g with cos_g carrying
    the 2^-9 (bf16) / 2^-12 (f16) rounding of the normalised operands -- which the reference rounds too -- so comparisons against
    exact math ON THE RAW INPUTS scale their bar with max(1, scale * groups / 16); comparisons against exact math on the 16-bit
    operands (oracle `operand_dtype`) do not."""
    return max(1.0, abs(scale) * groups / 16.0) if (dtype != "f32" and l2norm) else 1.0


def dynamic_shift_regime(dtype, scale, groups, l2norm, bias):
    """fcsa_capi.hip dynamic_shift(): logit ranges no constant exponent shift can hold run the per-row online reference and are
    normalised exactly (no 1e-10 clamp in exp(S - scale) units, like the reference's plain_cosine_sim_attention)."""
    bound = abs(scale) * groups
    if not l2norm:
        return False
    if dtype == "f16":
        return bound > 11 or bool(bias)
    return bound > 75 or (bool(bias) and bound > 40)


def op_kwargs(case):
    return dict(scale=case["scale"], groups=case["groups"], causal=case["causal"],
                l2norm_qk=case["l2norm"], attn_bias_batch_dim=case["bias_batch"])


def make_inputs(case, device="cpu"):
    """Seeded inputs (numpy legacy RandomState, stable across numpy versions), rounded to the case dtype."""
    rs = np.random.RandomState(1000 + case["seed"])
    dt = DTYPES[case["dtype"]]
    b, h, n, m, d = case["b"], case["h"], case["n"], case["m"], case["d"]

    def t(*shape):
        return torch.from_numpy(rs.standard_normal(shape).astype(np.float32)).to(dt).to(device)

    if case["merged"]:
        q, k, v = t(b, n, d), t(b, m, d), t(b, m, d)
    else:
        q = t(b, h, n, d)
        kvs = (b, m, d) if case["single_kv"] else (b, h, m, d)
        k, v = t(*kvs), t(*kvs)
    do = t(*q.shape)
    mask = None
    if case["mask"]:
        if case["mask_kind"] == "prefix":
            lens = rs.randint(1, m + 1, size=(b,))
            mk = np.arange(m)[None, :] < lens[:, None]
        else:
            mk = rs.randint(0, 2, size=(b, m)).astype(bool)
            mk[:, 0] = True
            if case["mask_kind"] == "one_batch_empty":
                mk[b - 1, :] = False
        mask = torch.from_numpy(mk).to(device)
    bias = None
    if case["bias"]:
        lead = b if (case["bias_batch"] or case["merged"]) else h
        bias = t(lead, n, m)
        if case.get("bias_std", 1.0) != 1.0:
            bias = (bias.float() * case["bias_std"]).to(dt)
    return dict(q=q, k=k, v=v, do=do, mask=mask, attn_bias=bias)
